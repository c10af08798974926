#!/bin/bash
OUT=gpurun_out/r03bc; mkdir -p $OUT
GSR_LIB_PATH=$PWD/tools/ab/pre_reclds.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_culled_binning.py tests/test_gpu_golden_preprocess.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
for r in 1 2 3; do for v in pre_base pre_reclds; do
  echo "== $v (round $r)"
  GSR_LIB_PATH=$PWD/tools/ab/$v.so PYTHONPATH=.:tools timeout -k 10 200 python tools/kernel_bench.py C4 20 --fused 2>/dev/null | grep -E "forward-only|preprocess_bwd|render_bwd" | cut -c1-70
done; done | tee $OUT/ab_rec_via_lds.txt
