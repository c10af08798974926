#!/bin/bash
OUT=gpurun_out/r03aq; mkdir -p $OUT
GSR_LIB_PATH=$PWD/tools/ab/batch60.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_culled_binning.py tests/test_gpu_miniblock_cull.py -m gpu -x -q > $OUT/pytest_b60.log 2>&1; echo "b60 pytest rc=$?"; tail -2 $OUT/pytest_b60.log
for r in 1 2; do for v in batch64 batch60 batch56; do
  echo "== $v (round $r)"
  for c in C4 C3 C2; do GSR_LIB_PATH=$PWD/tools/ab/$v.so PYTHONPATH=.:tools timeout -k 10 200 python tools/kernel_bench.py $c 20 --fused 2>/dev/null | grep -E "render_bwd"; done
  GSR_LIB_PATH=$PWD/tools/ab/$v.so PYTHONPATH=.:tools timeout -k 10 200 python tools/bench_heavy_tail.py 6000000 5 2>/dev/null | grep -E "render_bwd"
done; done | tee $OUT/ab_bwd_batch.txt
