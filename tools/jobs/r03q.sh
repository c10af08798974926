#!/bin/bash
OUT=gpurun_out/r03q; mkdir -p $OUT
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_culled_binning.py tests/test_gpu_heavy_tail.py tests/test_gpu_guard.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
for r in 1 2; do for v in sb16 half8; do
  echo "== $v (round $r)"
  GSR_LIB_PATH=$PWD/tools/ab/$v.so PYTHONPATH=.:tools timeout -k 10 200 python tools/kernel_bench.py C4 20 --fused 2>/dev/null | grep -E "render_bwd"
  GSR_LIB_PATH=$PWD/tools/ab/$v.so PYTHONPATH=.:tools timeout -k 10 200 python tools/kernel_bench.py C3 20 --fused 2>/dev/null | grep -E "render_bwd"
  GSR_LIB_PATH=$PWD/tools/ab/$v.so PYTHONPATH=.:tools timeout -k 10 200 python tools/kernel_bench.py C2 20 2>/dev/null | grep -E "render_bwd"
done; done > $OUT/ab_half.txt 2>&1
cat $OUT/ab_half.txt
for v in sb16 half8; do echo "== $v heavy / clustered"; GSR_LIB_PATH=$PWD/tools/ab/$v.so PYTHONPATH=.:tools timeout -k 10 200 python tools/bench_heavy_tail.py 6000000 5 2>/dev/null | grep -E "render_bwd"; (cd tools && GSR_LIB_PATH=$PWD/ab/$v.so PYTHONPATH=..:. timeout -k 10 200 python bench_clustered.py C3 2000000 2>/dev/null | grep -E "render_bwd|uniform|clustered"); done | tee $OUT/ab_half_heavy.txt
