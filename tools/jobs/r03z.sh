#!/bin/bash
OUT=gpurun_out/r03z; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_culled_binning.py tests/test_gpu_heavy_tail.py tests/test_gpu_guard.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for r in 1 2; do for v in bwd_mb16 bwd_mb16b; do
  echo "== $v (round $r)"
  for c in C4 C3 C2; do GSR_LIB_PATH=$PWD/tools/ab/$v.so PYTHONPATH=.:tools timeout -k 10 200 python tools/kernel_bench.py $c 20 --fused 2>/dev/null | grep -E "render_bwd"; done
  GSR_LIB_PATH=$PWD/tools/ab/$v.so PYTHONPATH=.:tools timeout -k 10 200 python tools/bench_heavy_tail.py 6000000 5 2>/dev/null | grep -E "render_bwd"
done; done | tee $OUT/ab_bwd_mb16b.txt
for c in C4 C3; do GSR_LIB_PATH=$PWD/tools/ab/bwd_prof.so PYTHONPATH=.:tools timeout -k 10 200 python tools/bwd_profile.py $c 5 2>&1 | tail -7; done | tee $OUT/bwd_profile.txt
