#!/bin/bash
OUT=gpurun_out/r03ak; mkdir -p $OUT
for v in bwd_rowdefer; do GSR_LIB_PATH=$PWD/tools/ab/$v.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_culled_binning.py tests/test_gpu_miniblock_cull.py tests/test_gpu_heavy_tail.py -m gpu -x -q > $OUT/pytest_$v.log 2>&1; echo "$v pytest rc=$?"; tail -5 $OUT/pytest_$v.log; done
for r in 1 2; do for v in bwd_dynchunk bwd_rowdefer; do
  echo "== $v (round $r)"
  for c in C4 C3 C2; do GSR_LIB_PATH=$PWD/tools/ab/$v.so PYTHONPATH=.:tools timeout -k 10 200 python tools/kernel_bench.py $c 20 --fused 2>/dev/null | grep -E "render_bwd"; done
  GSR_LIB_PATH=$PWD/tools/ab/$v.so PYTHONPATH=.:tools timeout -k 10 200 python tools/bench_heavy_tail.py 6000000 5 2>/dev/null | grep -E "render_bwd"
done; done | tee $OUT/ab_bwd_rowdefer.txt
for c in C4 C3; do GSR_LIB_PATH=$PWD/tools/ab/bwd_prof.so PYTHONPATH=.:tools timeout -k 10 200 python tools/bwd_profile.py $c 5 2>&1 | tail -7; done | tee $OUT/bwd_profile.txt
