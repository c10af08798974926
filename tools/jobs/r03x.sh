#!/bin/bash
OUT=gpurun_out/r03y; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for r in 1 2; do for v in bwd_sb16 bwd_mb16; do
  echo "== $v (round $r)"
  for c in C4 C3 C2; do GSR_LIB_PATH=$PWD/tools/ab/$v.so PYTHONPATH=.:tools timeout -k 10 200 python tools/kernel_bench.py $c 20 --fused 2>/dev/null | grep -E "render_bwd"; done
  GSR_LIB_PATH=$PWD/tools/ab/$v.so PYTHONPATH=.:tools timeout -k 10 200 python tools/bench_heavy_tail.py 6000000 5 2>/dev/null | grep -E "render_bwd"
done; done | tee $OUT/ab_bwd_mb16.txt
for c in C4 C3 heavy; do GSR_LIB_PATH=$PWD/tools/ab/bwd_prof.so PYTHONPATH=.:tools timeout -k 10 200 python tools/bwd_profile.py $c 5 2>&1 | tail -7; done | tee $OUT/bwd_profile.txt
