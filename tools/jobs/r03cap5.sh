#!/bin/bash
OUT=gpurun_out/r03cap5; mkdir -p $OUT
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/pytest.log
[ $rc -eq 0 ] || exit 1
for r in 1 2; do for v in bwd_dynfinal bwd_occ2; do
  echo "== $v (round $r)"
  for c in C4 C3; do GSR_LIB_PATH=$PWD/tools/ab/$v.so PYTHONPATH=.:tools timeout -k 10 200 python tools/kernel_bench.py $c 20 --fused 2>/dev/null | grep -E "render_bwd"; done
done; done | tee $OUT/ab_render_bwd_occupancy_dyn.txt
for c in C4 C3 C2 heavy; do GSR_LIB_PATH=$PWD/tools/ab/bwd_prof.so PYTHONPATH=.:tools timeout -k 10 200 python tools/bwd_profile.py $c 5 2>&1 | tail -7; done | tee $OUT/bwd_profile.txt
timeout -k 10 800 bash tools/capture_profiles.sh r03cap5
