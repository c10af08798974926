#!/bin/bash
OUT=gpurun_out/r03o; mkdir -p $OUT
for r in 1 2; do for v in bwd_occ4 bwd_occ3 bwd_occ2; do
  echo "== $v (round $r)"; GSR_LIB_PATH=$PWD/tools/ab/$v.so PYTHONPATH=.:tools timeout -k 10 200 python tools/kernel_bench.py C4 20 --fused 2>/dev/null | grep -E "render_bwd"
  GSR_LIB_PATH=$PWD/tools/ab/$v.so PYTHONPATH=.:tools timeout -k 10 200 python tools/kernel_bench.py C3 20 --fused 2>/dev/null | grep -E "render_bwd"
done; done > $OUT/bwd_occupancy.txt 2>&1
cat $OUT/bwd_occupancy.txt
echo "== wide depth"; PYTHONPATH=.:tools timeout -k 10 200 python tools/kernel_bench.py C4 20 --fused --wide-depth 2>/dev/null | grep -v amdgpu | tee $OUT/kernel_bench_c4_wide_depth.txt
