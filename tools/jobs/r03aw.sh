#!/bin/bash
OUT=gpurun_out/r03aw; mkdir -p $OUT
for r in 1 2; do for v in wpb4 wpb2 wpb1; do
  echo "== $v (round $r)"
  for c in C4 C3 C2; do GSR_LIB_PATH=$PWD/tools/ab/$v.so PYTHONPATH=.:tools timeout -k 10 200 python tools/kernel_bench.py $c 20 --fused 2>/dev/null | grep -E "render_bwd"; done
  GSR_LIB_PATH=$PWD/tools/ab/$v.so PYTHONPATH=.:tools timeout -k 10 200 python tools/bench_heavy_tail.py 6000000 5 2>/dev/null | grep -E "render_bwd"
done; done | tee $OUT/ab_bwd_waves_per_block.txt
