#!/bin/bash
OUT=gpurun_out/r03az; mkdir -p $OUT
for r in 1 2; do for v in pre_base pre_fast; do
  echo "== $v (round $r)"
  GSR_LIB_PATH=$PWD/tools/ab/$v.so PYTHONPATH=.:tools timeout -k 10 200 python tools/kernel_bench.py C4 20 --fused 2>/dev/null | grep -E "preprocess_fwd|preprocess_bwd"
done; done | tee $OUT/ab_preprocess_fastmath.txt
