#!/bin/bash
OUT=gpurun_out/r03ba; mkdir -p $OUT
for r in 1 2; do for v in pre_base pre_rec pre_sh pre_both; do
  echo "== $v (round $r)"
  GSR_LIB_PATH=$PWD/tools/ab/$v.so PYTHONPATH=.:tools timeout -k 10 200 python tools/kernel_bench.py C4 20 --fused 2>/dev/null | grep -E "forward-only" | cut -c1-60
done; done | tee $OUT/probe_preprocess_fwd_patterns.txt
