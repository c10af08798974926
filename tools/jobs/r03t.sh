#!/bin/bash
OUT=gpurun_out/r03t; mkdir -p $OUT
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
for r in 1 2; do for v in pbwd_rec pbwd_bin; do
  echo "== $v (round $r)"
  GSR_LIB_PATH=$PWD/tools/ab/$v.so PYTHONPATH=.:tools timeout -k 10 200 python tools/kernel_bench.py C4 20 --fused 2>/dev/null | grep -E "preprocess_bwd"
  GSR_LIB_PATH=$PWD/tools/ab/$v.so PYTHONPATH=.:tools timeout -k 10 200 python tools/kernel_bench.py C3 20 --fused 2>/dev/null | grep -E "preprocess_bwd"
  GSR_LIB_PATH=$PWD/tools/ab/$v.so PYTHONPATH=.:tools timeout -k 10 200 python tools/bench_heavy_tail.py 6000000 5 2>/dev/null | grep -E "preprocess_bwd"
done; done | tee $OUT/ab_pbwd_bininfo.txt
