#!/bin/bash
OUT=gpurun_out/r03an; mkdir -p $OUT
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for r in 1 2; do for v in row48 row36; do
  echo "== $v (round $r)"
  for c in C4 C3; do GSR_LIB_PATH=$PWD/tools/ab/$v.so PYTHONPATH=.:tools timeout -k 10 200 python tools/kernel_bench.py $c 20 --fused 2>/dev/null | grep -E "render_bwd|preprocess_bwd|bwd="; done
  GSR_LIB_PATH=$PWD/tools/ab/$v.so PYTHONPATH=.:tools timeout -k 10 200 python tools/bench_heavy_tail.py 6000000 5 2>/dev/null | grep -E "render_bwd|preprocess_bwd|backward"
done; done | tee $OUT/ab_row36.txt
