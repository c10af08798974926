#!/bin/bash
OUT=gpurun_out/r03r; mkdir -p $OUT
for r in 1 2; do for v in sb16 fwd_occ6 fwd_occ4; do
  echo "== $v (round $r)"
  GSR_LIB_PATH=$PWD/tools/ab/$v.so PYTHONPATH=.:tools timeout -k 10 200 python tools/kernel_bench.py C4 20 --fused 2>/dev/null | grep -E "render_fwd|forward-only"
  GSR_LIB_PATH=$PWD/tools/ab/$v.so PYTHONPATH=.:tools timeout -k 10 200 python tools/kernel_bench.py C3 20 --fused 2>/dev/null | grep -E "render_fwd "
done; done > $OUT/fwd_occupancy.txt 2>&1
cat $OUT/fwd_occupancy.txt
