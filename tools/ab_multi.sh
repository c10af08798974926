#!/bin/bash
# several builds of libgsr_hip.so on the same box: tools/ab_multi.sh <config> <iters> <lib.so>...   (two rounds, interleaved)
# Each build is loaded through GSR_LIB_PATH; the in-tree library is never overwritten.
set -e
cfg=$1; iters=$2; shift 2
for r in 1 2; do
  for v in "$@"; do
    echo "== $v (round $r)"; GSR_LIB_PATH=$(realpath "$v") PYTHONPATH=.:tools python tools/kernel_bench.py $cfg $iters --fused 2>/dev/null
  done
done
