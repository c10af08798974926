#!/bin/bash
# several builds of libgsr_hip.so on the same box: tools/ab_multi.sh <config> <iters> <lib.so>...   (two rounds, interleaved)
set -e
cfg=$1; iters=$2; shift 2
lib=mvs_gaussian_splatting_amd/libgsr_hip.so
cp $lib /tmp/orig.so
for r in 1 2; do
  for v in "$@"; do
    cp $v $lib; echo "== $v (round $r)"; PYTHONPATH=.:tools python tools/kernel_bench.py $cfg $iters --fused 2>/dev/null
  done
done
cp /tmp/orig.so $lib
