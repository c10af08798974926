#!/bin/bash
# A/B of two builds of libgsr_hip.so on the same box: tools/ab_bench.sh <base.so> [config] [iters] [rounds]
# (per-stage device times of tools/kernel_bench.py, alternating base / new so that clock drift hits both)
set -e
base=$1; cfg=${2:-C4}; iters=${3:-20}; rounds=${4:-2}
lib=mvs_gaussian_splatting_amd/libgsr_hip.so
cp $lib /tmp/new.so
for r in $(seq $rounds); do
  cp $base $lib; echo "== base (round $r)"; PYTHONPATH=.:tools python tools/kernel_bench.py $cfg $iters --fused
  cp /tmp/new.so $lib; echo "== new (round $r)"; PYTHONPATH=.:tools python tools/kernel_bench.py $cfg $iters --fused
done
