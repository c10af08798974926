#!/bin/bash
# A/B of two builds of libgsr_hip.so on the same box: tools/ab_bench.sh <base.so> [config] [iters] [rounds]
# (per-stage device times of tools/kernel_bench.py, alternating base / new so that clock drift hits both).
# The binding is pointed at each build through GSR_LIB_PATH: the in-tree library is never overwritten.
set -e
base=$(realpath "$1"); cfg=${2:-C4}; iters=${3:-20}; rounds=${4:-2}
new=$(realpath mvs_gaussian_splatting_amd/libgsr_hip.so)
for r in $(seq $rounds); do
  echo "== base (round $r)"; GSR_LIB_PATH=$base PYTHONPATH=.:tools python tools/kernel_bench.py $cfg $iters --fused
  echo "== new (round $r)"; GSR_LIB_PATH=$new PYTHONPATH=.:tools python tools/kernel_bench.py $cfg $iters --fused
done
