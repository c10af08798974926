#!/bin/bash
OUT=gpurun_out/r03ar; mkdir -p $OUT
export PYTHONPATH=$PWD:$PWD/tools
for c in C4 C3 C2; do timeout -k 10 200 python tools/bench_two_streams.py $c 100 2>&1 | tail -3; done | tee $OUT/two_streams.txt
