#!/bin/bash
OUT=$PWD/gpurun_out/r03au; mkdir -p $OUT
export PYTHONPATH=$PWD:$PWD/tools
timeout -k 10 1050 python tools/fuzz_more.py 180 290 > $OUT/fuzz_180_290.log 2>&1; echo "fuzz rc=$?"; tail -3 $OUT/fuzz_180_290.log | cut -c1-300
