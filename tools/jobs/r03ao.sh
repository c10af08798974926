#!/bin/bash
OUT=$PWD/gpurun_out/r03ao; mkdir -p $OUT
export PYTHONPATH=$PWD:$PWD/tools
timeout -k 10 1000 python tools/fuzz_more.py 72 180 > $OUT/fuzz_72_180.log 2>&1; echo "fuzz rc=$?"; tail -2 $OUT/fuzz_72_180.log
