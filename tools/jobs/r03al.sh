#!/bin/bash
OUT=$PWD/gpurun_out/r03al; mkdir -p $OUT
export PYTHONPATH=$PWD:$PWD/tools
timeout -k 10 800 python tools/fuzz_more.py 8 72 > $OUT/fuzz_8_72.log 2>&1; echo "fuzz rc=$?"; tail -2 $OUT/fuzz_8_72.log
python bench.py > $OUT/bench_c4_n1.json 2> $OUT/bench_c4_n1.err; tail -c 900 $OUT/bench_c4_n1.json
