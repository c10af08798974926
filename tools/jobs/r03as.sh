#!/bin/bash
OUT=gpurun_out/r03as; mkdir -p $OUT
export PYTHONPATH=$PWD:$PWD/tools
timeout -k 10 300 python -m pytest tests/test_gpu_graphed.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
for c in C4 C3 C2; do timeout -k 10 300 python tools/bench_multi_stream.py $c 96 2>&1 | grep -v amdgpu.ids; done | tee $OUT/multi_stream.txt
