#!/bin/bash
OUT=$PWD/gpurun_out/r03ap; mkdir -p $OUT
export PYTHONPATH=$PWD:$PWD/tools
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "decisions_pixel_by_pixel" > $OUT/pytest_decisions.log 2>&1; echo "pytest rc=$?"; grep -E "fwd/bwd decisions|passed|failed|Error|assert" $OUT/pytest_decisions.log | head -20
for s in 83 85 106 111 131 139; do timeout -k 10 120 python tools/fuzz_more.py $s $((s+1)) 2>&1 | grep -E "unmasked|seed $s|failed seeds"; done | cut -c1-330 | tee $OUT/fuzz_refailed.log
