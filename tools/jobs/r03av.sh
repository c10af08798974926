#!/bin/bash
OUT=$PWD/gpurun_out/r03av; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "decisions_pixel_by_pixel" > $OUT/pytest_decisions.log 2>&1; echo "pytest rc=$?"; grep -E "fwd/bwd decisions|passed|failed" $OUT/pytest_decisions.log | tail -4
