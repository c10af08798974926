#!/bin/bash
OUT=gpurun_out/r03ac; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_miniblock_cull.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest.log
