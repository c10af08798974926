#!/bin/bash
OUT=gpurun_out/r03cap7; mkdir -p $OUT
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 800 bash tools/capture_profiles.sh r03cap7
