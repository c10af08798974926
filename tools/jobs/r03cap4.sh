#!/bin/bash
mkdir -p gpurun_out/r03cap4
timeout -k 10 800 python -m pytest tests -m gpu -x -q > gpurun_out/r03cap4/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r03cap4/pytest.log
[ $rc -eq 0 ] && timeout -k 10 900 bash tools/capture_profiles.sh r03cap4
