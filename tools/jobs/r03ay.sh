#!/bin/bash
OUT=gpurun_out/r03ay; mkdir -p $OUT
export PYTHONPATH=$PWD:$PWD/tools
timeout -k 10 200 python tools/probe_sh_degree.py 2>&1 | grep -v amdgpu.ids | tee $OUT/probe_sh_degree.txt
