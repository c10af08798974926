#!/bin/bash
OUT=gpurun_out/r03aj; mkdir -p $OUT
for c in C4 C3 heavy; do GSR_LIB_PATH=$PWD/tools/ab/bwd_prof.so PYTHONPATH=.:tools timeout -k 10 200 python tools/bwd_profile.py $c 5 2>&1 | tail -7; done | tee $OUT/bwd_profile.txt
