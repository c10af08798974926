#!/bin/bash
OUT=gpurun_out/r03ax; mkdir -p $OUT
timeout -k 10 200 tools/ubench/bin/hbm_stream | tee $OUT/hbm_stream.txt
