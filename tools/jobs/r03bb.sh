#!/bin/bash
OUT=gpurun_out/r03bb; mkdir -p $OUT
python bench.py --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench_c4.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03bb/bench_c4.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['hbm_copy_measured_GBs'], d['hbm_elementwise_kernel_GBs'], d['roofline']['sane'])
PY
