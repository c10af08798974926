#!/bin/bash
OUT=gpurun_out/r03at; mkdir -p $OUT
python bench.py --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench_c4.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03at/bench_c4.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['fwd_ms_per_step'], d.get('fwd_multi_stream'), d['roofline']['traffic'])
PY
python bench.py --config C3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C3', d['value'], d['ms_per_step'], d.get('fwd_multi_stream'))"
