#!/bin/bash
OUT=gpurun_out/r03final; mkdir -p $OUT
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/pytest.log
[ $rc -eq 0 ] || exit 1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > $OUT/bench_c4_n1.json 2> $OUT/bench_c4_n1.err; echo "bench c4 rc=$?"
python bench.py --config C3 --no-cpu-baseline > $OUT/bench_c3_n1.json 2> $OUT/bench_c3_n1.err
python bench.py --config C2 --no-cpu-baseline > $OUT/bench_c2_n1.json 2> $OUT/bench_c2_n1.err
python - <<'PY'
import json
for n in ("c4","c3","c2"):
    d=json.loads(open(f'gpurun_out/r03final/bench_{n}_n1.json').read().strip().splitlines()[-1])
    print(n, d['value'], d['ms_per_step'], d['fwd_ms_per_step'], (d.get('fwd_multi_stream') or {}).get('mpixels_per_s'), d['roofline']['traffic'])
PY
