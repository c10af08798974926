#!/bin/bash
OUT=gpurun_out/r03m; mkdir -p $OUT
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
export PYTHONPATH=$PWD:$PWD/tools
python tools/host_overhead.py C2 2>&1 | grep -E "host issue"
python tools/bench_graphed.py C2 300 2>&1 | grep -v amdgpu
python bench.py --config C2 --no-cpu-baseline > $OUT/bench_C2.json 2>/dev/null; python -c "
import json; j=json.load(open('$OUT/bench_C2.json')); print('C2', j['value'], j['fwd_ms_per_step'], j['ms_per_step'], j['unfused_fwd_ms'], j['unfused_train_ms'])"
