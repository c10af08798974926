#!/bin/bash
OUT=$PWD/gpurun_out/r03s; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "8192 or forward_stages or full_size" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/tools
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/fwd_loop.py C4 20 > $OUT/stats.log 2>&1; echo "rocprof rc=$?"
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r03s/stats/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "gsr::" in r["Name"]:
        print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))
PY
