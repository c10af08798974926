#!/bin/bash
OUT=$PWD/gpurun_out/r03p; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/tools
timeout -k 10 300 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d $OUT/markers -- python3 tools/marker_demo.py > $OUT/markers.log 2>&1; echo "rocprof rc=$?"
python3 - <<'PY'
import csv, glob, collections
fs = glob.glob("gpurun_out/r03p/markers/**/*marker_api_trace.csv", recursive=True) + glob.glob("gpurun_out/r03p/markers/**/*marker*.csv", recursive=True)
print(sorted(set(fs)))
for f in sorted(set(fs))[:2]:
    rows = list(csv.DictReader(open(f)))
    print(f, len(rows), rows[:3])
PY
timeout -k 10 700 python tools/fuzz_more.py 8 40 > $OUT/fuzz_8_40.log 2>&1; echo "fuzz rc=$?"; tail -3 $OUT/fuzz_8_40.log
