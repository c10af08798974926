#!/bin/bash
# preprocess_fwd: occupancy sensitivity (dummy LDS) + L1/L2 request counters of the forward
OUT=gpurun_out/r03h; mkdir -p $OUT
for v in occ4 occ3 occ2; do
  echo "== $v"; GSR_LIB_PATH=$PWD/tools/ab/$v.so PYTHONPATH=.:tools timeout -k 10 200 python tools/kernel_bench.py C4 20 --fused 2>/dev/null | grep -E "preprocess_fwd|fwd="
done > $OUT/occupancy.txt 2>&1
cat $OUT/occupancy.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/tools
timeout -k 10 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/tcp -- python3 tools/fwd_loop.py C4 3 > $OUT/tcp.log 2>&1; echo "rocprof rc=$?"
python3 - <<'PY'
import csv, glob, collections
fs = glob.glob("gpurun_out/r03h/tcp/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in fs:
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k[:60], {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
