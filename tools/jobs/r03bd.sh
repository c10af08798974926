#!/bin/bash
ROOT=$PWD; OUT=$ROOT/gpurun_out/r03bd; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
rocprofv3 --pmc VALUBusy MemUnitBusy MemUnitStalled WriteUnitStalled --output-format csv -d $OUT/mem1 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prewarm > $OUT/mem1.log 2>&1; echo "pass1 rc=$?"
rocprofv3 --pmc TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/mem2 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prewarm > $OUT/mem2.log 2>&1; echo "pass2 rc=$?"
python - <<'PY'
import csv, glob, collections
for d in ("mem1","mem2"):
    fs = glob.glob(f"gpurun_out/r03bd/{d}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in fs:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        if "gsr::" in k and any(s in k for s in ("preprocess", "render_", "scatter", "emit", "compact")):
            print(d, k, {n: round(sum(v)/len(v), 2) for n, v in c.items()})
PY
