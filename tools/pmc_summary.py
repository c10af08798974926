"""Averages rocprofv3 --pmc counter_collection.csv per kernel: python tools/pmc_summary.py <dir> [filter]"""
import collections, csv, glob, sys
d = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[-48:]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    if flt in k:
        print(k, {c: round(sum(v) / len(v)) for c, v in sorted(agg[k].items())})
