"""Per-kernel means of rocprofv3 --pmc passes: python tools/pmc_table.py <dir> [<dir> ...] [--match render]"""
import collections, csv, glob, sys
dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
match = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--match=")]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    if match and not any(m in k for m in match):
        continue
    m = {c: sum(v) / len(v) for c, v in agg[k].items()}
    print(k[:90])
    for c in sorted(m):
        print(f"    {c:28s} {m[c]:16.0f}")
