"""GPU idle time between consecutive kernels of a rocprofv3 --kernel-trace run:
python tools/gap_analysis.py <dir with *kernel_trace.csv> [skip_first_n_kernels]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[-40:]))
rows.sort()
rows = rows[skip:]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = collections.defaultdict(lambda: [0, 0])
end = rows[0][1]
for (s, e, name), prev in zip(rows[1:], rows[:-1]):
    g = s - end
    if g > 0:
        key = f"{prev[2]} -> {name}"
        gaps[key][0] += g
        gaps[key][1] += 1
    end = max(end, e)
print(f"kernels {len(rows)}  span {span / 1e6:.3f} ms  busy {busy / 1e6:.3f} ms  idle {(span - busy) / 1e6:.3f} ms ({100 * (span - busy) / span:.1f} %)")
for k, (t, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"  {t / 1e3:9.1f} us total  {t / n / 1e3:7.1f} us avg x{n:4d}   {k}")
