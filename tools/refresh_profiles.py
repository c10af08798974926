"""Copies one profiling capture (bench JSON, rocprofv3 kernel stats, PMC passes) into profiles/<round>/ and regenerates
the summaries: python tools/refresh_profiles.py gpurun_out/r01c profiles/r01 "<label>"

The capture is produced on the GPU box by (cd /tmp; export TMPDIR=/tmp first):
  python bench.py                                                             > <cap>/bench_c4_n1.json
  rocprofv3 --kernel-trace --stats --output-format csv -d <cap>/stats -- python bench.py --no-cpu-baseline
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d <cap>/fetch -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d <cap>/write -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU \\
            --output-format csv -d <cap>/sq -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
"""
import collections
import csv
import glob
import os
import shutil
import subprocess
import sys

src, dst = sys.argv[1], sys.argv[2]
label = sys.argv[3] if len(sys.argv) > 3 else ""
os.makedirs(dst, exist_ok=True)
one = lambda pat: glob.glob(os.path.join(src, pat), recursive=True)[0]  # noqa: E731
shutil.copy(os.path.join(src, "bench_c4_n1.json"), os.path.join(dst, "bench_c4_n1.json"))
ks = one("stats/**/*_kernel_stats.csv")
shutil.copy(ks, os.path.join(dst, "rocprofv3_kernel_stats_bench_c4.csv"))
shutil.copy(one("fetch/**/*_counter_collection.csv"), os.path.join(dst, "pmc_FETCH_SIZE_bench_c4_counter_collection.csv"))
shutil.copy(one("write/**/*_counter_collection.csv"), os.path.join(dst, "pmc_WRITE_SIZE_bench_c4_counter_collection.csv"))
shutil.copy(one("sq/**/*_counter_collection.csv"), os.path.join(dst, "pmc_SQ_bench_c4_counter_collection.csv"))
rows = list(csv.DictReader(open(ks)))
with open(os.path.join(dst, "kernel_stats_summary.txt"), "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu-baseline   (C4: 6M Gaussians, "
            f"1080p, 1x MI355X; {label})\n# kernel | calls | avg us | % of GPU time\n")
    for r in rows[:28]:
        f.write(f"{r['Name'][:100]} | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f}\n")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(os.path.join(dst, "pmc_SQ_bench_c4_counter_collection.csv"))):
    agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
stats = {r["Name"].split("(")[0].replace("void ", ""): float(r["AverageNs"]) for r in rows}
with open(os.path.join(dst, "pmc_SQ_render_summary.txt"), "w") as f:
    f.write("# rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU "
            "-- python bench.py --steps 3 --warmup 1 --no-cpu-baseline\n# per-launch means; valu_issue_ms = SQ_INSTS_VALU x 4 cycles / "
            "(1024 SIMDs x 2.4 GHz): the time the wave64 VALU instructions alone need at one issue per 4 cycles per SIMD\n")
    for k in agg:
        if any(s in k for s in ("render_fwd", "render_bwd", "preprocess_fwd", "preprocess_bwd")):
            m = {c: sum(v) / len(v) for c, v in agg[k].items()}
            valu_ms = m.get("SQ_INSTS_VALU", 0) * 4 / (1024 * 2.4e9) * 1e3
            avg = stats.get(k, 0) / 1e6
            f.write(f"{k}: " + ", ".join(f"{c}={int(v)}" for c, v in sorted(m.items())) +
                    f" | valu_issue_ms={valu_ms:.3f} measured_avg_ms={avg:.3f} ratio={valu_ms / avg if avg else 0:.2f}\n")
here = os.path.dirname(os.path.abspath(__file__))
subprocess.run([sys.executable, os.path.join(here, "traffic_from_pmc.py"), os.path.join(src, "fetch"), os.path.join(src, "write"), "C4"],
               check=True, stdout=subprocess.DEVNULL)
# VALU instruction counts next to the HBM bytes: bench.py reports the issue floor of the dominant kernel from them
import json
tpath = os.path.join(os.path.dirname(os.path.abspath(dst.rstrip("/"))), "traffic.json")
tj = json.load(open(tpath))
names = {"preprocess_fwd_kernel": "preprocess_fwd", "render_fwd_kernel": "render_fwd", "render_bwd_kernel": "render_bwd",
         "preprocess_bwd_kernel": "preprocess_bwd"}
for k, m in agg.items():
    base = k.split("<")[0].split("::")[-1]
    if base in names and "SQ_INSTS_VALU" in m:
        e = tj["C4"].setdefault(names[base], {})
        vals = e.setdefault("_valu", [])
        vals.append(sum(m["SQ_INSTS_VALU"]) / len(m["SQ_INSTS_VALU"]))
for e in tj["C4"].values():
    if "_valu" in e:
        v = e.pop("_valu")
        e["valu_insts_per_launch"] = int(max(v))      # the tracking variant when a kernel has two
        e["valu_method"] = "rocprofv3 --pmc SQ_INSTS_VALU, per-launch mean (wave-level instructions)"
json.dump(tj, open(tpath, "w"), indent=1)
print(open(os.path.join(dst, "pmc_SQ_render_summary.txt")).read())
