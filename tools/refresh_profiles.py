"""Copies one profiling capture (bench JSON, rocprofv3 kernel stats, PMC passes) into profiles/<round>/ and regenerates
profiles/traffic.json (stamped with the kernel sources the passes were taken on) and the summaries:

    python tools/refresh_profiles.py gpurun_out/r02g profiles/r02 "<label>"

The capture is produced on the GPU box by tools/capture_profiles.sh (cd /tmp; export TMPDIR=/tmp first; --pmc passes are
separate runs, never combined with tracing):
  python bench.py                                                                  > <cap>/bench_c4_n1.json
  rocprofv3 --kernel-trace --stats --output-format csv -d <cap>/stats -- python bench.py --no-cpu-baseline
  rocprofv3 --pmc FETCH_SIZE ... -d <cap>/fetch -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
  rocprofv3 --pmc WRITE_SIZE ... -d <cap>/write -- ...
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU \\
            SQ_WAIT_INST_ANY ... -d <cap>/sq -- ...
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_SCA \\
            SQ_INSTS_VALU_TRANS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY ... -d <cap>/sq2 -- ...
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(here))
src, dst = sys.argv[1], sys.argv[2]
label = sys.argv[3] if len(sys.argv) > 3 else ""
os.makedirs(dst, exist_ok=True)
one = lambda pat: glob.glob(os.path.join(src, pat), recursive=True)[0]  # noqa: E731
for f in glob.glob(os.path.join(src, "bench_*.json")):
    shutil.copy(f, os.path.join(dst, os.path.basename(f)))
ks = one("stats/**/*_kernel_stats.csv")
shutil.copy(ks, os.path.join(dst, "rocprofv3_kernel_stats_bench_c4.csv"))
for d, name in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE"), ("sq", "SQ"), ("sq2", "SQ2"), ("sq3", "SQ3_valu_mix"),
                ("grbm", "GRBM_clock")):
    try:
        shutil.copy(one(f"{d}/**/*_counter_collection.csv"), os.path.join(dst, f"pmc_{name}_bench_c4_counter_collection.csv"))
    except IndexError:
        print("no capture for", d)
rows = list(csv.DictReader(open(ks)))
with open(os.path.join(dst, "kernel_stats_summary.txt"), "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu-baseline   (C4: 6M Gaussians, "
            f"1080p, 1x MI355X; {label})\n# kernel | calls | avg us | % of GPU time\n")
    for r in rows[:30]:
        f.write(f"{r['Name'][:100]} | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f}\n")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for name in ("SQ", "SQ2"):
    p = os.path.join(dst, f"pmc_{name}_bench_c4_counter_collection.csv")
    if os.path.exists(p):
        for r in csv.DictReader(open(p)):
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
stats = {r["Name"].split("(")[0].replace("void ", ""): float(r["AverageNs"]) for r in rows}
with open(os.path.join(dst, "pmc_SQ_render_summary.txt"), "w") as f:
    f.write("# rocprofv3 --pmc <SQ counters, two passes> -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline; per-launch means\n"
            "# issue utilisations at the rates measured by tools/ubench/valu_rate.hip (profiles/r02/valu_rate.txt):\n"
            "#   valu  = SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x 2.4 GHz x t)   (full-rate floor; v_cmp/v_cndmask/v_min/v_max 4.7, v_exp/v_rcp 8.3)\n"
            "#   salu  = SQ_INSTS_SALU x 4 cycles / (1024 SIMDs x 2.4 GHz x t)\n"
            "#   lds   = SQ_LDS_IDX_ACTIVE / (256 CUs x 2.4 GHz x t)\n")
    for k in agg:
        if any(s in k for s in ("render_fwd", "render_bwd", "preprocess_fwd", "preprocess_bwd")):
            m = {c: sum(v) / len(v) for c, v in agg[k].items()}
            t = stats.get(k, 0) / 1e9
            line = f"{k}: " + ", ".join(f"{c}={int(v)}" for c, v in sorted(m.items()))
            if t:
                line += (f" | avg_ms={t * 1e3:.3f} valu={m.get('SQ_INSTS_VALU', 0) * 2 / (1024 * 2.4e9 * t):.2f} "
                         f"salu={m.get('SQ_INSTS_SALU', 0) * 4 / (1024 * 2.4e9 * t):.2f} "
                         f"lds={m.get('SQ_LDS_IDX_ACTIVE', 0) / (256 * 2.4e9 * t):.2f}")
            f.write(line + "\n")
subprocess.run([sys.executable, os.path.join(here, "traffic_from_pmc.py"), os.path.join(src, "fetch"), os.path.join(src, "write"), "C4"],
               check=True, stdout=subprocess.DEVNULL)
# instruction counters next to the HBM bytes + the stamp of the kernel sources: bench.py nulls the fields on a mismatch
import bench  # noqa: E402
tpath = os.path.join(os.path.dirname(os.path.abspath(dst.rstrip("/"))), "traffic.json")
tj = json.load(open(tpath))
names = {"preprocess_fwd_kernel": "preprocess_fwd", "render_fwd_kernel": "render_fwd", "render_bwd_kernel": "render_bwd",
         "preprocess_bwd_kernel": "preprocess_bwd"}
best = {}
for k, m in agg.items():
    base = k.split("<")[0].split("::")[-1]
    if base in names and "SQ_INSTS_VALU" in m:
        sq = {c: sum(v) / len(v) for c, v in m.items()}
        if names[base] not in best or sq["SQ_INSTS_VALU"] > best[names[base]]["SQ_INSTS_VALU"]:
            best[names[base]] = sq          # the tracking variant when a kernel has two
for stage, sq in best.items():
    e = tj["C4"].setdefault(stage, {})
    e.pop("valu_insts_per_launch", None); e.pop("valu_method", None)
    e["sq"] = {c: int(v) for c, v in sq.items()}
    e["sq_method"] = "rocprofv3 --pmc SQ_* (two separate passes), per-launch mean, wave-level instruction counts"
# VALU instruction classes (one pass) and the clock every kernel ran at (GRBM_GUI_ACTIVE is summed over the 8 XCDs: cycles
# = value / 8; duration = the dispatch's own timestamps in that pass) -> the issue-roof model of bench.py
mix = collections.defaultdict(lambda: collections.defaultdict(list))
p3 = os.path.join(dst, "pmc_SQ3_valu_mix_bench_c4_counter_collection.csv")
if os.path.exists(p3):
    for r in csv.DictReader(open(p3)):
        base = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].split("::")[-1]
        if base in names:
            mix[names[base]][r["Counter_Name"]].append(float(r["Counter_Value"]))
clk = collections.defaultdict(list)
pg = os.path.join(dst, "pmc_GRBM_clock_bench_c4_counter_collection.csv")
if os.path.exists(pg):
    for r in csv.DictReader(open(pg)):
        base = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].split("::")[-1]
        dt = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        if base in names and r["Counter_Name"] == "GRBM_GUI_ACTIVE" and dt > 0:
            clk[names[base]].append(float(r["Counter_Value"]) / 8.0 / dt)      # cycles per ns = GHz
with open(os.path.join(dst, "pmc_valu_mix_summary.txt"), "w") as f:
    f.write("# rocprofv3 --pmc SQ_INSTS_VALU + its ADD/MUL/FMA/TRANS/INT32/INT64/CVT sub-counters (one pass) and --pmc "
            "GRBM_GUI_ACTIVE (its own pass) -- python bench.py --headline-only; per-launch means of the wave-level counts; "
            "clock = GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration (median over the launches; reads high on dispatches "
            "shorter than ~0.3 ms per the guide)\n")
    for stage in names.values():
        e = tj["C4"].setdefault(stage, {})
        if stage in mix:
            # a kernel with two template instances (render_fwd tracking / forward-only): the launches are pooled
            e["valu_mix"] = {c: int(sum(v) / len(v)) for c, v in mix[stage].items()}
        if stage in clk:
            v = sorted(clk[stage])
            e["clock_GHz_pmc"] = round(v[len(v) // 2], 3)
        if stage in mix or stage in clk:
            f.write(f"{stage}: clock {e.get('clock_GHz_pmc')} GHz; " + ", ".join(f"{c}={n}" for c, n in sorted(e.get("valu_mix", {}).items())) + "\n")
lib_abi = int([l for l in open(os.path.join(os.path.dirname(here), "include", "gsr.h")) if "define GSR_ABI_VERSION" in l][0].split()[-1])
tj.setdefault("_stamp", {})["C4"] = {"source_sha16": bench.source_stamp(), "abi": lib_abi, "label": label}
json.dump(tj, open(tpath, "w"), indent=1)
print(open(os.path.join(dst, "pmc_SQ_render_summary.txt")).read())
