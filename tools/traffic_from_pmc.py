"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into profiles/traffic.json entries.

FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes for wide
coalesced streaming reads (MI355X_MICROARCH.md, HBM section), so the read side is reported both raw and doubled.
    python tools/traffic_from_pmc.py <fetch_dir> <write_dir> <config> [out.json]
"""
import collections, csv, glob, json, os, sys


def per_kernel(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].split("::")[-1]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
cfg = sys.argv[3]
out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
names = {"preprocess_fwd_kernel": "preprocess_fwd", "scan_block_sums_kernel": "scan_block_sums",
         "duplicate_with_keys_kernel": "duplicate_with_keys", "identify_tile_ranges_kernel": "identify_tile_ranges",
         "render_fwd_kernel": "render_fwd", "render_bwd_kernel": "render_bwd", "preprocess_bwd_kernel": "preprocess_bwd"}
res = {}
for k, stage in names.items():
    if k in fetch or k in write:
        rd, wr = fetch.get(k, 0.0) * 1024, write.get(k, 0.0) * 1024
        # the compositing kernels read one 64-byte GeomRec line per instance (64-B requests: counted in full);
        # the streaming kernels read with wide coalesced accesses (128-B requests counted as 64 B: doubled)
        gather = stage in ("render_fwd", "render_bwd")
        res[stage] = {"hbm_bytes_per_launch": int((1 if gather else 2) * rd + wr), "read_bytes_raw_FETCH_SIZE": int(rd),
                      "read_bytes_x2_gfx950_correction": int(2 * rd), "write_bytes_WRITE_SIZE": int(wr),
                      "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes, per-launch mean; "
                                "read side doubled per MI355X_MICROARCH.md (128-B requests tallied at 64 B); gather-style "
                                "64-B reads are uncalibrated, so the true value lies between raw and doubled"}
sort_rd = sum(fetch.get(k, 0) for k in ("radix_hist_kernel", "radix_rowscan_kernel", "radix_scatter_kernel")) * 1024
sort_wr = sum(write.get(k, 0) for k in ("radix_hist_kernel", "radix_rowscan_kernel", "radix_scatter_kernel")) * 1024
res["radix_sort_per_pass"] = {"hbm_bytes_per_launch": int(2 * sort_rd + sort_wr), "read_bytes_raw_FETCH_SIZE": int(sort_rd),
                              "write_bytes_WRITE_SIZE": int(sort_wr), "method": "sum of hist+rowscan+scatter means (one pass)"}
data = json.load(open(out)) if os.path.exists(out) else {}
data[cfg] = res
json.dump(data, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
