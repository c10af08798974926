"""Launch-by-launch table of the binning of one C4 frame: what each launch has to move against what it takes
(VERDICT r04 #7).

    python tools/binning_table.py profiles/r05/rocprofv3_kernel_stats_bench_c4.csv [P V R tiles] > profiles/r05/binning_launch_table.md

Bytes = what the launch must read + write for THIS frame (two-level culled binning: P Gaussians, V visible, R instances);
floor = bytes / 6.2 TB/s (the measured streaming ceiling of an elementwise kernel on this part, bench.py
hbm_elementwise_kernel_GBs), never less than LAT us -- the duration of this pipeline's emptiest launches (a row scan of
256 blocks that moves 1 MB takes 5.1 us: launch ramp + one dependent trip to memory + tail), i.e. what a dependent
launch costs on this part whatever it moves.
"""
import csv
import sys

path = sys.argv[1]
P, V, R, T = (int(x) for x in sys.argv[2:6]) if len(sys.argv) >= 6 else (6_000_000, 3_880_905, 8_192_108, 8160)
STREAM = 6.2e12
LAT = 5.0
rows = {r["Name"]: (float(r["AverageNs"]) / 1e3, int(r["Calls"])) for r in csv.DictReader(open(path))}


def us(*needles):
    for name, (t, _) in rows.items():
        if all(n in name for n in needles):
            return t
    return float("nan")


SB = 4096                       # keys per sort block
nb_depth = (V + SB - 1) // SB
nb_tile = (R + SB - 1) // SB + 128
nblk = (P + 255) // 256
launches = [
    # (launch, kernel needles, count per frame, bytes)
    ("scan of the block totals (R, V, big list)", ("scan_block_sums",), 1, 3 * 8 * nblk),
    ("compact_visible", ("compact_visible",), 1, 16 * P + 8 * nblk + 12 * V + 4 * P),
    ("depth sort: histogram (8 bits) x3", ("radix_hist_kernel<unsigned int, 8",), 3, 4 * V + 4 * 256 * nb_depth),
    ("depth sort: row scan x3", ("radix_rowscan",), 3, 2 * 4 * 256 * nb_depth),
    ("depth sort: scatter (key + 8-byte payload) x3", ("radix_scatter_kernel<unsigned int, HIP_vector_type<unsigned int, 2u>, 8",), 3,
     2 * 12 * V + 4 * 256 * nb_depth),
    ("count_tiles", ("count_tiles",), 1, 8 * V + 4 * ((V + 255) // 256)),
    ("scan of the emission block totals", ("scan_block_sums",), 1, 8 * ((V + 255) // 256)),
    ("emit_instances", ("emit_instances",), 1, 8 * V + 8 * R),
    ("tile sort: histogram (6 bits)", ("radix_hist_kernel<unsigned int, 6",), 1, 4 * R + 4 * 64 * nb_tile),
    ("tile sort: row scan (6 bits)", ("radix_rowscan",), 1, 2 * 4 * 64 * nb_tile),
    ("tile sort: scatter (6 bits)", ("radix_scatter_kernel<unsigned int, unsigned int, 6",), 1, 2 * 8 * R),
    ("tile sort: histogram (7 bits, segmented)", ("radix_hist_kernel<unsigned int, 7",), 1, 4 * R + 4 * 128 * nb_tile),
    ("tile sort: row scan (7 bits) + per-tile runs", ("radix_rowscan",), 1, 2 * 4 * 128 * nb_tile + 8 * T),
    ("tile sort: scatter (7 bits, segmented)", ("radix_scatter_kernel<unsigned int, unsigned int, 7",), 1, 2 * 8 * R),
    ("tile ranges + tile order (one block)", ("ranges_and_order_from_sort",), 1, 16 * T + 4 * T),
]
print(f"# Binning of one C4 frame, launch by launch (P = {P}, V = {V}, R = {R}, {T} tiles; culled two-level binning)")
print(f"# kernel times: rocprofv3 --kernel-trace --stats means of `{path}`; floor = max(bytes / 6.2 TB/s, {LAT:.0f} us per dependent launch)")
print("| launch | per frame | us each | MB each | GB/s | floor us | x floor |")
print("|---|---|---|---|---|---|---|")
tot_t = tot_floor = tot_bw = tot_b = 0.0
for name, needles, n, b in launches:
    t = us(*needles)
    floor = max(b / STREAM * 1e6, LAT)
    print(f"| {name} | {n} | {t:.1f} | {b / 1e6:.1f} | {b / t / 1e3:.0f} | {floor:.1f} | {t / floor:.2f} |")
    tot_t += n * t; tot_floor += n * floor; tot_bw += n * b / STREAM * 1e6; tot_b += n * b
n_launch = sum(n for _, _, n, _ in launches)
print(f"| **sum** | {n_launch} | {tot_t:.0f} | {tot_b / 1e6:.0f} | {tot_b / tot_t / 1e3:.0f} | {tot_floor:.0f} | {tot_t / tot_floor:.2f} |")
print()
print(f"Bytes alone at 6.2 TB/s: {tot_bw:.0f} us; with {LAT:.0f} us for each of the {n_launch} dependent launches that move less than "
      f"{LAT * STREAM / 1e12:.0f} MB: {tot_floor:.0f} us; measured {tot_t:.0f} us (sum of kernel durations: the gaps between launches are "
      "on top of it).")
