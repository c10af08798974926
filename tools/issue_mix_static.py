"""Static VALU instruction mix of the compositing kernels' loops, for the issue-roof model of bench.py.

    python tools/issue_mix_static.py [out.json]        (runs hipcc --cuda-device-only -S: no GPU needed)

The hardware counters split a kernel's VALU instructions into ADD / MUL / FMA (f32), TRANS (f32), INT32, INT64, CVT and
the rest (SQ_INSTS_VALU minus those).  The rest is a mix of full-rate moves and of the half-rate class measured by
tools/ubench/valu_rate.hip (v_cmp / v_cndmask / v_min / v_max / v_med3 / v_readlane / DPP adds: 4.3-4.7 cycles per wave64
instruction per SIMD at 8 waves per SIMD, against 2.4 for v_fma / v_mov and 8.2 for v_exp / v_rcp:
profiles/r02/valu_rate.txt).  Which of the two a kernel's "rest" is made of cannot be read from a counter; this tool reads
it from the disassembly: every VALU instruction inside a loop of the kernel (an instruction inside n nested loops counts
n times -- the walk loops sit inside the round loop) is put into one of the cost classes below and the average cost of the
instructions the counters do NOT name is written out per kernel.  bench.py prices the counted classes at their measured
rates and the rest at that average.
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mvs_gaussian_splatting_amd", "csrc")
# cycles per wave64 instruction per SIMD at 8 resident waves (profiles/r02/valu_rate.txt)
COST = {"full": 2.4, "half": 4.5, "trans": 8.2}
FLAGS = {"render.hip": ["-fno-slp-vectorize"], "preprocess.hip": ["-ffp-contract=off"]}
KERNELS = {"render.hip": ["render_fwd_kernel", "render_bwd_kernel"],
           "preprocess.hip": ["preprocess_fwd_kernel", "preprocess_bwd_kernel"]}

TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")
F32_COUNTED = ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_mac_f32", "v_mad_f32",
               "v_fmaak_f32", "v_fmamk_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32", "v_mul_legacy_f32")
INT_COUNTED = ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_add_co", "v_addc_co", "v_sub_co", "v_subb_co", "v_and_b32",
               "v_or_b32", "v_xor_b32", "v_not_b32", "v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_bfe_", "v_bfi_",
               "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_u32_u24", "v_mad_u32_u24", "v_mad_u64_u32", "v_lshl_add_u32",
               "v_add_lshl_u32", "v_lshl_or_b32", "v_and_or_b32", "v_or3_b32", "v_add3_u32", "v_xad_u32", "v_bcnt_",
               "v_mbcnt_", "v_ffbl_", "v_ffbh_", "v_lshlrev_b64", "v_lshrrev_b64", "v_mul_i32_i24", "v_sad_", "v_alignbit",
               "v_perm_b32", "v_add_nc_u32", "v_sub_nc_u32", "v_mad_i32_i24")
HALF = ("v_cmp", "v_cndmask", "v_min", "v_max", "v_med3", "v_readlane", "v_readfirstlane", "v_writelane", "v_permlane",
        "v_swap", "v_cmpx")


def classify(op: str) -> str:
    """-> one of: trans | f32 | int | cvt (named by the counters) | rest_half | rest_full (priced from this table)."""
    dpp = op.endswith(("_dpp", "_sdwa")) or "_dpp" in op
    base = op.replace("_e32", "").replace("_e64", "")
    if base.startswith(TRANS):
        return "trans"
    if dpp:
        return "rest_half"             # DPP adds / moves run at the half rate whatever the opcode
    if base.startswith("v_cvt_"):
        return "cvt"
    if base.startswith(F32_COUNTED):
        return "f32"
    if base.startswith(HALF):
        return "rest_half"
    if base.startswith(INT_COUNTED):
        return "int"
    return "rest_full"                 # v_mov_b32, v_accvgpr_*, v_nop, v_ldexp, v_frexp, v_fract, ... : full rate


def device_asm(src: str) -> str:
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "--cuda-device-only", "-S",
               os.path.join(CSRC, src), "-o", out] + FLAGS.get(src, [])
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        return open(out).read()


def kernels_of(asm: str):
    cur, body = None, {}
    for line in asm.split("\n"):
        m = re.match(r"^(_ZN3gsr\w+):", line)
        if m:
            cur = m.group(1)
            body[cur] = []
        elif cur is not None:
            if line.startswith(".Lfunc_end"):
                cur = None
            else:
                body[cur].append(line)
    return body


def loop_weighted_mix(lines):
    labels, insts = {}, []
    for line in lines:
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        t = line.strip()
        if not t or t.startswith((".", ";", "//")):
            continue
        insts.append(t.split()[0])
    depth = [0] * len(insts)
    for i, op in enumerate(insts):
        pass
    # back edges: a branch to a label at or before it
    raw = [l.strip() for l in lines if l.strip() and not l.strip().startswith((".", ";", "//")) and not re.match(r"^\.LBB", l)]
    for i, t in enumerate(raw):
        op = t.split()[0]
        if op.startswith(("s_cbranch", "s_branch")):
            tgt = t.split()[-1]
            if tgt in labels and labels[tgt] <= i:
                for j in range(labels[tgt], i + 1):
                    depth[j] += 1
    mix = collections.Counter()
    ops = collections.Counter()
    for op, d in zip(insts, depth):
        if op.startswith("v_") and d > 0:
            mix[classify(op)] += d
            ops[op] += d
    return mix, ops


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "valu_static_mix.json")
    res = {"_method": __doc__.strip().split("\n\n")[1].replace("\n", " "), "_cost_cycles": COST}
    for src, names in KERNELS.items():
        for sym, lines in kernels_of(device_asm(src)).items():
            name = next((n for n in names if n in sym), None)
            if name is None:
                continue
            mix, ops = loop_weighted_mix(lines)
            rest = mix["rest_half"] + mix["rest_full"]
            entry = {"symbol": sym, "loop_weighted_valu": dict(mix),
                     "rest_half_share": round(mix["rest_half"] / rest, 4) if rest else 0.0,
                     "rest_cost_cycles": round((mix["rest_half"] * COST["half"] + mix["rest_full"] * COST["full"]) / rest, 3) if rest else COST["full"],
                     "top_rest_opcodes": [f"{o} x{n}" for o, n in ops.most_common(40) if classify(o).startswith("rest")][:12]}
            key = name.replace("_kernel", "")
            # a kernel with several template instances: keep the one with the most loop instructions (the tracking variant)
            if key not in res or sum(mix.values()) > sum(res[key]["loop_weighted_valu"].values()):
                res[key] = entry
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res.items():
        if not k.startswith("_"):
            print(k, v["loop_weighted_valu"], "rest cost", v["rest_cost_cycles"], "cycles; half share", v["rest_half_share"])
            print("   ", ", ".join(v["top_rest_opcodes"]))


if __name__ == "__main__":
    main()
