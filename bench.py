#!/usr/bin/env python3
"""Benchmark of the hot path (BASELINE.json): forward Mpixels/s and train-step ms at 1920x1080 on 6 M
synthetic Gaussians, one view per GPU (weak scaling by independent views; the only collective is the
4-float loss all-reduce, RCCL over xGMI).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C4|C3|C2] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Both forms work for N > 1.  Without a torchrun environment (WORLD_SIZE unset) `python bench.py --gpus N` is the LAUNCHER:
it starts `python -m torch.distributed.run ... bench.py <same arguments>` as a child process -- before anything in this
process has touched the GPU --, hands rank 0's single JSON line on to its own stdout and exits with the children's
worst exit code (`launch_ranks`).

Two HEADLINE regions, each EXACTLY K steps bracketed by barrier + torch.cuda.synchronize() and reduced with
MAX over ranks:  (A) forward-only render under no_grad  -> `value` (Mpixels/s, whole job);
                 (B) train step = render + L1 + backward + densification stats (+ loss all-reduce)
                     -> `ms_per_step`.
Rank r renders view r of the eight C5 orbit views in them (one view per GPU: weak scaling; N = 8 is SURVEY §8e's partition).
Further regions of the same protocol, reported as extra fields of the same line (never as `value`):
  cold_*                 the first K steps after the W warm-ups, before the untimed pre-warm steps (GPU clocks still ramping);
  *_readback_every_frame the same steps with the count read-back in every frame (GSR_SYNC_FREE=0, upstream's behaviour);
  train_step_ms_l1_dssim the train step with the reference's real loss, 0.8 L1 + 0.2 (1 - SSIM) (train.py:99-101);
  c5_eight_views         the EIGHT orbit views partitioned {r, r + N, ...} over the N ranks: every rank renders
                         its 8 / N views per round -- at N = 1 the rotating-camera loop (the instance count changes from
                         frame to frame: the capacity-based forward is exercised under a varying R);
  per_rank_*             every rank's own forward / train ms of the headline regions (gathered);
  morton_layout_*        the same view of the same cloud after layout.reorder_gaussians_ (the Gaussians stored along a Morton
                         curve: an optional step for callers that own their model; the headline stays on the cloud as generated).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling


def algorithmic_bytes(P, M, R, W, H, passes):
    """SURVEY.md §8(d) model v1, bytes per launch of each stage (every Gaussian priced as if it were visible)."""
    T = ((W + 15) // 16) * ((H + 15) // 16)
    return {
        "preprocess_fwd": P * (44 + 12 * M + 75),
        "scan_block_sums": 8 * P,
        "duplicate_with_keys": 20 * P + 12 * R,
        "radix_sort": 8 * R + passes * 24 * R,
        "identify_tile_ranges": 8 * R + 8 * T,
        "render_fwd": 40 * R + 20 * W * H,
        "render_bwd": 40 * R + 80 * R + 20 * W * H,
        "preprocess_bwd": P * (175 + 24 * M),
    }


def bytes_really_moved(P, M, V, R, W, H):
    """Model v2 for the two per-Gaussian stages: the bytes the kernels have to move for THIS frame.
    A Gaussian outside the frustum costs its 44 bytes of position / scale / rotation / opacity and nothing else: its
    12 M-byte SH row is never read and its 75 bytes of per-Gaussian state are never written (V = visible Gaussians, a
    lower bound of the rows read: an on-screen centre whose footprint rounds to zero tiles has its row read too).
    Backward: every Gaussian gets its gradient rows written (56 + 12 M bytes: the operator returns dense tensors) and
    its radius read; a visible one also reads its record, slot, position / scale / rotation, SH row (for d rgb / d dir)
    and its 37-byte instance rows (36-byte row + flag byte)."""
    return {
        "preprocess_fwd": 44 * P + (12 * M + 75) * V,
        "preprocess_bwd": (56 + 12 * M + 4) * P + (64 + 4 + 40 + 12 * M) * V + 37 * R,
    }


def source_stamp():
    """sha256[:16] over the kernel sources + ABI header: ties profiles/traffic.json (PMC passes taken at one state of
    the kernels) to the library this run measures; .git does not travel to the GPU box, the sources do."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "mvs_gaussian_splatting_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(ROOT, "mvs_gaussian_splatting_amd", "csrc", "*.h")) +
                   [os.path.join(ROOT, "include", "gsr.h")])
    for f in files:
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def host_cores():
    """(cores this process may use by affinity mask and cgroup CPU quota, os.cpu_count() of the host)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p))))
    except Exception:
        pass
    return max(1, n), os.cpu_count() or 1


def usable_cores():
    """Threads the CPU baseline runs on: the usable cores, capped at 16 (a 1-GPU box's share; os.cpu_count() on a big
    host oversubscribes the OpenMP pool by an order of magnitude).  The uncapped numbers are reported beside it."""
    return min(host_cores()[0], 16)


def cpu_baseline(cfg, seed, budget_tiles=1024, bwd_tiles=96):
    """Pure-PyTorch CPU oracle (float32) timed on the host cores over a bounded sample of the same workload:
    forward = full preprocess + full binning + compositing of every k-th tile, extrapolated;
    train step = the same under autograd + L1 + backward, on a smaller tile sample (the autograd graph of the
    compositing holds every per-chunk intermediate), extrapolated the same way."""
    from oracle import RasterSettings, preprocess_ref, bin_ref, render_tiles_ref
    from mvs_gaussian_splatting_amd.synthetic import make_scene
    threads = usable_cores()
    usable, host = host_cores()
    torch.set_num_threads(threads)
    print(f"[bench] cpu_baseline: oracle on {torch.get_num_threads()} threads ...", file=sys.stderr, flush=True)
    model, cam, bg, target = make_scene(cfg, seed=seed)
    st = RasterSettings(cam.image_height, cam.image_width, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), bg, 1.0,
                        cam.world_view_transform, cam.full_proj_transform, cfg.sh_degree, cam.camera_center)
    W, H = cfg.width, cfg.height
    with torch.no_grad():
        t0 = time.perf_counter()
        pre = preprocess_ref(model.get_xyz, model.get_opacity, st, shs=model.get_features,
                             scales=model.get_scaling, rotations=model.get_rotation)
        t1 = time.perf_counter()
        print(f"[bench] cpu_baseline: preprocess {t1 - t0:.1f}s", file=sys.stderr, flush=True)
        keys, plist, ranges = bin_ref(pre)
        t2 = time.perf_counter()
        print(f"[bench] cpu_baseline: binning {t2 - t1:.1f}s", file=sys.stderr, flush=True)
        gx, gy = pre["grid"]
        n_tiles = gx * gy
        stride = max(1, int(math.ceil(n_tiles / budget_tiles)))
        tiles = list(range(stride // 2, n_tiles, stride))
        t3 = time.perf_counter()
        render_tiles_ref(pre, plist, ranges, st, tiles=tiles)
        t4 = time.perf_counter()
    scale = n_tiles / len(tiles)
    fwd_s = (t1 - t0) + (t2 - t1) + (t4 - t3) * scale
    print(f"[bench] cpu_baseline: forward sample {t4 - t3:.1f}s; train-step sample ...", file=sys.stderr, flush=True)
    # ---- train step: forward under autograd + L1 + backward (fp32), compositing restricted to a tile sample
    del pre
    leaves = [p.detach().clone().requires_grad_(True) for p in
              (model._xyz, model._features_dc, model._features_rest, model._scaling, model._rotation, model._opacity)]
    stride_b = max(1, int(math.ceil(n_tiles / bwd_tiles)))
    tiles_b = list(range(stride_b // 2, n_tiles, stride_b))
    b0 = time.perf_counter()
    pre = preprocess_ref(leaves[0], torch.sigmoid(leaves[5]), st, shs=torch.cat((leaves[1], leaves[2]), dim=1),
                         scales=torch.exp(leaves[3]), rotations=torch.nn.functional.normalize(leaves[4]))
    b1 = time.perf_counter()
    # backward in two legs so that only the tile-dependent one is extrapolated: compositing (down to the per-Gaussian
    # screen-space quantities), then preprocess (every Gaussian, independent of the tile sample).  The compositing leg is
    # timed twice and the faster pass kept: its autograd backward scatters into V-sized tensors once per tile and has
    # been seen to take 1.3 s or 15 s for the same 96 tiles depending on what else the host is doing (x85 extrapolated)
    mids = [pre[k] for k in ("v_xy", "v_conic", "v_opacity", "v_rgb") if pre[k].requires_grad]
    best = None
    for _ in range(2):
        c0 = time.perf_counter()
        col = render_tiles_ref(pre, plist, ranges, st, tiles=tiles_b)[0]
        c1 = time.perf_counter()
        loss = (col - target).abs().mean()
        g_mid = torch.autograd.grad(loss, mids)
        c2 = time.perf_counter()
        if best is None or (c2 - c0) < (best[1] - best[0]) + (best[2] - best[1]):
            best = (c0, c1, c2)
    b2 = b1 + (best[1] - best[0])
    b3 = b2 + (best[2] - best[1])
    b3_wall = time.perf_counter()
    torch.autograd.backward(mids, g_mid)
    b4 = b3 + (time.perf_counter() - b3_wall)
    scale_b = n_tiles / len(tiles_b)
    step_s = (b1 - b0) + (t2 - t1) + ((b2 - b1) + (b3 - b2)) * scale_b + (b4 - b3)
    return {
        "value": W * H / fwd_s / 1e6, "unit": "Mpixels/s", "cores": torch.get_num_threads(), "kind": "port",
        "host_cores_usable": usable, "host_cores_total": host,
        "sample": (f"pure-PyTorch fp32 oracle, forward: full preprocess ({t1 - t0:.1f}s) + full binning "
                   f"({t2 - t1:.1f}s) + compositing of {len(tiles)}/{n_tiles} tiles ({t4 - t3:.1f}s, x{scale:.0f} "
                   f"extrapolated) -> {fwd_s:.1f}s per {W}x{H} frame; train step: preprocess under autograd "
                   f"({b1 - b0:.1f}s) + binning + compositing of {len(tiles_b)}/{n_tiles} tiles ({b2 - b1:.1f}s) + L1 "
                   f"+ compositing backward ({b3 - b2:.1f}s) [both x{scale_b:.0f}] + preprocess backward "
                   f"({b4 - b3:.1f}s) -> {step_s:.0f}s per step"),
        "fwd_seconds_extrapolated": fwd_s,
        "train_step_seconds_extrapolated": step_s,
        "train_step_ms": step_s * 1e3,
    }

# cycles per wave64 VALU instruction per SIMD at 8 resident waves (tools/ubench/valu_rate.hip, profiles/r02/valu_rate.txt):
# v_fma / v_add / v_mul / v_mov / integer ops 2.4; v_exp / v_rcp / v_rsq / v_log 8.2; the rest (v_cmp, v_cndmask, v_min, v_max,
# v_med3, v_readlane, DPP adds mixed with moves) is priced per kernel by tools/issue_mix_static.py (profiles/valu_static_mix.json)
VALU_CYCLES_FULL, VALU_CYCLES_TRANS = 2.4, 8.2
SIMDS = 1024


def issue_model(valu_mix, rest_cost_cycles, t_s, clock_hz):
    """Instruction-issue roof of a VALU-bound kernel: the SIMD cycles its vector instructions need at their measured issue
    rates, over the SIMD cycles the launch had (1024 SIMDs x clock x duration).
      valu_mix: per-launch wave-level counts from one rocprofv3 --pmc pass (SQ_INSTS_VALU and its ADD / MUL / FMA / TRANS /
      INT32 / INT64 / CVT sub-counters); what the sub-counters do not name is priced at rest_cost_cycles."""
    total = float(valu_mix.get("SQ_INSTS_VALU", 0))
    if total <= 0 or t_s <= 0:
        return None
    full = sum(float(valu_mix.get(k, 0)) for k in ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32",
                                                    "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_CVT"))
    trans = float(valu_mix.get("SQ_INSTS_VALU_TRANS_F32", 0))
    rest = max(0.0, total - full - trans)
    cycles = VALU_CYCLES_FULL * full + VALU_CYCLES_TRANS * trans + rest_cost_cycles * rest
    return {"valu_insts": int(total), "full_rate": int(full), "transcendental": int(trans), "rest": int(rest),
            "rest_cost_cycles": rest_cost_cycles, "valu_issue_cycles": int(cycles), "clock_GHz": round(clock_hz / 1e9, 3),
            "frac_of_issue_roof": round(cycles / (SIMDS * clock_hz * t_s), 4),
            "cycles_per_valu_inst": round(cycles / total, 3),
            "uncertainty": "about +-5 %: the class rates are measured in isolation at 8 waves per SIMD, the clock comes from a "
                           "separate --pmc pass of the same kernel; a value at (or a few per cent above) 1 means the launch "
                           "left no vector issue slot unused"}


def c5_views_of_rank(rank, world, n_views=8):
    """rank r renders views {r, r + N, ...} of the eight."""
    from mvs_gaussian_splatting_amd.dist import views_of_rank
    return views_of_rank(n_views, rank, world)


def c5_summary(per_rank_fwd_s, per_rank_train_s, rounds, world, W, H, n_views=8):
    """Whole-job numbers of the eight-view loop from every rank's wall time over `rounds` rounds (a round = every rank
    renders its 8 / N views once): the job is as slow as its slowest rank."""
    fwd, train = max(per_rank_fwd_s), max(per_rank_train_s)
    per_rank_views = [len(c5_views_of_rank(r, world, n_views)) for r in range(world)]
    return {"views": n_views, "views_per_rank": per_rank_views, "rounds": rounds,
            "fwd_ms_per_round": round(fwd / rounds * 1e3, 3), "train_ms_per_round": round(train / rounds * 1e3, 3),
            "fwd_mpixels_per_s": round(n_views * W * H * rounds / fwd / 1e6, 2),
            "train_mpixels_per_s": round(n_views * W * H * rounds / train / 1e6, 2),
            "per_rank_fwd_ms_per_view": [round(t / rounds / max(v, 1) * 1e3, 3) for t, v in zip(per_rank_fwd_s, per_rank_views)],
            "per_rank_train_ms_per_view": [round(t / rounds / max(v, 1) * 1e3, 3) for t, v in zip(per_rank_train_s, per_rank_views)]}


def visible_gpus():
    """Devices this process would see.  torch.cuda.device_count() reads the count without creating a HIP context on this
    image (torch.cuda.is_available() would: never call that in the launcher)."""
    try:
        return int(torch.cuda.device_count())
    except Exception:  # noqa: BLE001
        return 0


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def pick_result_line(stdout_text):
    """The one result line among whatever the rank processes wrote to stdout: the LAST line that parses as a JSON object
    with a "metric" key (rank 0 prints exactly one; banners of native libraries are not JSON)."""
    found = None
    for ln in stdout_text.splitlines():
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            try:
                if "metric" in json.loads(ln):
                    found = ln
            except ValueError:
                pass
    return found


def launch_ranks(n, argv, script=None, out=None, env=None, timeout=None, gpus_visible=None):
    """`python bench.py --gpus N` without a torchrun environment: start N fresh rank processes and relay rank 0's line.

    The parent must not have initialised the GPU (it has only imported torch): the ranks are CHILD processes of
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>` --
    nothing is exec'ed over a process that holds a HIP context.  Returns the exit code: the launcher's (non-zero when any
    rank failed: torchrun tears the others down), or 1 when every rank exited 0 but no result line arrived.
    With fewer visible devices than ranks (a one-card rehearsal) the ranks share cuda:0 (GSR_BENCH_SHARE_GPU=1): RCCL
    refuses a duplicate device, the ranks agree on gloo for the loss all-reduce, and the line says "valid": false.
    `script` / `out` / `env` / `gpus_visible` are injectable for the CPU test (tests/test_bench_protocol.py)."""
    import subprocess
    script = script or os.path.abspath(__file__)
    out = out if out is not None else sys.stdout
    env = dict(os.environ if env is None else env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE", "ROLE_RANK"):
        env.pop(k, None)
    env["MASTER_ADDR"] = "127.0.0.1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL's peer mappings need it on this pool
    have = visible_gpus() if gpus_visible is None else int(gpus_visible)
    if have < n and env.get("GSR_BENCH_SHARE_GPU") != "1":
        print(f"[bench] launcher: {have} device(s) visible for {n} ranks -- REHEARSAL: the ranks share cuda:0 "
              "(GSR_BENCH_SHARE_GPU=1), RCCL cannot form a communicator over one device, the line will say valid: false",
              file=sys.stderr, flush=True)
        env["GSR_BENCH_SHARE_GPU"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script] + list(argv)
    print("[bench] launcher: " + " ".join(cmd), file=sys.stderr, flush=True)
    try:
        proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, timeout=timeout)
    except subprocess.TimeoutExpired as ex:
        print(f"[bench] launcher: ranks did not finish within {timeout} s", file=sys.stderr, flush=True)
        if ex.stdout:
            sys.stderr.write(ex.stdout.decode(errors="replace"))
        return 124
    text = proc.stdout.decode(errors="replace")
    line = pick_result_line(text)
    for ln in text.splitlines():                                # whatever else reached stdout goes to stderr
        if ln.strip() and ln.strip() != line:
            print(ln, file=sys.stderr)
    if proc.returncode != 0:
        print(f"[bench] launcher: rank processes failed (exit code {proc.returncode})", file=sys.stderr, flush=True)
        return proc.returncode if 0 < proc.returncode < 256 else 1
    if line is None:
        print("[bench] launcher: every rank exited 0 but rank 0 printed no result line", file=sys.stderr, flush=True)
        return 1
    out.write(line + "\n")
    out.flush()
    return 0


def rccl_single_rank_probe(timeout=120):
    """RCCL exercised on whatever this box has: a ONE-rank communicator (backend "nccl") in a fresh child process does the
    path's collective -- all_reduce(SUM) of the 4-float loss vector -- 200 times on cuda:0.  Says that librccl loads, a
    communicator comes up and its kernel runs on gfx950, and what one such collective costs to issue; it says nothing about
    xGMI (there is no peer).  Child process: a hung communicator costs the timeout, not the bench."""
    import subprocess
    code = (
        "import os, json, time, torch, torch.distributed as dist\n"
        "os.environ.update(MASTER_ADDR='127.0.0.1', RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')\n"
        "dev = torch.device('cuda', 0); torch.cuda.set_device(dev)\n"
        "dist.init_process_group(backend='nccl', device_id=dev)\n"
        "v = torch.tensor([0.25, 0.25, 1.0, 0.0], device=dev)\n"
        "for _ in range(20): dist.all_reduce(v)\n"
        "torch.cuda.synchronize(dev); t0 = time.perf_counter()\n"
        "for _ in range(200): dist.all_reduce(v)\n"
        "torch.cuda.synchronize(dev); dt = (time.perf_counter() - t0) / 200\n"
        "ok = bool(torch.equal(v.cpu(), torch.tensor([0.25, 0.25, 1.0, 0.0])))\n"
        "ver = '.'.join(str(x) for x in torch.cuda.nccl.version())\n"
        "print(json.dumps({'ok': ok, 'ranks': dist.get_world_size(), 'backend': dist.get_backend(), 'rccl_version': ver, "
        "'allreduce_16B_us': round(dt * 1e6, 2)}))\n"
        "dist.destroy_process_group()\n")
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    env["MASTER_PORT"] = str(free_port())
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    except subprocess.TimeoutExpired:
        return {"ok": False, "why": f"no answer within {timeout} s"}
    for ln in reversed(r.stdout.decode(errors="replace").splitlines()):
        if ln.startswith("{"):
            try:
                d = json.loads(ln)
                d["what"] = "one-rank RCCL communicator on cuda:0: all_reduce(SUM) of the 4-float loss vector, 200 calls"
                return d
            except ValueError:
                pass
    return {"ok": False, "why": (r.stderr.decode(errors="replace").strip().splitlines() or ["no output"])[-1][:300],
            "returncode": r.returncode}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C4", choices=["C2", "C3", "C4"])
    ap.add_argument("--gaussians", type=int, default=None, help="override P (debug only; marks the line invalid)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--unfused", action="store_true",
                    help="feed the operator through the reference getters (cat/exp/normalize/sigmoid in torch)")
    ap.add_argument("--fuse-stats", action="store_true",
                    help="take the densification statistics in preprocess_bwd's epilogue instead of the stand-alone kernel")
    ap.add_argument("--no-prewarm", action="store_true",
                    help="skip the extra untimed steps in front of each headline region (GPU clock ramp)")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the two headline regions + the stage-timer pass (what the rocprofv3 passes run: every launch "
                         "of a kernel then belongs to the headline view)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-rccl-probe", action="store_true", help="skip the one-rank RCCL communicator probe (N = 1 only)")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # the plain command form: this process becomes the launcher of N rank processes (it has not touched the GPU)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    rccl_probe = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.headline_only and not args.no_rccl_probe:
        # a child process, started BEFORE this process creates its HIP context (a process that holds the GPU starts nothing)
        rccl_probe = rccl_single_rank_probe()
    # stdout carries exactly one line, the JSON result: native libraries (the RCCL / gloo banners) write to fd 1 too,
    # so fd 1 is pointed at stderr for the run and the result goes to a duplicate of the original stdout
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch.distributed as dist
    from mvs_gaussian_splatting_amd import render, l1_loss, add_densification_stats, _lib
    from mvs_gaussian_splatting_amd.synthetic import CONFIGS, make_scene, PipelineParams

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}: launch as `python bench.py --gpus N` or as "
                         "`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the product path has no CPU fallback)")
    # one process per GPU; GSR_BENCH_SHARE_GPU=1 (rehearsal on a 1-GPU box, gloo backend) lets ranks share cuda:0
    share = os.environ.get("GSR_BENCH_SHARE_GPU") == "1"
    dev = torch.device("cuda", 0 if share else local_rank)
    torch.cuda.set_device(dev)
    backend = os.environ.get("GSR_BENCH_BACKEND", "nccl")     # "nccl" is RCCL on ROCm
    backend_note = backend
    from mvs_gaussian_splatting_amd.dist import negotiate_collectives
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # one group, two backends: RCCL for device tensors, gloo for host tensors
        dist.init_process_group(backend="cpu:gloo,cuda:nccl" if backend == "nccl" else backend)

    def _probe():
        probe = torch.ones(1, device=dev)
        dist.all_reduce(probe)
        torch.cuda.synchronize(dev)
        return float(probe.item())

    def _agree(flag):
        t = torch.tensor([flag], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)          # host tensor -> gloo
        return int(t.item())

    # the probe / agree / fall-back protocol lives in mvs_gaussian_splatting_amd/dist.py (unit-tested with a fake
    # process group): a run that had to fall back to gloo still reports its timings as diagnostics but is marked invalid
    plan = negotiate_collectives(world, backend, _probe, _agree)
    cpu_collectives, rccl_ranks, rccl_failed, backend_note = (plan.cpu_collectives, plan.rccl_ranks, plan.rccl_failed,
                                                              plan.backend_note)
    if rccl_failed:
        print(f"[bench] rank {rank}: RCCL unavailable on this or another rank ({plan.why!r}); every rank switches to "
              "gloo for the 16-byte loss all-reduce and the line is marked invalid", file=sys.stderr, flush=True)

    def all_reduce(t, op):
        if cpu_collectives:
            h = t.detach().cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=op)

    cfg = CONFIGS[args.config]
    P = args.gaussians or cfg.P
    # every rank: same Gaussians (replicated parameters), its own view  -> weak scaling over views
    model, cam, bg, target = make_scene(cfg, seed=args.seed, P=P, view=rank, n_views=max(world, 8))
    model.to(dev)
    cam.to(dev)
    bg, target = bg.to(dev), target.to(dev)
    for p in model.parameters():
        p.requires_grad_(True)
    pipe = PipelineParams()
    pipe.fuse_activations = not args.unfused
    # densification statistics: the stand-alone kernel of add_densification_stats (45 us).  The epilogue fused into
    # preprocess_bwd (GsrGrads.stats_*, --fuse-stats) is slower on this part: the three extra read-modify-write streams
    # cost the bandwidth-bound kernel 55-70 us (profiles/r03/ab_densify_stats.txt)
    pipe.fuse_densify_stats = bool(args.fuse_stats)
    W, H = cfg.width, cfg.height
    M = (cfg.sh_degree + 1) ** 2

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            if cpu_collectives:
                dist.all_reduce(torch.zeros(1))
            else:
                dist.barrier(device_ids=[dev.index])
        torch.cuda.synchronize(dev)

    def fwd_step(c=None):
        with torch.no_grad():
            return render(c or cam, model, pipe, bg)

    loss_vec = torch.zeros(4, device=dev)
    from mvs_gaussian_splatting_amd import l1_dssim_loss

    def train_step(c=None, tgt=None, dssim=False):
        for p in model.parameters():
            p.grad = None
        pkg = render(c or cam, model, pipe, bg)
        img, gt = pkg["render"], (target if tgt is None else tgt)
        loss = l1_dssim_loss(img, gt, 0.2) if dssim else l1_loss(img, gt)     # train.py:99-101 (lambda_dssim = 0.2) / C4's L1
        loss.backward()
        add_densification_stats(model, pkg["viewspace_points"], pkg["radii"])
        if world > 1:     # the path's only collective: [loss_sum, l1_sum, n_views, pad]
            loss_vec[0] = loss.detach(); loss_vec[1] = loss.detach(); loss_vec[2] = 1.0
            all_reduce(loss_vec, dist.ReduceOp.SUM)
        return pkg, loss

    # the eight C5 orbit views partitioned {r, r + N, ...}
    from mvs_gaussian_splatting_amd.synthetic import orbit_camera
    my_views = c5_views_of_rank(rank, world)
    c5_cams = [orbit_camera(v, 8, W, H, cfg.fx, cfg.fy, device=dev) for v in my_views]
    c5_targets = [torch.rand(3, H, W, generator=torch.Generator().manual_seed(1 + v)).to(dev) for v in my_views]

    for _ in range(args.warmup):
        fwd_step()
        pkg, loss = train_step()
    radii = pkg["radii"]
    visible = int((radii > 0).sum())

    def timed(fn, k):
        """-> (wall seconds of the K steps, MAX over ranks; per-step device ms of this rank from events recorded on
        the stream the kernels run on -- the operator launches on torch's current stream)."""
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(k + 1)]
        barrier()
        t0 = time.perf_counter()
        evs[0].record()
        for i in range(k):
            fn()
            evs[i + 1].record()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            all_reduce(t, dist.ReduceOp.MAX)
            dt = float(t.item())
        in_order = [evs[i].elapsed_time(evs[i + 1]) for i in range(k)]
        return dt, (sorted(in_order), in_order)

    def pct(v, q):
        return round(v[min(len(v) - 1, int(q * len(v)))], 4)

    def copy_ceiling():
        """Measured device-to-device copy rate (bytes read + bytes written per second): the practical HBM ceiling
        the streaming stages are judged against, next to the 8 TB/s vendor peak."""
        n = 1 << 28                                             # 1 GiB of float32
        a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
        b = torch.empty_like(a)
        for _ in range(3):
            b.copy_(a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            b.copy_(a)
        e1.record()
        torch.cuda.synchronize(dev)
        rate_copy = 10 * 2 * 4 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9
        # the runtime's copy is not the fastest way to move bytes on this part: an elementwise KERNEL (one 16-byte access
        # per lane) streams ~20 % faster (tools/ubench/hbm_stream.hip: 6.2-6.6 TB/s); reported next to it
        for _ in range(3):
            torch.add(a, 1.0, out=b)
        e0.record()
        for _ in range(10):
            torch.add(a, 1.0, out=b)
        e1.record()
        torch.cuda.synchronize(dev)
        copy_ceiling.kernel_rate = 10 * 2 * 4 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9
        return rate_copy

    K = args.steps
    # The W warm-up steps above take ~10 ms in all: not long enough for the GPU to leave its idle clocks (per-frame time
    # falls from 1.03 to 0.91 ms over the first ~30 frames of a cold region, profiles/r03/clock_ramp.txt; the host issues
    # a frame in 0.13 ms and is never what the GPU waits for).  A sustained run is what the metric describes, so each
    # headline region is preceded by a fixed number of further UNTIMED steps of its own kind; they are reported in the
    # line, and the first K of them are timed as the COLD region and reported beside the headline.
    prewarm = {"fwd": 0 if args.no_prewarm else 60, "train": 0 if args.no_prewarm else 25}
    cold = {}
    if not args.no_prewarm:
        t_c, _ = timed(fwd_step, K)
        cold["cold_fwd_ms"] = round(t_c / K * 1e3, 3)
        for _ in range(max(0, prewarm["fwd"] - K)):
            fwd_step()
    # Headline regions: EXACTLY K steps each, nothing but the steps between the barriers.
    t_fwd, (fwd_steps, fwd_order) = timed(fwd_step, K)
    if not args.no_prewarm:
        t_c, _ = timed(train_step, K)
        cold["cold_train_ms"] = round(t_c / K * 1e3, 3)
        for _ in range(max(0, prewarm["train"] - K)):
            train_step()
    t_train, (train_steps, train_order) = timed(train_step, K)
    # every rank's own device time of the two headline regions (sum of its per-step event intervals), gathered
    per_rank = torch.zeros(2 * world, dtype=torch.float64, device=dev)
    per_rank[2 * rank] = sum(fwd_order) / K
    per_rank[2 * rank + 1] = sum(train_order) / K
    if world > 1:
        all_reduce(per_rank, dist.ReduceOp.SUM)
    per_rank = per_rank.cpu().tolist()
    # Per-kernel times for the roofline objects: the same K + K steps once more with the library's stage timers on (a HIP
    # event pair around every stage, on the stream the kernels run on).  Kept out of the headline regions because the
    # 14-16 event records per frame cost the frame 5-10 % (they serialise the stream: `profiled_*_ms` below shows it).
    prof = _lib.StageProfile()
    with prof:
        t_fwd_prof, _ = timed(fwd_step, K)
        stages_fwd = prof.collect()
        t_train_prof, _ = timed(train_step, K)
        stages_train = prof.collect()
    prof.close()
    # the reference's real training loss (train.py:99-101): same step, fused L1 + D-SSIM kernel pair instead of L1
    extras = not args.headline_only
    t_dssim = float("nan")
    if extras:
        for _ in range(3):
            train_step(dssim=True)
        t_dssim, _ = timed(lambda: train_step(dssim=True), K)
    # upstream's host synchronisation: the count read back in every frame (GSR_SYNC_FREE=0)
    from mvs_gaussian_splatting_amd import rasterizer as _rz
    sync_mode = _rz.sync_free_mode()
    _rz.synchronize_counts()
    readback = {}
    if extras and sync_mode != _rz.SYNC_OFF:
        _rz.set_sync_free(False)
        for _ in range(3):
            fwd_step(); train_step()
        t_rb_f, _ = timed(fwd_step, K)
        t_rb_t, _ = timed(train_step, K)
        readback = {"fwd_ms_readback_every_frame": round(t_rb_f / K * 1e3, 3),
                    "train_ms_readback_every_frame": round(t_rb_t / K * 1e3, 3)}
        _rz.set_sync_free(sync_mode)
    # SURVEY §8e's partition of the EIGHT views: a round = every rank renders its 8 / N views once (at N = 1 the rotating
    # camera: R changes from frame to frame, so the capacity-based forward runs under a varying instance count)
    rounds = max(2, K // 4)

    def c5_fwd_round():
        for c in c5_cams:
            fwd_step(c)

    def c5_train_round():
        for c, tg in zip(c5_cams, c5_targets):
            train_step(c, tg)

    c5 = None
    if extras:
        c5_fwd_round(); c5_train_round()
        t_c5_f, _ = timed(c5_fwd_round, rounds)
        t_c5_t, _ = timed(c5_train_round, rounds)
        c5 = c5_summary([t_c5_f] * world, [t_c5_t] * world, rounds, world, W, H)     # timed() already took the MAX over ranks
        c5["reissued_frames"] = _rz.reissued_frames(dev, P, W, H)
    # The literal drop-in (INTEGRATION.md option A: the reference's own render() feeding the operator through the
    # getters of scene/gaussian_model.py:151-183 -- torch cat / exp / normalize / sigmoid and their autograd) next to
    # the fused raw-parameter path the headline numbers use; a shorter timed region of the same protocol.
    unfused = None
    if extras and pipe.fuse_activations and world == 1:
        pipe.fuse_activations = False
        fuse_stats_was, pipe.fuse_densify_stats = pipe.fuse_densify_stats, False
        k2 = max(3, K // 4)
        for _ in range(2):
            fwd_step(); train_step()
        t_uf, _ = timed(fwd_step, k2)
        t_ut, _ = timed(train_step, k2)
        unfused = {"unfused_fwd_ms": round(t_uf / k2 * 1e3, 3), "unfused_train_ms": round(t_ut / k2 * 1e3, 3),
                   "unfused_steps": k2}
        pipe.fuse_activations = True
        pipe.fuse_densify_stats = fuse_stats_was

    # instances of this rank's view: read back from the stage the operator itself ran
    from mvs_gaussian_splatting_amd.rasterizer import frame_counts
    pkg, _ = train_step()
    R = int(frame_counts(pkg["render"])[0]) if pkg["render"].grad_fn is not None else 0

    # every rank's instance count: the views differ (8.2 M .. 8.6 M instances at C4), and the job is as slow as its slowest rank
    per_rank_R = torch.zeros(world, dtype=torch.float64, device=dev)
    per_rank_R[rank] = float(R)
    if world > 1:
        all_reduce(per_rank_R, dist.ReduceOp.SUM)
    per_rank_R = [int(v) for v in per_rank_R.cpu().tolist()]
    # A cloud shaped like a TRAINED scene (synthetic.make_heavy_tail_model: footprints with a heavy tail, dense blobs, depths
    # over seven binades -- R / P = 4.6 against the headline cloud's 1.4, per-tile lists up to 31 k long, every frame needs
    # the depth sort's fourth pass): the same P, camera, image size and protocol, reported as heavy_tail_* and never as
    # `value` (the realistic-footprint regime timed by the driver, not only in profiles/).
    heavy = None
    if extras and world == 1 and args.config == "C4" and args.gaussians is None:
        from mvs_gaussian_splatting_amd.synthetic import make_heavy_tail_model
        ht = make_heavy_tail_model(P, cfg.sh_degree, seed=3, log_footprint_mean=math.log(0.0013))
        keep_model = model
        model = ht.to(dev)                       # fwd_step / train_step read `model` from this scope
        for prm in model.parameters():
            prm.requires_grad_(True)
        for _ in range(3):
            fwd_step(); pkg_h, _ = train_step()
        R_h, V_h = (int(v) for v in frame_counts(pkg_h["render"]))
        for _ in range(20):
            fwd_step()
        t_hf, _ = timed(fwd_step, K)
        for _ in range(10):
            train_step()
        t_ht, _ = timed(train_step, K)
        heavy = {"heavy_tail_fwd_ms": round(t_hf / K * 1e3, 3), "heavy_tail_train_ms": round(t_ht / K * 1e3, 3),
                 "heavy_tail_mpixels_per_s": round(W * H / (t_hf / K) / 1e6, 1),
                 "heavy_tail_scene": {"gaussians": P, "visible": V_h, "instances_R": R_h,
                                      "instances_per_gaussian": round(R_h / P, 2),
                                      "what": "synthetic.make_heavy_tail_model(seed=3, log_footprint_mean=log 0.0013): "
                                              "heavy-tailed footprints, a third of the cloud in dense blobs, depths 0.3..60"}}
        model = keep_model
        del ht, pkg_h
        torch.cuda.empty_cache()
    # The same cloud stored along a Morton curve (mvs_gaussian_splatting_amd/layout.py: an optional step a trainer runs
    # after densify_and_prune).  The headline numbers are measured on SURVEY 8d's cloud as it is generated (uniformly random
    # index order, the worst case for coherence); this region reports what the layout step is worth on it.  LAST region:
    # it reorders the model in place.
    layout = None
    if extras:
        from mvs_gaussian_splatting_amd.layout import reorder_gaussians_
        reorder_gaussians_(model)
        for prm in model.parameters():
            prm.requires_grad_(True)
        fwd_step(); train_step()
        for _ in range(prewarm["fwd"]):              # the headline regions' protocol: the same untimed steps in front
            fwd_step()
        t_lf, _ = timed(fwd_step, K)
        for _ in range(prewarm["train"]):
            train_step()
        t_lt, _ = timed(train_step, K)
        layout = {"morton_layout_fwd_ms": round(t_lf / K * 1e3, 3), "morton_layout_train_ms": round(t_lt / K * 1e3, 3),
                  "morton_layout_mpixels_per_s": round(world * W * H / (t_lf / K) / 1e6, 1)}
    if rank == 0:
        tiles = ((W + 15) // 16) * ((H + 15) // 16)
        tb = max(1, (tiles - 1).bit_length())
        passes = (32 + tb + 7) // 8
        alg_v1 = algorithmic_bytes(P, M, R, W, H, passes)
        moved = bytes_really_moved(P, M, visible, R, W, H)
        # what `achieved` / `frac` are priced on: model v2 where it exists, model v1 elsewhere; v1 is printed beside it
        alg = dict(alg_v1, **moved)
        by_kernel = {}
        for src, n_steps in ((stages_fwd, K), (stages_train, K)):
            for name, (ms, cnt) in src.items():
                if cnt:
                    d = by_kernel.setdefault(name, [0.0, 0])
                    d[0] += ms; d[1] += cnt
        table = {}
        n_fwd = by_kernel.get("preprocess_fwd", [0.0, 1])[1]      # forward calls seen by the stage timers
        n_bwd = by_kernel.get("preprocess_bwd", [0.0, 1])[1]
        for name, (ms, cnt) in by_kernel.items():
            # a stage may be timed in several intervals per frame (two-level binning sorts twice): price the
            # stage per frame against its per-frame algorithmic bytes
            frames = n_bwd if name in ("render_bwd", "preprocess_bwd") else n_fwd
            avg_ms = ms / max(frames, 1)
            gbs = alg[name] / (avg_ms * 1e-3) / 1e9
            table[name] = {"avg_ms": round(avg_ms, 4), "launches": cnt, "frames": frames, "algorithmic_bytes": alg[name],
                           "achieved_GBs": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)}
            if name in moved:
                g1 = alg_v1[name] / (avg_ms * 1e-3) / 1e9
                table[name].update({"byte_model": "v2: bytes this frame has to move (culled Gaussians cost 44 B)",
                                    "algorithmic_bytes_model_v1": alg_v1[name], "achieved_GBs_model_v1": round(g1, 1),
                                    "frac_of_hbm_peak_model_v1": round(g1 / HBM_PEAK_GBS, 4)})
            if name == "radix_sort":
                # SURVEY's model v1 prices upstream's 64-bit (tile, depth) sort of R pairs (152 R).  The default binning
                # does not run that sort: it compacts the V visible Gaussians (16 P read, 8 V written), sorts them by
                # 24-bit depth keys (3 passes x (4 V histogram read + 8 V read + 8 V written)) and partitions the R
                # instances by tile id (2 passes x (4 R + 8 R + 8 R)).  Priced on the bytes it really moves:
                mv = 16 * P + 8 * visible + 3 * 20 * visible + 2 * 20 * R
                table[name]["bytes_moved_two_level"] = mv
                table[name]["achieved_GBs_two_level"] = round(mv / (avg_ms * 1e-3) / 1e9, 1)
                table[name]["frac_of_hbm_peak_two_level"] = round(mv / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        fwd_names = ["preprocess_fwd", "scan_block_sums", "duplicate_with_keys", "radix_sort", "identify_tile_ranges",
                     "render_fwd"]
        # `roofline`: the longest kernel of the metric's first half (the forward); `roofline_step`: the longest kernel of
        # its second half (the train step) -- named explicitly, so that the headline cannot drift to a flattering kernel
        dominant = max((n for n in table), key=lambda n: table[n]["avg_ms"] * (1 if n in fwd_names else 0))
        dominant_step = max((n for n in table), key=lambda n: table[n]["avg_ms"])
        # HBM bytes and instruction counts per launch come from SEPARATE rocprofv3 --pmc passes of this same command
        # (tools/capture_profiles.sh + tools/refresh_profiles.py -> profiles/traffic.json).  The file carries the stamp
        # of the kernel sources it was measured on: on a mismatch the fields are nulled rather than quoted stale.
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        try:
            tj = json.load(open(tfile))
        except Exception:  # noqa: BLE001
            tj = {}
        try:
            static_mix = json.load(open(os.path.join(ROOT, "profiles", "valu_static_mix.json")))
        except Exception:  # noqa: BLE001
            static_mix = {}
        stamp = tj.get("_stamp", {}).get(args.config, {})
        abi_now = _lib.load().gsr_abi_version()
        stamp_ok = bool(stamp) and stamp.get("source_sha16") == source_stamp() and stamp.get("abi") == abi_now
        if not stamp:
            pmc_note = "no PMC measurement on file for this config"
        elif not stamp_ok:
            pmc_note = (f"profiles/traffic.json was measured at kernel sources {stamp.get('source_sha16')} / ABI "
                        f"{stamp.get('abi')}, this run is {source_stamp()} / ABI {abi_now}: PMC-derived fields nulled")
        else:
            pmc_note = f"rocprofv3 --pmc passes at kernel sources {stamp.get('source_sha16')} ({stamp.get('label', '')})"

        def roofline_of(kernel):
            """roofline object of one stage: algorithmic bytes / measured launch time against the HBM peak, plus (when a
            PMC measurement of these very kernel sources is on file) the measured HBM bytes and the issue utilisations."""
            t_s = table[kernel]["avg_ms"] * 1e-3
            entry = tj.get(args.config, {}).get(kernel, {}) if stamp_ok else {}
            issue = None
            sq = entry.get("sq", {})
            if sq.get("SQ_INSTS_VALU"):
                clk = 2.4e9
                # measured issue rates (tools/ubench/valu_rate.hip, profiles/r02/valu_rate.txt): a wave64 VALU instruction
                # occupies its SIMD for >= 2.4 cycles (v_cmp / v_cndmask / v_min / v_max 4.7, v_exp / v_rcp 8.3), a SALU
                # instruction is issued at most every ~4.7 cycles per SIMD; 1024 SIMDs, 256 LDS units
                util = {"valu_issue_floor (2 cycles per wave64 instruction)": round(sq["SQ_INSTS_VALU"] * 2.0 / (1024 * clk * t_s), 3),
                        "salu_issue (4 cycles per instruction per SIMD)": round(sq.get("SQ_INSTS_SALU", 0) * 4.0 / (1024 * clk * t_s), 3)}
                if sq.get("SQ_LDS_IDX_ACTIVE"):
                    util["lds_busy (SQ_LDS_IDX_ACTIVE / CU cycles)"] = round(sq["SQ_LDS_IDX_ACTIVE"] / (256 * clk * t_s), 3)
                if sq.get("SQ_WAVE_CYCLES"):
                    # share of the resident waves' lifetime spent parked on s_waitcnt / s_barrier vs waiting for an issue slot
                    util["wave_time_parked (SQ_WAIT_ANY / SQ_WAVE_CYCLES)"] = round(sq.get("SQ_WAIT_ANY", 0) / sq["SQ_WAVE_CYCLES"], 3)
                    util["wave_time_waiting_for_issue (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES)"] = round(sq.get("SQ_WAIT_INST_ANY", 0) / sq["SQ_WAVE_CYCLES"], 3)
                units = {k: v for k, v in util.items() if k.startswith(("valu", "salu", "lds"))}
                issue = {"insts_per_launch": {k: int(v) for k, v in sq.items()}, "utilisation": util,
                         "limiter": max(units, key=units.get),
                         "note": "utilisations are lower bounds of a unit's busy share at the 2.4 GHz peak clock (the chip "
                                 "clocks at 1.9-2.3 GHz under VALU load; half- and quarter-rate instructions occupy the "
                                 "VALU for 2-3.5x the floor) and cannot exceed 1"}
            # Which roof binds: the HBM fraction (algorithmic bytes / time / 8 TB/s) against the instruction-issue fraction
            # (VALU instructions of one --pmc pass priced at their measured issue rates, over the SIMD cycles of the
            # launch at the clock the GRBM counter of another pass gives for this kernel).  Both are in the object; `bound`
            # names the larger one, `frac` stays the HBM fraction north_star asks for.
            im = None
            mix = entry.get("valu_mix")
            if mix:
                rest_cost = static_mix.get(kernel, {}).get("rest_cost_cycles", 4.5)
                im = issue_model(mix, rest_cost, t_s, float(entry.get("clock_GHz_pmc") or 2.4) * 1e9)
            hbm_frac = table[kernel]["frac_of_hbm_peak"]
            bound = "valu_issue" if (im and im["frac_of_issue_roof"] > hbm_frac) else "hbm"
            out = {"kernel": kernel, "bound": bound, "achieved": table[kernel]["achieved_GBs"], "peak": HBM_PEAK_GBS,
                   "unit": "GB/s", "frac": hbm_frac, "traffic": entry.get("hbm_bytes_per_launch"),
                   "avg_launch_ms": table[kernel]["avg_ms"], "algorithmic_bytes_per_launch": alg[kernel],
                   "frac_of_issue_roof": im["frac_of_issue_roof"] if im else None, "issue_model": im,
                   "issue": issue, "pmc_source": pmc_note}
            if kernel in moved:
                out.update({"byte_model": table[kernel]["byte_model"],
                            "achieved_model_v1": table[kernel]["achieved_GBs_model_v1"],
                            "frac_model_v1": table[kernel]["frac_of_hbm_peak_model_v1"],
                            "algorithmic_bytes_per_launch_model_v1": alg_v1[kernel]})
            # no kernel moves bytes faster than the device can copy them: an `achieved` far above the measured copy rate
            # means the byte model counts bytes the kernel never touches
            out["sane"] = bool(out["achieved"] <= 1.3 * copy_GBs)
            if not out["sane"]:
                print(f"[bench] roofline of {kernel}: achieved {out['achieved']} GB/s exceeds 1.3 x the measured copy rate "
                      f"{copy_GBs} GB/s -- the byte model over-counts", file=sys.stderr, flush=True)
            return out

        copy_GBs = round(copy_ceiling(), 1)
        roof = roofline_of(dominant)
        fwd_ms = t_fwd / K * 1e3
        train_ms = t_train / K * 1e3
        line = {
            "metric": "Mpixels/s fwd + train-step ms @1080p, 6M Gaussians, 1->8 MI355X",
            "value": round(world * W * H / (t_fwd / K) / 1e6, 2), "unit": "Mpixels/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup,
            "extra_untimed_warmup_steps": dict(prewarm, why="GPU clock ramp after idle: see profiles/r03/clock_ramp.txt; "
                                                            "the first K of them are the cold_* region; --no-prewarm makes "
                                                            "the cold region the headline"),
            "ms_per_step": round(train_ms, 3), "fwd_ms_per_step": round(fwd_ms, 3),
            "train_mpixels_per_s": round(world * W * H / (t_train / K) / 1e6, 2),
            "fwd_fps": round(1e3 / fwd_ms, 2),
            "train_step_ms_l1_dssim": round(t_dssim / K * 1e3, 3) if extras else None,
            "per_rank_fwd_ms": [round(per_rank[2 * r], 4) for r in range(world)],
            "per_rank_train_ms": [round(per_rank[2 * r + 1], 4) for r in range(world)],
            "per_rank_instances": per_rank_R,
            "c5_eight_views": c5,
            "profiled_fwd_ms": round(t_fwd_prof / K * 1e3, 3), "profiled_train_ms": round(t_train_prof / K * 1e3, 3),
            "fwd_step_ms_in_order": [round(v, 3) for v in fwd_order], "train_step_ms_in_order": [round(v, 3) for v in train_order],
            "fwd_step_ms_p10_p50_p90": [pct(fwd_steps, 0.1), pct(fwd_steps, 0.5), pct(fwd_steps, 0.9)],
            "train_step_ms_p10_p50_p90": [pct(train_steps, 0.1), pct(train_steps, 0.5), pct(train_steps, 0.9)],
            "hbm_copy_measured_GBs": copy_GBs,
            "hbm_elementwise_kernel_GBs": round(getattr(copy_ceiling, "kernel_rate", 0.0), 1),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: {P} Gaussians, SH degree {cfg.sh_degree}, {W}x{H}, one view per GPU "
                                   f"(rank r renders orbit view r of 8); value = forward-only steps, ms_per_step = "
                                   "render+L1+backward+densify-stats steps",
                       "gaussians": P, "visible": visible, "instances_R": R, "views_per_step": world,
                       "inputs": ("raw parameters (split SH, exp/normalize/sigmoid inside the kernels; INTEGRATION.md "
                                  "option B -- the literal drop-in through the reference getters is unfused_*_ms)"
                                  if pipe.fuse_activations else "reference getters (torch cat/exp/normalize/sigmoid)"),
                       "host_sync": {_rz.SYNC_OFF: "num_rendered read back in every frame",
                                     _rz.SYNC_VERIFIED: "first frame of a shape reads num_rendered back; later frames are enqueued "
                                                        "whole with a capacity, the host waits for the scan kernel's event only "
                                                        "and re-issues a frame that did not fit (never an incomplete image)",
                                     _rz.SYNC_DEFERRED: "deferred count check (opt-in: an overflowed frame raises later)"}[sync_mode],
                       "densify_stats": "fused into preprocess_bwd" if pipe.fuse_densify_stats else "stand-alone kernel",
                       "collective_backend": backend_note if world > 1 else None,
                       "rccl_ranks": rccl_ranks,
                       # invalid: a reduced problem, or ranks on distinct devices whose RCCL communicator did not come
                       # up (the gloo numbers are diagnostics, not the north-star collective)
                       # ... and a rehearsal with several ranks on ONE card (GSR_BENCH_SHARE_GPU=1) is never a multi-GPU result
                       "valid": args.gaussians is None and not rccl_failed and not (share and world > 1)},
            "roofline": roof,
            "roofline_step": roofline_of(dominant_step),
            # the two compositing kernels north_star singles out (VALU-issue-bound: DESIGN.md section 5)
            "roofline_render_fwd": roofline_of("render_fwd") if "render_fwd" in table else None,
            "roofline_render_bwd": roofline_of("render_bwd") if "render_bwd" in table else None,
            "roofline_by_kernel": table,
            "fwd_algorithmic_GB": round(sum(alg[n] for n in fwd_names) / 1e9, 3),
        }
        line.update(cold)
        line.update(readback)
        if unfused:
            line.update(unfused)
        if layout:
            line.update(layout)
        if heavy:
            line.update(heavy)
        if rccl_probe is not None:
            line["rccl_single_rank_probe"] = rccl_probe
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(cfg, args.seed)
            except Exception as ex:  # noqa: BLE001
                line["cpu_baseline"] = {"value": None, "unit": "Mpixels/s", "cores": os.cpu_count(), "kind": "port",
                                        "sample": f"failed: {ex!r}"}
        os.write(result_fd, (json.dumps(line) + "\n").encode())
    if world > 1:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
