"""Per-stage device times of the raw C-ABI forward/backward on a bench config (no Python host work in
between): python tools/kernel_bench.py [C4] [iters] [--fused]."""
import sys

import torch

from mvs_gaussian_splatting_amd import _lib
from scene_gpu import GpuScene

args = [a for a in sys.argv[1:] if not a.startswith("--")]
cfg = args[0] if len(args) > 0 else "C4"
iters = int(args[1]) if len(args) > 1 else 10
def wide_depth(model):
    """--wide-depth: the same screen positions and footprints with the depths spread over 23 binades (0.25 .. 2e6): every
    frame then needs the depth sort's fourth pass (the bench cloud's z in 3..9 never takes it)."""
    import math
    g = torch.Generator().manual_seed(11)
    P = model._xyz.shape[0]
    z = torch.exp(torch.rand(P, generator=g) * (math.log(2.0e6) - math.log(0.25)) + math.log(0.25))
    s_ = z / model._xyz[:, 2]
    model._xyz *= s_[:, None]
    model._scaling += torch.log(s_)[:, None]


def morton(model):
    """--morton: the same cloud with its Gaussians stored along a Morton curve (mvs_gaussian_splatting_amd/layout.py)"""
    from mvs_gaussian_splatting_amd.layout import reorder_gaussians_
    reorder_gaussians_(model)


s = GpuScene(cfg, fused="--fused" in sys.argv,       # --fused: raw parameters + split SH, as render() feeds them
             mutate=wide_depth if "--wide-depth" in sys.argv else morton if "--morton" in sys.argv else None)
s.fuse_stats = "--stats" in sys.argv                 # --stats: the backward also takes the densification statistics
dL = torch.sign(torch.rand(3, s.H, s.W, device=s.dev) - 0.5) / (3 * s.H * s.W)
for _ in range(2):
    s.forward(); s.backward(dL)
torch.cuda.synchronize()
prof = _lib.StageProfile()
s.params.profile = prof._h
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True); t2 = torch.cuda.Event(enable_timing=True)
tf = tb = 0.0
for _ in range(iters):
    t0.record(); s.forward(); t1.record(); s.backward(dL); t2.record()
    torch.cuda.synchronize()
    tf += t0.elapsed_time(t1); tb += t1.elapsed_time(t2)
res = prof.collect()
print(f"{cfg}: P={s.P} R={s.R} visible={int((s.radii > 0).sum())} fwd={tf / iters:.3f} ms bwd={tb / iters:.3f} ms")
for k, (ms, n) in res.items():
    if n:
        print(f"  {k:24s} {ms / n:8.4f} ms")
# the forward-only variant (what render() runs under no_grad: no backward state is kept)
s.params.forward_only = 1
s.forward()
torch.cuda.synchronize()
tf = 0.0
for _ in range(iters):
    t0.record(); s.forward(); t1.record()
    torch.cuda.synchronize()
    tf += t0.elapsed_time(t1)
res = prof.collect()
s.params.profile = None
s.params.forward_only = 0
print(f"forward-only: fwd={tf / iters:.3f} ms  " + "  ".join(f"{k}={ms / n:.4f}" for k, (ms, n) in res.items() if n))
# the forward without the count read-back (gsr_forward, capacity = 1.5 x the instance count): what every frame after the
# first runs through the operator
s.forward()
s.forward_sync_free()
torch.cuda.synchronize()
tf = 0.0
wall0 = __import__("time").perf_counter()
t0.record()
for _ in range(iters):
    s.forward_sync_free()
t1.record()
torch.cuda.synchronize()
wall = (__import__("time").perf_counter() - wall0) * 1e3 / iters
print(f"sync-free forward: {t0.elapsed_time(t1) / iters:.3f} ms per frame on the device, {wall:.3f} ms wall (back to back, capacity {s.cap})")
# ... and without the depth sort's fourth pass (GsrParams.depth_span_lt24: what the operator issues while the frames it has
# seen span fewer than 0.9 x 2^24 depth-key steps)
if "--wide-depth" not in sys.argv:
    s.params.depth_span_lt24 = 1
    s.forward_sync_free()
    torch.cuda.synchronize()
    t0.record()
    for _ in range(iters):
        s.forward_sync_free()
    t1.record()
    torch.cuda.synchronize()
    s.params.depth_span_lt24 = 0
    print(f"sync-free forward, depth_span_lt24: {t0.elapsed_time(t1) / iters:.3f} ms per frame on the device")
    # ... with the counts event recorded behind the scan kernel (what the verified mode adds on the device side)
    import ctypes
    ev = ctypes.c_void_p()
    _lib.check(s.lib.gsr_event_create(ctypes.byref(ev)), "event_create")
    s.params.depth_span_lt24 = 1
    s.forward_sync_free(event=ev)
    torch.cuda.synchronize()
    t0.record()
    for _ in range(iters):
        s.forward_sync_free(event=ev)
    t1.record()
    torch.cuda.synchronize()
    s.params.depth_span_lt24 = 0
    print(f"sync-free forward, depth_span_lt24, counts event recorded: {t0.elapsed_time(t1) / iters:.3f} ms per frame on the device")
    _lib.check(s.lib.gsr_event_destroy(ev), "event_destroy")
