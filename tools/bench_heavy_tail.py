"""BASELINE config C4 with a cloud shaped like a trained scene (synthetic.make_heavy_tail_model: heavy-tailed footprints,
dense blobs, depths over seven binades) instead of SURVEY §8(d)'s uniform box -- the evidence that nothing collapses on
upstream-like footprints: per-stage times, instance counts, list-length percentiles, how many Gaussians
take the rare paths (more than 64 gradient rows, rects beyond the packed sort payload), whether the depth sort's fourth
pass ran, and the forward without the count read-back.

    PYTHONPATH=.:tools python tools/bench_heavy_tail.py [P] [iters] [log footprint mean] [--morton]

--morton: the same cloud stored along a Morton curve (mvs_gaussian_splatting_amd/layout.py).
"""
import ctypes as C
import math
import sys

import numpy as np
import torch

from mvs_gaussian_splatting_amd import _lib
from mvs_gaussian_splatting_amd.synthetic import make_heavy_tail_model
from scene_gpu import GpuScene

argv = [a for a in sys.argv[1:] if not a.startswith("--")]
P = int(argv[0]) if len(argv) > 0 else 6_000_000
iters = int(argv[1]) if len(argv) > 1 else 10
lfm = float(argv[2]) if len(argv) > 2 else math.log(0.0013)


def mutate(model):
    ht = make_heavy_tail_model(model._xyz.shape[0], model.max_sh_degree, seed=3, log_footprint_mean=lfm)
    model._xyz, model._scaling, model._opacity = ht._xyz, ht._scaling, ht._opacity
    if "--morton" in sys.argv:
        from mvs_gaussian_splatting_amd.layout import reorder_gaussians_
        reorder_gaussians_(model)


s = GpuScene("C4", P=P, mutate=mutate, fused=True)
dL = torch.sign(torch.rand(3, s.H, s.W, device=s.dev) - 0.5) / (3 * s.H * s.W)
for _ in range(2):
    s.forward(); s.backward(dL)
torch.cuda.synchronize()
prof = _lib.StageProfile()
s.params.profile = prof._h
t0, t1, t2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
tf = tb = 0.0
for _ in range(iters):
    t0.record(); s.forward(); t1.record(); s.backward(dL); t2.record()
    torch.cuda.synchronize()
    tf += t0.elapsed_time(t1); tb += t1.elapsed_time(t2)
res = prof.collect()
s.params.profile = None
R, V = s.R, s.V
counts = (C.c_uint32 * 8)()
_lib.check(s.lib.gsr_debug_read_counts(s.geom.data_ptr(), s.P, counts, s.stream), "read_counts")
tiles = ((s.W + 15) // 16) * ((s.H + 15) // 16)
ranges = torch.zeros(tiles, 2, dtype=torch.int32, device=s.dev)
nc = torch.zeros(s.H, s.W, dtype=torch.int32, device=s.dev)
_lib.check(s.lib.gsr_debug_read_image(s.img.data_ptr(), s.W, s.H, None, nc.data_ptr(), ranges.data_ptr(), s.stream), "read_image")
torch.cuda.synchronize()
ln = (ranges[:, 1] - ranges[:, 0]).cpu().numpy().astype(np.int64)
last = nc.cpu().numpy().astype(np.int64)
radii = s.radii.cpu().numpy()
vis = radii > 0
tiles_of = np.ceil(2 * radii[vis] / 16.0 + 1) ** 2
print(f"heavy-tailed C4: P={s.P} visible V={V} instances R={R} (R/P {R / s.P:.2f}, R/V {R / max(V, 1):.2f})")
print(f"  radii (px): median {np.median(radii[vis]):.0f}  p90 {np.percentile(radii[vis], 90):.0f}  p99 {np.percentile(radii[vis], 99):.0f}"
      f"  p99.9 {np.percentile(radii[vis], 99.9):.0f}  max {radii.max()}")
print(f"  per-tile list length: mean {ln.mean():.0f}  p50 {np.percentile(ln, 50):.0f}  p90 {np.percentile(ln, 90):.0f}  p99 {np.percentile(ln, 99):.0f}  max {ln.max()}")
print(f"  last contributor per pixel: mean {last.mean():.0f}  p99 {np.percentile(last, 99):.0f}  max {last.max()}")
print(f"  Gaussians with more than 64 gradient rows (big_list): {counts[2]};  square rects above 16 tiles (payload fall-back, upper bound): {int((tiles_of > 16).sum())}")
print(f"  depth keys: span {counts[5] - (~counts[4] & 0xffffffff)} steps -> fourth depth-sort pass "
      f"{'RAN' if counts[6] else 'not needed'} ({counts[6]} elements)")
print(f"  forward {tf / iters:.3f} ms  backward {tb / iters:.3f} ms  (two-call forward, raw C ABI, fused inputs)")
for k, (ms, n) in res.items():
    if n:
        print(f"   {k:24s} {ms / iters:8.4f} ms/frame  ({n // iters} interval(s) per frame)")
s.forward()
s.forward_sync_free()
torch.cuda.synchronize()
t0.record()
for _ in range(iters):
    s.forward_sync_free()
t1.record()
torch.cuda.synchronize()
print(f"  forward without the count read-back (capacity {s.cap}): {t0.elapsed_time(t1) / iters:.3f} ms per frame")
