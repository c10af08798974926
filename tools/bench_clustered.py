"""Stage times on a CLUSTERED cloud (long per-tile lists, as trained scenes have), next to the uniform bench cloud:
python tools/bench_clustered.py [C3] [P] [fraction in cluster] [log-scale mean]."""
import math
import sys

import numpy as np
import torch

from mvs_gaussian_splatting_amd import _lib
from scene_gpu import GpuScene

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
P = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
lsm = float(sys.argv[4]) if len(sys.argv) > 4 else math.log(0.02)
tail = float(sys.argv[5]) if len(sys.argv) > 5 else 0.6
op_shift = float(sys.argv[6]) if len(sys.argv) > 6 else -1.0


def mutate(model):
    g = torch.Generator().manual_seed(5)
    n = int(frac * model._xyz.shape[0])
    # a dense blob in front of the camera covering ~15 % of the frame, plus a surface-like sheet
    model._xyz[:n] = torch.randn(n, 3, generator=g) * torch.tensor([0.9, 0.6, 0.5]) + torch.tensor([0.5, -0.2, 5.0])
    model._scaling[:] = lsm + tail * torch.randn(model._scaling.shape, generator=g)    # heavy tail: a few huge splats
    model._opacity[:] = 1.5 * torch.randn(model._opacity.shape, generator=g) + op_shift


for name, mut in (("uniform", None), ("clustered", mutate)):
    s = GpuScene(cfg, P=P, mutate=mut)
    dL = torch.sign(torch.rand(3, s.H, s.W, device=s.dev) - 0.5) / (3 * s.H * s.W)
    for _ in range(2):
        s.forward(); s.backward(dL)
    torch.cuda.synchronize()
    prof = _lib.StageProfile()
    s.params.profile = prof._h
    iters = 5
    for _ in range(iters):
        s.forward(); s.backward(dL)
    torch.cuda.synchronize()
    res = prof.collect()
    s.params.profile = None
    tiles = ((s.W + 15) // 16) * ((s.H + 15) // 16)
    ranges = np.zeros((tiles, 2), dtype=np.uint32)
    ft = np.zeros((s.H, s.W), dtype=np.float32); nc = np.zeros((s.H, s.W), dtype=np.uint32)
    _lib.check(s.lib.gsr_debug_read_image(s.img.data_ptr(), s.W, s.H, ft.ctypes.data, nc.ctypes.data, ranges.ctypes.data,
                                          s.stream), "read_image")
    ln = (ranges[:, 1] - ranges[:, 0]).astype(np.int64)
    tm = nc.reshape(s.H // 1, s.W)          # last contributor per pixel
    print(f"{name}: P={s.P} V={s.V} R={s.R}  list len mean {ln.mean():.0f} p99 {np.percentile(ln, 99):.0f} max {ln.max()}"
          f"  mean last-contributor {tm.mean():.0f} max {tm.max()}")
    for k, (ms, n) in res.items():
        if n:
            print(f"   {k:24s} {ms / iters:8.4f} ms/frame")
    del s
    torch.cuda.empty_cache()
