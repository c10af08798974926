"""densify_and_prune at C4 size (6 M Gaussians, SH 3, Adam moments): HIP plan + read-once/write-once passes."""
import math
import sys
import time

import torch

from mvs_gaussian_splatting_amd.densify import densify_and_prune, GROUP_ATTR

P = int(sys.argv[1]) if len(sys.argv) > 1 else 6_000_000
dev = torch.device("cuda:0")


class M:
    percent_dense = 0.01


def make():
    g = torch.Generator(device=dev).manual_seed(0)
    m = M()
    shapes = {"xyz": (P, 3), "f_dc": (P, 1, 3), "f_rest": (P, 15, 3), "opacity": (P, 1), "scaling": (P, 3), "rotation": (P, 4)}
    for k, a in GROUP_ATTR.items():
        t = torch.randn(shapes[k], device=dev, generator=g)
        if k == "scaling":
            t = math.log(0.05) + 1.2 * t
        if k == "opacity":
            t = 2.5 * t - 1.0
        setattr(m, a, torch.nn.Parameter(t))
    m.optimizer = torch.optim.Adam([{"params": [getattr(m, a)], "lr": 0.0, "name": k} for k, a in GROUP_ATTR.items()], eps=1e-15)
    for k, a in GROUP_ATTR.items():
        getattr(m, a).grad = torch.zeros_like(getattr(m, a))
    m.optimizer.step()
    m.denom = torch.randint(0, 4, (P, 1), device=dev, generator=g).float()
    m.xyz_gradient_accum = torch.rand(P, 1, device=dev, generator=g) * 0.0006 * m.denom
    m.max_radii2D = torch.zeros(P, device=dev)
    return m


for it in range(3):
    m = make()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    info = densify_and_prune(m, 0.0002, 0.005, 5.0, 20)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    moved = (P + info["points"]) * 59 * 3 * 4
    print(f"P={P} -> {info}  {dt * 1e3:.2f} ms wall, {moved / dt / 1e9:.0f} GB/s of parameter+moment bytes (read + written)")
    del m
