"""distCUDA2 (3-NN mean squared distance) timing: python tools/bench_knn.py [N ...]"""
import sys
import time
import torch
from mvs_gaussian_splatting_amd.knn import distCUDA2

dev = torch.device("cuda:0")
for n in [int(a) for a in sys.argv[1:]] or [100_000, 1_000_000, 6_000_000]:
    g = torch.Generator(device=dev).manual_seed(0)
    for name, pts in (("uniform", torch.rand(n, 3, device=dev, generator=g) * torch.tensor([12.0, 6.8, 6.0], device=dev)),
                      ("clustered", torch.randn(n, 3, device=dev, generator=g) * torch.tensor([0.9, 0.6, 0.5], device=dev)
                       * (1 + 9 * (torch.rand(n, 1, device=dev, generator=g) < 0.1)))):
        distCUDA2(pts)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        d = distCUDA2(pts)
        torch.cuda.synchronize()
        print(f"N={n:>8} {name:9s}: {1e3 * (time.perf_counter() - t0):8.2f} ms   mean dist2 {float(d.mean()):.3e}")
