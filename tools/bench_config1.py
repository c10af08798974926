"""BASELINE config 1 step (N = 1000, K = 101, 128 x 128, lambda 0.2): HIP path vs the CPU oracle on this box."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mvs_gaussian_splatting_amd.splat2d import generate_2D_gaussian_splatting, combined_loss  # noqa: E402

g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "splat2d_c1.npz"))
dev = torch.device("cuda:0")
names = ["sx", "sy", "rho", "coords", "colours"]
leaves = [torch.tensor(g[k]).to(dev).requires_grad_(True) for k in names]
K, size = int(g["K"]), tuple(int(v) for v in g["size"])
target = torch.tensor(g["target"].astype(np.float32)).to(dev)


def step():
    img = generate_2D_gaussian_splatting(K, *leaves, size)
    loss = combined_loss(img, target, 0.2)
    return torch.autograd.grad(loss, leaves)


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 50
for _ in range(n):
    step()
torch.cuda.synchronize()
hip_ms = (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    for _ in range(5):
        generate_2D_gaussian_splatting(K, *leaves, size)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        generate_2D_gaussian_splatting(K, *leaves, size)
    torch.cuda.synchronize()
fwd_ms = (time.perf_counter() - t0) / n * 1e3
print(f"HIP: forward {fwd_ms:.3f} ms, fwd+loss+bwd step {hip_ms:.3f} ms (wall, includes the det<=0 read-back sync)")

if "--cpu" in sys.argv:
    from oracle.splat2d_ref import splat2d_ref, combined_loss_ref
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    cl = [torch.tensor(g[k]).requires_grad_(True) for k in names]
    t0 = time.perf_counter()
    img = splat2d_ref(K, *cl, size)
    loss = combined_loss_ref(img, target.cpu(), 0.2)
    torch.autograd.grad(loss, cl)
    print(f"CPU oracle ({torch.get_num_threads()} threads): step {(time.perf_counter() - t0) * 1e3:.0f} ms")
