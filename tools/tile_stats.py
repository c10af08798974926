"""Distribution of per-tile instance counts on a bench config."""
import sys
import numpy as np
import torch
from mvs_gaussian_splatting_amd import _lib
from scene_gpu import GpuScene
s = GpuScene(sys.argv[1] if len(sys.argv) > 1 else "C4")
s.forward()
gx, gy = (s.W + 15) // 16, (s.H + 15) // 16
ranges = torch.empty(gx * gy, 2, dtype=torch.int32, device=s.dev)
_lib.check(s.lib.gsr_debug_read_image(s.img.data_ptr(), s.W, s.H, None, None, ranges.data_ptr(), s.stream), "img")
torch.cuda.synchronize()
r = ranges.cpu().numpy().astype(np.int64)
n = r[:, 1] - r[:, 0]
print("tiles", n.size, "sum", n.sum(), "mean", n.mean(), "max", n.max(), "pcts(50,90,99)", np.percentile(n, [50, 90, 99]))
img = n.reshape(gy, gx)
print("row means", img.mean(axis=1).astype(int)[::8])
print("col means", img.mean(axis=0).astype(int)[::12])
