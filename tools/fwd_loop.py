"""Forward-only frames back to back through the C ABI (for kernel traces): python tools/fwd_loop.py [C4] [frames]"""
import sys

import torch

from scene_gpu import GpuScene

cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
s = GpuScene(cfg, fused=True)
s.params.forward_only = 1
for _ in range(3):
    s.forward()
torch.cuda.synchronize()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(n):
    s.forward()
t1.record()
torch.cuda.synchronize()
print(f"{cfg}: {t0.elapsed_time(t1) / n:.4f} ms per forward-only frame")
