"""Host time to ISSUE one forward frame (no GPU wait): raw C ABI gsr_forward vs the Python operator behind render().
python tools/host_overhead.py [C4] -- if the host needs longer per frame than the GPU, the GPU idles."""
import sys
import time

import torch

from mvs_gaussian_splatting_amd import render
from mvs_gaussian_splatting_amd.synthetic import CONFIGS, make_scene, PipelineParams
from scene_gpu import GpuScene

cfgname = sys.argv[1] if len(sys.argv) > 1 else "C4"
n = 12
s = GpuScene(cfgname, fused=True)
s.params.forward_only = 1
s.forward(); s.forward_sync_free()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    s.forward_sync_free()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{cfgname} raw gsr_forward: host issue {1e3 * (t1 - t0) / n:.3f} ms/frame, drained after {1e3 * (t2 - t0) / n:.3f} ms/frame")
del s
torch.cuda.empty_cache()
cfg = CONFIGS[cfgname]
dev = torch.device("cuda:0")
model, cam, bg, _ = make_scene(cfg)
model.to(dev); cam.to(dev); bg = bg.to(dev)
pipe = PipelineParams()
with torch.no_grad():
    for _ in range(3):
        render(cam, model, pipe, bg)
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            render(cam, model, pipe, bg)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{cfgname} render() under no_grad: host issue {1e3 * (t1 - t0) / n:.3f} ms/frame, drained after {1e3 * (t2 - t0) / n:.3f} ms/frame")
import cProfile
import pstats
with torch.no_grad():
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        render(cam, model, pipe, bg)
    pr.disable()
    torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
