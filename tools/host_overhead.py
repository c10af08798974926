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
# ---- where the host time of a render() frame goes: the two native calls of the verified mode, timed from Python -------------
from mvs_gaussian_splatting_amd import _lib  # noqa: E402
lib = _lib.load()
acc = {"gsr_forward": 0.0, "gsr_event_wait": 0.0}
orig = {k: getattr(lib, k) for k in acc}


def _timed(name):
    fn = orig[name]

    def wrapped(*a):
        t = time.perf_counter()
        r = fn(*a)
        acc[name] += time.perf_counter() - t
        return r
    return wrapped


for k in acc:
    setattr(lib, k, _timed(k))
with torch.no_grad():
    for rep in range(2):
        for k in acc:
            acc[k] = 0.0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4 * n):
            render(cam, model, pipe, bg)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        per = 1e3 / (4 * n)
        print(f"{cfgname} render() frame: {per * (t2 - t0):.4f} ms wall; host: gsr_forward {per * acc['gsr_forward']:.4f} ms (enqueue of the "
              f"whole frame), gsr_event_wait {per * acc['gsr_event_wait']:.4f} ms (scan kernel's event), Python around them "
              f"{per * ((t1 - t0) - acc['gsr_forward'] - acc['gsr_event_wait']):.4f} ms")
for k in acc:
    setattr(lib, k, orig[k])
