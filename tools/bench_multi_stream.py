"""Forward-only frames of a fixed model with 1 .. 4 frames in flight (MultiStreamRenderer: one GraphedRenderer per stream,
own workspaces, shared read-only parameters).  A frame is HBM-bound for its first two thirds (preprocess, binning) and
VALU-bound for the last (compositing): frames on different streams overlap the two.
python tools/bench_multi_stream.py [C2|C3|C4] [frames]"""
import sys
import time

import torch

from mvs_gaussian_splatting_amd.graphed import GraphedRenderer
from mvs_gaussian_splatting_amd.synthetic import CONFIGS, make_scene, PipelineParams

cfgname = sys.argv[1] if len(sys.argv) > 1 else "C4"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 100
cfg = CONFIGS[cfgname]
dev = torch.device("cuda:0")
model, cam0, bg, _ = make_scene(cfg)
model.to(dev)
bg = bg.to(dev)
cams = []
for v in range(8):
    _, c, _, _ = make_scene(cfg, P=1, view=v)
    cams.append(c.to(dev))
pipe = PipelineParams()
W, H = cfg.width, cfg.height

from mvs_gaussian_splatting_amd.graphed import MultiStreamRenderer

with torch.no_grad():
    base = None
    for n in (1, 2, 3, 4):
        mr = MultiStreamRenderer(model, pipe, bg, streams=n)
        views = [cams[i % 8] for i in range(frames)]
        for _ in mr.render_views(views[:16]):
            pass
        mr.check()
        best = None
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in mr.render_views(views):
                pass
            mr.check()
            dt = (time.perf_counter() - t0) / frames * 1e3
            best = dt if best is None else min(best, dt)
        base = base or best
        print(f"{cfgname}: {n} stream(s) {best:.3f} ms/frame ({W * H / best / 1e3:.0f} Mpixels/s): x{base / best:.2f}")
        del mr
