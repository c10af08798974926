"""Forward-only frames per second of a fixed model: eager render() (one Python call + ~27 launches per frame) against
GraphedRenderer (three small copies + one graph launch per frame).  python tools/bench_graphed.py [C2|C3|C4] [frames]"""
import sys
import time

import torch

from mvs_gaussian_splatting_amd import render
from mvs_gaussian_splatting_amd.graphed import GraphedRenderer
from mvs_gaussian_splatting_amd.synthetic import CONFIGS, make_scene, PipelineParams

cfgname = sys.argv[1] if len(sys.argv) > 1 else "C2"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 200
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


def timed(fn):
    for i in range(10):
        fn(cams[i % 8])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(frames):
        fn(cams[i % 8])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / frames * 1e3


with torch.no_grad():
    eager = timed(lambda c: render(c, model, pipe, bg))
    gr = GraphedRenderer(model, pipe, bg)
    graphed = timed(lambda c: gr.render(c))
    gr.check()
    verified = timed(lambda c: gr.render(c, verify=True))
    same = all(torch.equal(gr.render(c, verify=True)["render"], render(c, model, pipe, bg)["render"]) for c in cams)
W, H = cfg.width, cfg.height
print(f"{cfgname} ({cfg.P} Gaussians, {W}x{H}, 8 orbit views, {frames} frames): eager render() {eager:.3f} ms/frame "
      f"({W * H / eager / 1e3:.0f} Mpixels/s); graphed {graphed:.3f} ms/frame ({W * H / graphed / 1e3:.0f} Mpixels/s); "
      f"graphed with per-frame verification {verified:.3f} ms/frame; images identical: {same}")
