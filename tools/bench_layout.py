"""Effect of the spatial (Morton) order of the Gaussians on the frame: python tools/bench_layout.py [C4] [steps]"""
import sys, time, torch
from mvs_gaussian_splatting_amd import render, l1_loss, add_densification_stats
from mvs_gaussian_splatting_amd.layout import reorder_gaussians_
from mvs_gaussian_splatting_amd.synthetic import CONFIGS, make_scene, PipelineParams

cfg = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C4"]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
model, cam, bg, target = make_scene(cfg)
model.to(dev); cam.to(dev); bg, target = bg.to(dev), target.to(dev)
pipe = PipelineParams()

def run(label):
    for p in model.parameters():
        p.requires_grad_(True)
    def fwd():
        with torch.no_grad():
            return render(cam, model, pipe, bg)["render"]
    def step():
        for p in model.parameters():
            p.grad = None
        pkg = render(cam, model, pipe, bg)
        l1_loss(pkg["render"], target).backward()
        add_densification_stats(model, pkg["viewspace_points"], pkg["radii"])
    for _ in range(40):
        fwd()
    for _ in range(10):
        step()
    out = []
    for fn in (fwd, step):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(K):
            fn()
        torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / K * 1e3)
    img = fwd()
    print(f"{label}: forward {out[0]:.3f} ms, train step {out[1]:.3f} ms", flush=True)
    return img

a = run("index order (SURVEY 8d: seeded uniform random)")
reorder_gaussians_(model)
b = run("Morton order")
print("max |image difference|:", float((a - b).abs().max()))
