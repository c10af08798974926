"""Three train steps at C3 with the roctx ranges on (gsr_enable_markers): under
    rocprofv3 --marker-trace --kernel-trace --stats -d <dir> -- python3 tools/marker_demo.py
the marker trace lists one "gsr:<stage>" range per stage and frame (tracing hooks)."""
import torch

from mvs_gaussian_splatting_amd import _lib, render, l1_loss, add_densification_stats
from mvs_gaussian_splatting_amd.synthetic import CONFIGS, make_scene, PipelineParams

_lib.enable_markers(True)
dev = torch.device("cuda:0")
model, cam, bg, target = make_scene(CONFIGS["C3"])
model.to(dev); cam.to(dev)
bg, target = bg.to(dev), target.to(dev)
for p in model.parameters():
    p.requires_grad_(True)
for _ in range(3):
    for p in model.parameters():
        p.grad = None
    pkg = render(cam, model, PipelineParams(), bg)
    l1_loss(pkg["render"], target).backward()
    add_densification_stats(model, pkg["viewspace_points"], pkg["radii"])
torch.cuda.synchronize()
print("done")
