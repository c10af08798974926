import sys, math
import numpy as np
import torch
sys.path.insert(0, "tests")
from conftest import make_settings, small_scene
from gpu_util import forward_with_state, product_settings
from oracle import rasterize_ref
dev = torch.device("cuda:0")
model, cam, bg, _ = small_scene(P=2500, sh_degree=0, width=160, height=96)
g = torch.Generator().manual_seed(11)
z = torch.exp(torch.rand(2500, generator=g) * (math.log(2.0e6) - math.log(0.25)) + math.log(0.25))
scale = z / model._xyz[:, 2]
model._xyz *= scale[:, None]
model._scaling += torch.log(scale)[:, None]
col, radii, aux = rasterize_ref(model.get_xyz, None, model.get_opacity, make_settings(cam, bg, 0), shs=model.get_features,
                                scales=model.get_scaling, rotations=model.get_rotation, want_aux=True, want_margin=True)
out = forward_with_state(dev, product_settings(cam, bg, 0, dev), model.get_xyz, model.get_opacity, shs=model.get_features,
                         scales=model.get_scaling, rotations=model.get_rotation)
bad = (out["radii"] != radii).nonzero().flatten()
print("radii mismatches", bad.numel(), "V", int((radii > 0).sum()))
for i in bad[:8].tolist():
    print(i, "hip", int(out["radii"][i]), "oracle", int(radii[i]), "z", float(model.get_xyz[i, 2]), "scale", model.get_scaling[i].tolist())
print("keys equal", np.array_equal(out["keys"], aux["keys"]) if out["keys"].shape == aux["keys"].shape else (out["keys"].shape, aux["keys"].shape))
