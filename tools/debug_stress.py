import math, sys, torch, numpy as np
sys.path.insert(0, "tests")
from conftest import make_settings
from test_gpu_parity import _stress_model
from mvs_gaussian_splatting_amd import GaussianRasterizer
from mvs_gaussian_splatting_amd.synthetic import orbit_camera
from gpu_util import product_settings
from oracle import rasterize_ref
dev = torch.device("cuda:0")
model = _stress_model()
cam = orbit_camera(1, 8, 208, 136, 120.0, 120.0)
bg = torch.tensor([0.2, 0.4, 0.1]); target = torch.rand(3, 136, 208, generator=torch.Generator().manual_seed(5))
deg, smod = 2, 1.3
st_o = make_settings(cam, bg, deg, scale_modifier=smod)
d = torch.float64
names = ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity")
def run_oracle(dtype, upstream=True):
    leaves = {k: getattr(model, k).detach().to(dtype).requires_grad_(True) for k in names}
    c, r, a = rasterize_ref(leaves["_xyz"], None, torch.sigmoid(leaves["_opacity"]), st_o,
                            shs=torch.cat((leaves["_features_dc"], leaves["_features_rest"]), 1), scales=torch.exp(leaves["_scaling"]),
                            rotations=torch.nn.functional.normalize(leaves["_rotation"]), want_aux=True, want_margin=True, upstream_grad=upstream)
    (c - target.to(dtype)).abs().mean().backward()
    return {k: v.grad for k, v in leaves.items()}, r, a
g64, radii, a64 = run_oracle(d)
g32, _, _ = run_oracle(torch.float32)
st = product_settings(cam, bg, deg, dev, scale_modifier=smod)
gl = {k: getattr(model, k).detach().to(dev).requires_grad_(True) for k in names}
img, rad = GaussianRasterizer(st)(means3D=gl["_xyz"], means2D=torch.zeros(gl["_xyz"].shape[0], 3, device=dev), opacities=torch.sigmoid(gl["_opacity"]),
                                shs=torch.cat((gl["_features_dc"], gl["_features_rest"]), 1), scales=torch.exp(gl["_scaling"]),
                                rotations=torch.nn.functional.normalize(gl["_rotation"]))
(img - target.to(dev)).abs().mean().backward()
n = 1200 // 8
for k in names:
    ref = g64[k]; got = gl[k].grad.cpu().to(d); o32 = g32[k].to(d)
    scale = float(ref.abs().max())
    e_hip = (got - ref).abs().reshape(ref.shape[0], -1).max(dim=1).values / scale
    e_o32 = (o32 - ref).abs().reshape(ref.shape[0], -1).max(dim=1).values / scale
    i = int(e_hip.argmax())
    print(k, "scale", scale, "worst idx", i, "group", i // n, "hip err", float(e_hip[i]), "oracle-fp32 err at idx", float(e_o32[i]), "max oracle-fp32 err", float(e_o32.max()), "radius", int(radii[i]))
    print("   ref", ref[i].flatten()[:4].tolist(), "hip", got[i].flatten()[:4].tolist(), "o32", o32[i].flatten()[:4].tolist())
    # per-group worst
    print("   per-group worst hip err:", [round(float(e_hip[g*n:(g+1)*n].max()), 6) for g in range(8)])
    print("   per-group worst o32 err:", [round(float(e_o32[g*n:(g+1)*n].max()), 6) for g in range(8)])
