"""Generates tests/golden/preprocess.npz from the reference's own importable Python (run ONLY in the build container,
where /root/reference exists; the fixture -- plain input/output arrays -- is committed):

    python tests/golden/make_golden_preprocess.py

Pins the two preprocess sub-steps of the rasterizer the reference holds a Python twin of:

* SH -> RGB exactly as gaussian_renderer/__init__.py:76-80 spells it: features [P,M,3] -> transpose/view [P,3,M],
  dir = (xyz - camera_center) / |.|, eval_sh(active_sh_degree, ...) (the reference's function, imported), then
  clamp_min(. + 0.5, 0); the `< 0` clamp mask (the rasterizer's `clamped` flags) is recorded beside it;
* cov3D exactly as scene/gaussian_model.py:28-32 spells it, through the reference's own build_scaling_rotation and
  strip_symmetric (utils/general_utils.py:64-110, imported).  Those two hard-code device="cuda"; this container has
  no GPU, so they are called under a patch of torch.zeros that drops the `device` keyword (nothing else changes).
"""
import os
import sys
from unittest import mock

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    sys.path.insert(0, REF)
    from utils.sh_utils import eval_sh
    from utils.general_utils import build_scaling_rotation, strip_symmetric

    g = torch.Generator().manual_seed(4321)
    P, max_deg = 600, 3
    M = (max_deg + 1) ** 2
    # a cloud in front of a camera that sits off the origin (camera_center != 0 matters for the view directions)
    xyz = (torch.rand(P, 3, generator=g) * 2 - 1) * torch.tensor([2.5, 1.5, 2.0]) + torch.tensor([0.0, 0.0, 6.0])
    camera_center = torch.tensor([0.3, -0.2, -0.5])
    f_dc = torch.randn(P, 1, 3, generator=g)
    f_rest = 0.6 * torch.randn(P, M - 1, 3, generator=g)          # strong enough that many channels clamp at 0
    features = torch.cat((f_dc, f_rest), dim=1)                   # get_features layout [P, M, 3]
    out = {"xyz": xyz.numpy(), "camera_center": camera_center.numpy(), "features": features.numpy()}
    for deg in range(max_deg + 1):
        shs_view = features.transpose(1, 2).view(-1, 3, M)
        dir_pp = xyz - camera_center.repeat(features.shape[0], 1)
        dir_pp_normalized = dir_pp / dir_pp.norm(dim=1, keepdim=True)
        sh2rgb = eval_sh(deg, shs_view, dir_pp_normalized)
        out[f"colors_deg{deg}"] = torch.clamp_min(sh2rgb + 0.5, 0.0).numpy()
        out[f"clamped_deg{deg}"] = (sh2rgb + 0.5 < 0).numpy()
    assert out["clamped_deg3"].mean() > 0.05

    scaling = torch.exp(torch.randn(P, 3, generator=g) * 0.7 - 3.0)          # activated scales (get_scaling)
    rotation_raw = torch.randn(P, 4, generator=g)                            # un-normalised, as stored in _rotation
    out.update(scaling=scaling.numpy(), rotation_raw=rotation_raw.numpy(), opacity=torch.rand(P, 1, generator=g).numpy())
    real_zeros = torch.zeros

    def zeros_on_cpu(*a, **kw):
        kw.pop("device", None)
        return real_zeros(*a, **kw)

    for mod in (1.0, 1.7):
        with mock.patch.object(torch, "zeros", zeros_on_cpu):
            L = build_scaling_rotation(mod * scaling, rotation_raw)
            actual_covariance = L @ L.transpose(1, 2)
            symm = strip_symmetric(actual_covariance)
        out[f"cov3D_mod{mod}"] = symm.numpy()
    np.savez_compressed(os.path.join(OUT, "preprocess.npz"), **out)
    print("wrote preprocess.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
