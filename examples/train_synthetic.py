"""A minimal training loop in the shape of the reference's ``train.py:54-160`` on a synthetic scene, built only from
this repo's drop-ins: ``render`` -> fused L1 + D-SSIM loss -> backward -> ``add_densification_stats`` -> Adam step ->
``densify_and_prune`` every ``densification_interval`` iterations -> ``save_ply``.

    python examples/train_synthetic.py [iterations]

It fits a perturbed copy of a small Gaussian cloud to images rendered from the unperturbed cloud (8 orbit views).
"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mvs_gaussian_splatting_amd import render, l1_dssim_loss, add_densification_stats  # noqa: E402
from mvs_gaussian_splatting_amd.densify import densify_and_prune, GROUP_ATTR  # noqa: E402
from mvs_gaussian_splatting_amd.synthetic import SyntheticGaussianModel, PipelineParams, orbit_camera  # noqa: E402


def make_problem(dev, P=4000, W=256, H=160, n_views=8, seed=0):
    """(ground-truth images per view, cameras, trainable model)."""
    gt = SyntheticGaussianModel(P, 3, seed=seed, log_scale_mean=math.log(0.06), extent=(1.6, 1.0, 0.8), centre=(0, 0, 4.0))
    gt._opacity += 1.0
    gt.to(dev)
    cams = [orbit_camera(v, n_views, W, H, 220.0, 220.0, centre=(0.0, 0.0, 4.0), device=dev) for v in range(n_views)]
    bg = torch.zeros(3, device=dev)
    pipe = PipelineParams()
    with torch.no_grad():
        targets = [render(c, gt, pipe, bg)["render"].clone() for c in cams]
    model = SyntheticGaussianModel(P, 3, seed=seed, log_scale_mean=math.log(0.06), extent=(1.6, 1.0, 0.8), centre=(0, 0, 4.0))
    g = torch.Generator().manual_seed(seed + 7)
    # start well away from the optimum: colours forgotten, positions / sizes jittered, everything half transparent
    model._xyz += 0.03 * torch.randn(model._xyz.shape, generator=g)
    model._features_dc = 0.3 * torch.randn(model._features_dc.shape, generator=g)
    model._features_rest = torch.zeros_like(model._features_rest)
    model._scaling += 0.3 * torch.randn(model._scaling.shape, generator=g)
    model._opacity = torch.zeros_like(model._opacity)
    model.to(dev)
    model.percent_dense = 0.01
    for k, a in GROUP_ATTR.items():
        setattr(model, a, torch.nn.Parameter(getattr(model, a).requires_grad_(True)))
    # arguments/__init__.py:85-92 (position_lr_init * spatial_lr_scale, feature_lr, feature_lr / 20, opacity_lr,
    # scaling_lr, rotation_lr); the colours get a larger step because this toy starts from forgotten colours
    lrs = {"xyz": 1.6e-4, "f_dc": 2e-2, "f_rest": 1e-3, "opacity": 0.05, "scaling": 5e-3, "rotation": 1e-3}
    model.optimizer = torch.optim.Adam([{"params": [getattr(model, a)], "lr": lrs[k], "name": k}
                                        for k, a in GROUP_ATTR.items()], lr=0.0, eps=1e-15)
    pipe.fuse_densify_stats = True      # the backward takes the densification statistics; add_densification_stats below
                                        # stays where the reference has it and recognises such a frame
    return targets, cams, bg, pipe, model


def train(dev, iterations=60, densification_interval=20, densify_from_iter=10, extent=2.0, grad_threshold=0.0006, log=None,
          spatial_order=False):
    """spatial_order: after every densification the cloud (and the Adam moments) is stored along a Morton curve
    (mvs_gaussian_splatting_amd/layout.py) instead of the reference's [kept | clones | children] order."""
    targets, cams, bg, pipe, model = make_problem(dev)
    history, sizes = [], []
    for it in range(1, iterations + 1):
        v = (it * 3) % len(cams)
        pkg = render(cams[v], model, pipe, bg)
        loss = l1_dssim_loss(pkg["render"], targets[v], 0.2)
        loss.backward()
        with torch.no_grad():
            add_densification_stats(model, pkg["viewspace_points"], pkg["radii"])
            model.optimizer.step()
            model.optimizer.zero_grad(set_to_none=True)
            if it > densify_from_iter and it % densification_interval == 0:
                info = densify_and_prune(model, grad_threshold, 0.005, extent, 20, spatial_order=spatial_order)
                sizes.append(info["points"])
                if log:
                    log(f"  iteration {it}: densify_and_prune -> {info}")
        history.append(float(loss.detach()))
        if log and it % 10 == 0:
            log(f"iteration {it}: loss {history[-1]:.5f}  points {model._xyz.shape[0]}")
    return model, history, sizes


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    model, history, _ = train(torch.device("cuda:0"), iterations=n, log=print)
    from mvs_gaussian_splatting_amd.ply_io import save_ply
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "point_cloud.ply")
    save_ply(model, out)
    print(f"loss {history[0]:.5f} -> {history[-1]:.5f}; wrote {out}")
