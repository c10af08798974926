"""End to end: the reference's training step (train.py:91-134) assembled from this repo's drop-ins -- render, fused
L1 + D-SSIM loss, backward, densification statistics, Adam, densify_and_prune with optimizer-state surgery, PLY round
trip -- runs, converges and keeps every piece of state consistent."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "examples"))


@pytest.mark.parametrize("spatial_order", [False, True])
def test_training_loop_converges_and_state_stays_consistent(tmp_path, spatial_order):
    """spatial_order: every densification also stores the cloud and the Adam moments along a Morton curve (layout.py)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from train_synthetic import train
    from mvs_gaussian_splatting_amd.densify import GROUP_ATTR
    from mvs_gaussian_splatting_amd.ply_io import save_ply, load_ply
    dev = torch.device("cuda:0")
    model, history, sizes = train(dev, iterations=100, densification_interval=20, densify_from_iter=10,
                                  spatial_order=spatial_order)
    if spatial_order:
        from mvs_gaussian_splatting_amd.layout import morton_permutation
        perm = morton_permutation(model._xyz)      # positions moved since the last densification: mostly, not exactly, sorted
        assert float((perm[1:] > perm[:-1]).float().mean()) > 0.9
    first, last = sum(history[:8]) / 8, sum(history[-8:]) / 8
    assert all(h == h for h in history) and last < 0.7 * first, (first, last)
    assert len(sizes) == 5 and all(s > 0 for s in sizes) and sizes[-1] > 4000        # the cloud grew
    n = model._xyz.shape[0]
    assert n == sizes[-1]
    for group in model.optimizer.param_groups:            # the optimizer owns exactly the model's tensors, with state
        p = group["params"][0]
        assert p is getattr(model, GROUP_ATTR[group["name"]]) and p.shape[0] == n and p.requires_grad
        st = model.optimizer.state[p]
        assert st["exp_avg"].shape == p.shape and st["exp_avg_sq"].shape == p.shape
    assert model.xyz_gradient_accum.shape == (n, 1) and model.denom.shape == (n, 1) and model.max_radii2D.shape == (n,)
    path = str(tmp_path / "point_cloud.ply")
    save_ply(model, path)
    back = load_ply(path)
    for k in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
        assert torch.equal(back[k].cpu(), getattr(model, k).detach().cpu()), k


@pytest.mark.parametrize("spatial_order", [False, True])
def test_offline_render_of_a_saved_ply_matches_the_live_model(tmp_path, spatial_order):
    """render.py's path: save_ply -> load_ply -> render under no_grad gives the images of the live model bit for bit;
    with the loaded model stored along a Morton curve (the example's default) up to the order of equal-depth ties."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from train_synthetic import make_problem
    from render_ply import render_set
    from mvs_gaussian_splatting_amd import render
    from mvs_gaussian_splatting_amd.ply_io import save_ply
    from mvs_gaussian_splatting_amd.synthetic import orbit_camera
    dev = torch.device("cuda:0")
    _, _, bg, pipe, model = make_problem(dev, P=3000)
    path = str(tmp_path / "point_cloud.ply")
    save_ply(model, path)
    _, images = render_set(path, str(tmp_path / "renders"), n_views=3, width=200, height=120, spatial_order=spatial_order)
    for v, img in enumerate(images):
        cam = orbit_camera(v, 3, 200, 120, 220.0, 220.0, centre=(0.0, 0.0, 4.0), device=dev)
        with torch.no_grad():
            live = render(cam, model, pipe, bg)["render"]
        if spatial_order:      # a pair of equal float32 depths in one tile may blend in the other order (layout.py)
            assert float((img != live).any(dim=0).float().mean()) < 1e-3 and float((img - live).abs().max()) < 5e-3
        else:
            assert torch.equal(img, live)
        data = open(tmp_path / "renders" / f"{v:05d}.ppm", "rb").read()
        assert data.startswith(b"P6\n200 120\n255\n") and len(data) == 15 + 200 * 120 * 3
