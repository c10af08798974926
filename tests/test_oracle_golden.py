"""Oracle and host helpers against the fixtures generated from the reference's importable Python
(tests/golden/make_golden.py, make_golden_splat2d.py).  CPU only."""
import math
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_oracle_eval_sh_matches_reference():
    from oracle import eval_sh_ref
    g = np.load(os.path.join(GOLD, "sh.npz"))
    sh_cm = torch.tensor(g["sh"])                  # reference layout [P, 3, 16]
    sh = sh_cm.transpose(1, 2).contiguous()        # rasterizer layout [P, 16, 3]
    dirs = torch.tensor(g["dirs"])
    for deg in range(4):
        out = eval_sh_ref(deg, sh, dirs)
        assert torch.allclose(out, torch.tensor(g[f"eval_deg{deg}"]), rtol=1e-6, atol=1e-6), deg


def test_host_eval_sh_and_rgb2sh_match_reference():
    from mvs_gaussian_splatting_amd.sh import eval_sh, RGB2SH, SH2RGB
    g = np.load(os.path.join(GOLD, "sh.npz"))
    sh, dirs = torch.tensor(g["sh"]), torch.tensor(g["dirs"])
    for deg in range(4):
        assert torch.allclose(eval_sh(deg, sh, dirs), torch.tensor(g[f"eval_deg{deg}"]), rtol=1e-6, atol=1e-6)
    rgb = torch.tensor(g["rgb"])
    assert torch.allclose(RGB2SH(rgb), torch.tensor(g["rgb2sh"]))
    assert torch.allclose(SH2RGB(RGB2SH(rgb)), torch.tensor(g["sh2rgb"]))


def test_synthetic_camera_matches_reference_matrices():
    from mvs_gaussian_splatting_amd.synthetic import SyntheticCamera, get_world2view2, get_projection_matrix, focal2fov
    g = np.load(os.path.join(GOLD, "camera.npz"))
    for i in range(3):
        R, t = g[f"R{i}"], g[f"t{i}"]
        fx, fy = float(g[f"fx{i}"]), float(g[f"fy{i}"])
        assert math.isclose(focal2fov(fx, 1920), float(g[f"fovx{i}"]), rel_tol=1e-12)
        assert np.allclose(get_world2view2(R, t), g[f"w2v{i}"], atol=1e-7)
        assert np.allclose(get_world2view2(R, t, np.array([0.1, -0.2, 0.3]), 1.5), g[f"w2v_ts{i}"], atol=1e-6)
        proj = get_projection_matrix(0.01, 100.0, float(g[f"fovx{i}"]), float(g[f"fovy{i}"]))
        assert np.allclose(proj.numpy(), g[f"proj{i}"], atol=1e-7)
        cam = SyntheticCamera(1920, 1080, fx, fy, R=R, T=t)
        assert np.allclose(cam.full_proj_transform.numpy(), g[f"full{i}"], rtol=1e-5, atol=1e-6)
        assert np.allclose(cam.camera_center.numpy(), g[f"center{i}"], rtol=1e-5, atol=1e-5)
        # the oracle's projection of points == geom_transform_points of the reference
        from oracle.rasterizer_ref import _xform4
        pts = torch.tensor(g[f"pts{i}"])
        ph = _xform4(pts, cam.full_proj_transform)
        proj_pts = torch.stack([ph[0], ph[1], ph[2]], dim=1) / (ph[3] + 1e-7)[:, None]
        assert np.allclose(proj_pts.numpy(), g[f"pts_proj{i}"], rtol=1e-4, atol=1e-5)


def test_l1_loss_matches_reference():
    from oracle import l1_loss_ref
    g = np.load(os.path.join(GOLD, "loss.npz"))
    a = torch.tensor(g["a"], requires_grad=True)
    b = torch.tensor(g["b"])
    l = l1_loss_ref(a, b)
    assert math.isclose(l.item(), float(g["l1"]), rel_tol=1e-6)
    (ga,) = torch.autograd.grad(l, a)
    assert torch.allclose(ga, torch.tensor(g["l1_grad"]))


def test_splat2d_matches_reference_functions():
    """BASELINE config 1 (plumbing): 128x128 dense 2D splat + combined loss + gradients."""
    from oracle.splat2d_ref import splat2d_ref, combined_loss_ref
    g = np.load(os.path.join(GOLD, "splat2d.npz"))
    t = lambda k: torch.tensor(g[k]).requires_grad_(True)  # noqa: E731
    sx, sy, rho, coords, col = t("sx"), t("sy"), t("rho"), t("coords"), t("colours")
    img = splat2d_ref(int(g["K"]), sx, sy, rho, coords, col, tuple(int(v) for v in g["size"]))
    assert img.shape == (128, 128, 3)
    assert float((img.detach() - torch.tensor(g["image"])).abs().max()) < 2e-5
    loss = combined_loss_ref(img, torch.tensor(g["target"]), 0.2)
    assert math.isclose(loss.item(), float(g["loss"]), rel_tol=1e-5)
    grads = torch.autograd.grad(loss, [sx, sy, rho, coords, col])
    for got, k in zip(grads, ["g_sx", "g_sy", "g_rho", "g_coords", "g_colours"]):
        ref = torch.tensor(g[k])
        assert float((got - ref).abs().max()) <= 1e-4 * float(ref.abs().max()) + 1e-9, k


def test_splat2d_full_config1_matches_reference_functions():
    """BASELINE config 1 at its real size (N = 1000, K = 101, 128 x 128, lambda 0.2): oracle vs the reference's own
    function bodies (tests/golden/make_golden_splat2d.py)."""
    from oracle.splat2d_ref import splat2d_ref, combined_loss_ref
    g = np.load(os.path.join(GOLD, "splat2d_c1.npz"))
    t = lambda k: torch.tensor(g[k]).requires_grad_(True)  # noqa: E731
    sx, sy, rho, coords, col = t("sx"), t("sy"), t("rho"), t("coords"), t("colours")
    img = splat2d_ref(int(g["K"]), sx, sy, rho, coords, col, tuple(int(v) for v in g["size"]))
    assert float((img.detach() - torch.tensor(g["image"])).abs().max()) < 2e-5
    loss = combined_loss_ref(img, torch.tensor(g["target"].astype(np.float32)), 0.2)
    assert math.isclose(loss.item(), float(g["loss"]), rel_tol=1e-5)
    grads = torch.autograd.grad(loss, [sx, sy, rho, coords, col])
    for got, k in zip(grads, ["g_sx", "g_sy", "g_rho", "g_coords", "g_colours"]):
        ref = torch.tensor(g[k])
        assert float((got - ref).abs().max()) <= 1e-4 * float(ref.abs().max()) + 1e-9, k


# ---------------------------------------------------------------------------------------------------
# preprocess sub-steps the reference holds in Python (tests/golden/make_golden_preprocess.py)
# ---------------------------------------------------------------------------------------------------
def golden_preprocess_scene(deg):
    """The fixture's cloud with a camera at the fixture's camera_center looking down +z."""
    from mvs_gaussian_splatting_amd.synthetic import SyntheticCamera
    g = np.load(os.path.join(GOLD, "preprocess.npz"))
    cc = g["camera_center"].astype(np.float64)
    cam = SyntheticCamera(320, 200, 260.0, 250.0, R=np.eye(3), T=-cc)
    assert np.allclose(cam.camera_center.numpy(), g["camera_center"], atol=1e-6)
    t = lambda k: torch.tensor(g[k])  # noqa: E731
    rot = torch.nn.functional.normalize(t("rotation_raw"))          # what get_rotation hands the rasterizer
    return g, cam, t("xyz"), t("features"), t("opacity"), t("scaling"), rot


def test_oracle_cov3d_matches_reference_build_scaling_rotation():
    """scene/gaussian_model.py:28-32 through the reference's own build_scaling_rotation / strip_symmetric."""
    from oracle import build_cov3d_ref
    g, cam, xyz, feats, op, scaling, rot = golden_preprocess_scene(3)
    for mod in (1.0, 1.7):
        want = torch.tensor(g[f"cov3D_mod{mod}"])
        got = build_cov3d_ref(scaling, mod, rot)
        scale = want.abs().max(dim=1, keepdim=True).values
        assert float(((got - want).abs() / scale).max()) <= 2e-6, mod


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_oracle_sh_to_rgb_with_clamp_mask_matches_reference(deg):
    """gaussian_renderer/__init__.py:76-80: clamp_min(eval_sh(...) + 0.5, 0) and the `< 0` mask the backward uses."""
    from conftest import make_settings
    from oracle import preprocess_ref
    g, cam, xyz, feats, op, scaling, rot = golden_preprocess_scene(deg)
    pre = preprocess_ref(xyz, op, make_settings(cam, torch.zeros(3), deg), shs=feats, scales=scaling, rotations=rot)
    keep = pre["keep"]
    gid = pre["idx"][keep]
    assert gid.numel() > 400                                           # most of the cloud is in view
    want, mask = torch.tensor(g[f"colors_deg{deg}"])[gid], torch.tensor(g[f"clamped_deg{deg}"])[gid]
    got, got_mask = pre["v_rgb"][keep], pre["v_clamped"][keep]
    raw = want.abs() < 1e-6                                            # channels sitting on the clamp may flip
    assert float((got - want).abs().max()) <= 2e-6
    assert torch.equal(got_mask.bool() | raw, mask | raw)
    assert int(mask.sum()) > 0
