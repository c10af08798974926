"""GSR_BINNING_TWO_LEVEL_CULLED: the two-level binning minus the (Gaussian, tile) instances whose tile the
alpha >= 1/255 ellipse provably cannot reach.

What is tested (through the C ABI / the product operator):
* colour, final transmittance, radii and every gradient are BIT-IDENTICAL to the un-culled two-level mode (whose lists
  are bit-exact against the oracle, tests/test_gpu_parity.py);
* the culled lists are order-preserving sub-lists of the un-culled ones;
* every dropped instance is dead in the ORACLE too: evaluated in float64 from the oracle's own preprocess results, no
  pixel of the tile reaches alpha >= 1/255 (so upstream's per-pixel alpha test would have rejected all 256 of them).
"""
import math

import numpy as np
import pytest
import torch

from conftest import make_settings, small_scene

pytestmark = pytest.mark.gpu
TWO_LEVEL, CULLED = 0, 2


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _states(dev, model, cam, bg, deg, scale_modifier=1.0):
    from gpu_util import forward_with_state, product_settings
    st = product_settings(cam, bg, deg, dev, scale_modifier=scale_modifier)
    return [forward_with_state(dev, st, model.get_xyz, model.get_opacity, shs=model.get_features,
                               scales=model.get_scaling, rotations=model.get_rotation, binning_mode=m)
            for m in (TWO_LEVEL, CULLED)]


def _check_sublist_and_dead(o0, o2, model, cam, bg, deg, scale_modifier=1.0):
    """-> fraction of instances dropped."""
    from oracle import preprocess_ref
    assert torch.equal(o0["color"], o2["color"]) and torch.equal(o0["final_T"], o2["final_T"])
    assert torch.equal(o0["radii"], o2["radii"])
    # identify an instance by (key, gaussian): keys are tile << 32 | depth bits
    id0 = np.stack([o0["keys"], o0["point_list"].astype(np.uint64)], axis=1)
    id2 = np.stack([o2["keys"], o2["point_list"].astype(np.uint64)], axis=1)
    v0 = np.ascontiguousarray(id0).view([("k", np.uint64), ("g", np.uint64)]).ravel()
    v2 = np.ascontiguousarray(id2).view([("k", np.uint64), ("g", np.uint64)]).ravel()
    kept = np.isin(v0, v2)
    assert int(kept.sum()) == v2.size and np.array_equal(v0[kept], v2)            # an order-preserving sub-list
    assert o2["R"] == v2.size and np.array_equal(np.bincount(o2["point_list"], minlength=o2["tiles"].size), o2["tiles"])
    # the dropped instances, judged by the oracle in float64
    st = make_settings(cam, bg, deg, scale_modifier=scale_modifier)
    pre = preprocess_ref(model.get_xyz, model.get_opacity, st, shs=model.get_features, scales=model.get_scaling,
                         rotations=model.get_rotation)
    drop_tile = (o0["keys"][~kept] >> np.uint64(32)).astype(np.int64)
    drop_g = o0["point_list"][~kept].astype(np.int64)
    if drop_g.size:
        gx = (int(st.image_width) + 15) // 16
        slot_of = np.full(model.get_xyz.shape[0], -1, dtype=np.int64)
        slot_of[pre["idx"].numpy()] = np.arange(pre["idx"].numel())
        sl = slot_of[drop_g]
        assert np.all(sl >= 0)
        xy = pre["v_xy"].double().numpy()[sl]
        con = pre["v_conic"].double().numpy()[sl]
        op = pre["v_opacity"].double().numpy().reshape(-1)[sl]
        px = (drop_tile % gx)[:, None] * 16 + (np.arange(256) % 16)[None, :]
        py = (drop_tile // gx)[:, None] * 16 + (np.arange(256) // 16)[None, :]
        dx, dy = xy[:, 0:1] - px, xy[:, 1:2] - py
        power = -0.5 * (con[:, 0:1] * dx * dx + con[:, 2:3] * dy * dy) - con[:, 1:2] * dx * dy
        alpha = np.where(power > 0, 0.0, np.minimum(0.99, op[:, None] * np.exp(np.minimum(power, 0.0))))
        worst = float(alpha.max())
        assert worst < 1.0 / 255.0, f"a dropped instance reaches alpha {worst} >= 1/255"
    return 1.0 - v2.size / max(v0.size, 1)


@pytest.mark.parametrize("deg,P,w,h,scale", [(3, 6000, 320, 176, 0.01), (0, 4000, 200, 200, 0.02), (1, 3000, 97, 131, 0.03),
                                             (2, 1500, 64, 48, 0.05)])
def test_culled_lists_are_dead_sublists_and_pixels_identical(dev, deg, P, w, h, scale):
    model, cam, bg, _ = small_scene(P=P, sh_degree=deg, width=w, height=h, scale=scale)
    o0, o2 = _states(dev, model, cam, bg, deg)
    frac = _check_sublist_and_dead(o0, o2, model, cam, bg, deg)
    assert frac > 0.05, f"scene does not exercise the culling ({frac:.3f} dropped)"


def test_culled_binning_on_the_rare_branch_scene(dev):
    """Screen-filling (> 32 tiles: never culled), sub-pixel, needle, guard-band, near-plane, opaque and alpha < 1/255
    Gaussians (the latter lose every instance but keep their radius)."""
    from test_gpu_parity import _stress_model
    from mvs_gaussian_splatting_amd.synthetic import orbit_camera
    model = _stress_model()
    cam = orbit_camera(1, 8, 208, 136, 120.0, 120.0)
    bg = torch.tensor([0.2, 0.4, 0.1])
    o0, o2 = _states(dev, model, cam, bg, 2, scale_modifier=1.3)
    _check_sublist_and_dead(o0, o2, model, cam, bg, 2, scale_modifier=1.3)
    n = model._xyz.shape[0] // 8
    transparent = slice(6 * n, 7 * n)
    assert int(o2["tiles"][transparent].sum()) == 0 and int(o0["tiles"][transparent].sum()) > 0
    assert torch.equal(o0["radii"][transparent], o2["radii"][transparent])
    assert o2["V"] < o0["V"]


def _train_step(dev, monkeypatch, mode, model, cam, bg, target, fused):
    from mvs_gaussian_splatting_amd import render, l1_loss
    from mvs_gaussian_splatting_amd.synthetic import PipelineParams
    from mvs_gaussian_splatting_amd import rasterizer
    monkeypatch.setattr(rasterizer, "_binning_mode_value", rasterizer._binning_from_name(mode))
    for p in model.parameters():
        p.grad = None
    pipe = PipelineParams()
    pipe.fuse_activations = fused
    pkg = render(cam, model, pipe, bg)
    l1_loss(pkg["render"], target).backward()
    return (pkg["render"].detach().clone(), pkg["radii"].clone(), pkg["viewspace_points"].grad.detach().clone(),
            [p.grad.detach().clone() for p in model.parameters()], int(rasterizer.frame_counts(pkg["render"])[0]))


@pytest.mark.parametrize("fused", [True, False])
def test_culled_gradients_bit_identical_small_and_stress(dev, monkeypatch, fused):
    from test_gpu_parity import _stress_model
    from mvs_gaussian_splatting_amd.synthetic import orbit_camera
    scenes = [small_scene(P=5000, sh_degree=3, width=320, height=176, scale=0.012)[:3]]
    scenes.append((_stress_model(), orbit_camera(1, 8, 208, 136, 120.0, 120.0), torch.tensor([0.2, 0.4, 0.1])))
    for model, cam, bg in scenes:
        model.to(dev); cam.to(dev)
        bg = bg.to(dev)
        target = torch.rand(3, cam.image_height, cam.image_width, device=dev)
        for p in model.parameters():
            p.requires_grad_(True)
        a = _train_step(dev, monkeypatch, "two_level", model, cam, bg, target, fused)
        b = _train_step(dev, monkeypatch, "culled", model, cam, bg, target, fused)
        assert b[4] < a[4]
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
        assert all(torch.equal(x, y) for x, y in zip(a[3], b[3]))


def test_culled_full_size_C4_bit_identical_and_smaller(dev, monkeypatch):
    """BASELINE config C4 (6 M Gaussians, 1080p): same image and gradients bit for bit, fewer instances."""
    from mvs_gaussian_splatting_amd.synthetic import CONFIGS, make_scene
    model, cam, bg, target = make_scene(CONFIGS["C4"])
    model.to(dev); cam.to(dev)
    bg, target = bg.to(dev), target.to(dev)
    for p in model.parameters():
        p.requires_grad_(True)
    a = _train_step(dev, monkeypatch, "two_level", model, cam, bg, target, True)
    b = _train_step(dev, monkeypatch, "culled", model, cam, bg, target, True)
    assert b[4] < 0.8 * a[4], (a[4], b[4])
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert all(torch.equal(x, y) for x, y in zip(a[3], b[3]))


@pytest.mark.parametrize("W,H,P,f", [(3840, 2160, 1_000_000, 2400.0), (7680, 4320, 300_000, 4800.0), (48, 16, 5000, 30.0)])
def test_all_binning_modes_agree_at_4k(dev, W, H, P, f):
    """3840x2160 (32 400 tiles: 15 tile bits -> 8+7-bit tile passes, 47-bit keys -> 6 passes in keys64 mode), 7680x4320
    (129 600 tiles, 17 bits) and a 3-tile image: the three binning modes give the same image bit for bit; two_level
    and keys64 the same lists."""
    from gpu_util import forward_with_state, product_settings
    from mvs_gaussian_splatting_amd.synthetic import SceneConfig, make_scene
    cfg = SceneConfig("big", P, 1, W, H, f, f, math.log(0.012 if W > 100 else 0.2))
    model, cam, bg, _ = make_scene(cfg)
    st = product_settings(cam, bg, 1, dev)
    outs = [forward_with_state(dev, st, model.get_xyz, model.get_opacity, shs=model.get_features,
                               scales=model.get_scaling, rotations=model.get_rotation, binning_mode=m) for m in (0, 1, 2)]
    assert outs[0]["R"] == outs[1]["R"] > outs[2]["R"] > 0
    assert np.array_equal(outs[0]["keys"], outs[1]["keys"]) and np.array_equal(outs[0]["point_list"], outs[1]["point_list"])
    assert np.array_equal(outs[0]["ranges"], outs[1]["ranges"])
    for o in outs[1:]:
        assert torch.equal(outs[0]["color"], o["color"]) and torch.equal(outs[0]["final_T"], o["final_T"])
        assert torch.equal(outs[0]["radii"], o["radii"])
    k = outs[2]["keys"]
    assert np.all(k[1:] >= k[:-1]) and int(k[-1] >> np.uint64(32)) < ((W + 15) // 16) * ((H + 15) // 16)
