"""The forward's 4x4 mini-block visit lists: an (instance, mini-block) pair is dropped only when
every pixel of the mini-block fails alpha >= 1/255 in the evaluation's own float32 arithmetic.  With the cull switched
off (GsrParams.debug_flags & GSR_DEBUG_NO_MINIBLOCK_CULL: every staged instance enters all 16 lists of its tile) the colour, the final
transmittance and the last-contributor index of every pixel must be BIT-IDENTICAL -- on random clouds, on the
rare-branch soup (screen-filling, sub-pixel, needle, guard-band, near-plane, opaque and transparent Gaussians) and on
elongated splats at every orientation, which is what the row-span test (xc(dy) +- hw(dy), leftmost / rightmost rows) has
to get right."""
import math

import numpy as np
import pytest
import torch

from conftest import small_scene

pytestmark = pytest.mark.gpu


def _both(dev, monkeypatch, model, cam, bg, deg, scale_modifier=1.0):
    from gpu_util import forward_with_state, product_settings
    st = product_settings(cam, bg, deg, dev, scale_modifier=scale_modifier)
    outs = []
    from mvs_gaussian_splatting_amd import _lib, rasterizer
    for flags in (0, _lib.DEBUG_NO_MINIBLOCK_CULL):
        monkeypatch.setattr(rasterizer, "_debug_flags_value", flags)
        outs.append(forward_with_state(dev, st, model.get_xyz, model.get_opacity, shs=model.get_features,
                                       scales=model.get_scaling, rotations=model.get_rotation, binning_mode=2, want_stats=True))
    monkeypatch.setattr(rasterizer, "_debug_flags_value", 0)
    # the switch is live: without the cull every instance sits in all 16 lists of its tile (mini-blocks that lie outside
    # the image or are saturated walk nothing, hence <=)
    on, off = outs[0]["stats"], outs[1]["stats"]
    assert on["instances"] == off["instances"] and 0.5 * 16 * off["instances"] < off["pairs"] <= 16 * off["instances"]
    assert on["pairs"] < off["pairs"]            # (the rare-branch soup is mostly screen-filling splats: 9 % dropped)
    return outs


def _identical(a, b):
    assert torch.equal(a["color"], b["color"])
    assert torch.equal(a["final_T"], b["final_T"])
    assert torch.equal(a["n_contrib"], b["n_contrib"])


@pytest.mark.parametrize("seed,scale", [(0, 0.02), (1, 0.06), (2, 0.2), (3, 0.006)])
def test_random_clouds(gpu_device, monkeypatch, seed, scale):
    model, cam, bg, _ = small_scene(P=4000, sh_degree=1, width=243, height=139, scale=scale, seed=seed, view=seed)
    a, b = _both(gpu_device, monkeypatch, model, cam, torch.tensor([0.2, 0.3, 0.1]), 1)
    assert int((a["n_contrib"] > 0).sum()) > 1000
    _identical(a, b)


def test_rare_branch_soup(gpu_device, monkeypatch):
    from test_gpu_parity import _stress_model
    from mvs_gaussian_splatting_amd.synthetic import orbit_camera
    model = _stress_model()
    cam = orbit_camera(1, 8, 208, 136, 120.0, 120.0)
    a, b = _both(gpu_device, monkeypatch, model, cam, torch.tensor([0.2, 0.4, 0.1]), 2, scale_modifier=1.3)
    _identical(a, b)


@pytest.mark.parametrize("ratio", [4.0, 20.0, 60.0])
def test_needles_at_every_orientation(gpu_device, monkeypatch, ratio):
    """Elongated splats rotated through all orientations, opacities from just above 1/255 to 1: the tilted ellipse's
    row spans, its leftmost / rightmost rows and the inflated margins are what decides which mini-blocks are kept."""
    model, cam, bg, _ = small_scene(P=3000, sh_degree=0, width=200, height=120, scale=0.05, seed=5)
    g = torch.Generator().manual_seed(17)
    P = 3000
    long_axis = torch.exp(torch.rand(P, generator=g) * math.log(8.0) + math.log(0.03))
    model._scaling[:, 0] = torch.log(long_axis)
    model._scaling[:, 1] = torch.log(long_axis / ratio)
    model._scaling[:, 2] = torch.log(long_axis / ratio)
    ang = torch.rand(P, generator=g) * math.pi                       # rotation about the view axis: every orientation
    model._rotation[:] = torch.stack([torch.cos(ang / 2), torch.zeros(P), torch.zeros(P), torch.sin(ang / 2)], dim=1)
    model._opacity[:] = torch.logit(torch.exp(torch.rand(P, 1, generator=g) * (math.log(0.999) - math.log(0.0045)) + math.log(0.0045)))
    a, b = _both(gpu_device, monkeypatch, model, cam, torch.tensor([0.0, 0.0, 0.0]), 0)
    assert int((a["n_contrib"] > 0).sum()) > 5000
    _identical(a, b)


def _grads(dev, monkeypatch, flags, model, cam, bg, deg, target, scale_modifier=1.0):
    from gpu_util import grads_product, product_settings
    from mvs_gaussian_splatting_amd import rasterizer
    monkeypatch.setattr(rasterizer, "_debug_flags_value", flags)
    st = product_settings(cam, bg, deg, dev, scale_modifier=scale_modifier)
    g, col = grads_product(dev, model, st, target.to(dev), torch.ones_like(target).to(dev))
    monkeypatch.setattr(rasterizer, "_debug_flags_value", 0)
    return g, col


@pytest.mark.parametrize("which", ["cloud", "dense", "soup"])
def test_backward_does_not_depend_on_the_lists(gpu_device, monkeypatch, which):
    """The backward walks the forward's (instance, mini-block) pairs.  With the cull off every instance sits in all 16
    lists of its tile: every round overflows the 26 list positions a pass can hold and is done in three passes, every
    instance sums 16 slots instead of two or three.  The extra pairs contribute exact zeros and an instance's slots are
    summed in mini-block order whatever else shares its round, so every gradient must come out BIT-IDENTICAL."""
    from mvs_gaussian_splatting_amd import _lib
    if which == "soup":
        from test_gpu_parity import _stress_model
        from mvs_gaussian_splatting_amd.synthetic import orbit_camera
        model, deg, sm = _stress_model(), 2, 1.3
        cam = orbit_camera(1, 8, 208, 136, 120.0, 120.0)
        bg = torch.tensor([0.2, 0.4, 0.1])
    else:
        # "dense": footprints of tens of pixels, lists of more than 26 entries with the cull ON as well
        model, cam, bg, _ = small_scene(P=6000, sh_degree=1, width=243, height=139, scale=0.02 if which == "cloud" else 0.15,
                                        seed=11, view=2)
        deg, sm = 1, 1.0
        bg = torch.tensor([0.1, 0.2, 0.3])
    target = torch.rand(3, cam.image_height, cam.image_width, generator=torch.Generator().manual_seed(5))
    ga, ca = _grads(gpu_device, monkeypatch, 0, model, cam, bg, deg, target, sm)
    gb, cb = _grads(gpu_device, monkeypatch, _lib.DEBUG_NO_MINIBLOCK_CULL, model, cam, bg, deg, target, sm)
    assert torch.equal(ca, cb)
    assert float(ga["xyz"].abs().max()) > 0
    for k in ga:
        assert torch.equal(ga[k], gb[k]), k
