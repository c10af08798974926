"""add_densification_stats + the max_radii2D update against the reference's three torch lines
(scene/gaussian_model.py:775-777, train.py:130), restated here on the device tensors:

    max_radii2D[vis] = max(max_radii2D[vis], radii[vis])
    xyz_gradient_accum[vis] += norm(viewspace_points.grad[vis, :2], dim=-1, keepdim=True)
    denom[vis] += 1                                         with vis = radii > 0

denom / max_radii2D and every untouched row are compared bit for bit, the norm to 1e-6 relative.
"""
import types

import pytest
import torch

from conftest import small_scene

pytestmark = pytest.mark.gpu


def _reference_update(accum, denom, max_radii2D, grad, radii):
    vis = radii > 0
    max_radii2D = max_radii2D.clone()
    accum, denom = accum.clone(), denom.clone()
    max_radii2D[vis] = torch.max(max_radii2D[vis], radii[vis].to(max_radii2D.dtype))
    accum[vis] += torch.norm(grad[vis, :2], dim=-1, keepdim=True)
    denom[vis] += 1
    return accum, denom, max_radii2D


@pytest.mark.parametrize("P", [1, 255, 256, 257, 100_003])
def test_densify_stats_matches_reference_lines(gpu_device, P):
    from mvs_gaussian_splatting_amd import add_densification_stats
    dev = gpu_device
    g = torch.Generator().manual_seed(P)
    grad = torch.randn(P, 3, generator=g) * torch.tensor([1e-3, 5e-4, 7.0])     # z is large: it must not enter the norm
    radii = torch.randint(-1, 40, (P,), generator=g, dtype=torch.int32)
    radii[torch.rand(P, generator=g) < 0.4] = 0                                  # invisible rows stay untouched
    accum0 = torch.rand(P, 1, generator=g)
    denom0 = torch.randint(0, 9, (P, 1), generator=g).float()
    maxr0 = torch.randint(0, 30, (P,), generator=g).float()
    model = types.SimpleNamespace(xyz_gradient_accum=accum0.to(dev), denom=denom0.to(dev), max_radii2D=maxr0.to(dev))
    vsp = torch.zeros(P, 3, device=dev, requires_grad=True)
    vsp.grad = grad.to(dev)
    want = _reference_update(model.xyz_gradient_accum, model.denom, model.max_radii2D, vsp.grad, radii.to(dev))
    for rep in range(2):                                                         # two frames accumulate
        add_densification_stats(model, vsp, radii.to(dev))
        if rep == 0:
            want2 = _reference_update(*want, vsp.grad, radii.to(dev))
            got1 = (model.xyz_gradient_accum.clone(), model.denom.clone(), model.max_radii2D.clone())
    for got, ref in ((got1, want), ((model.xyz_gradient_accum, model.denom, model.max_radii2D), want2)):
        assert torch.equal(got[1], ref[1])                                       # denom: exact
        assert torch.equal(got[2], ref[2])                                       # max_radii2D: exact
        assert float(((got[0] - ref[0]).abs() / ref[0].abs().clamp(min=1e-12)).max()) <= 1e-6
        inv = (radii <= 0).to(dev)
        assert torch.equal(got[0][inv], accum0.to(dev)[inv])                     # untouched rows: bit-identical
    assert model.xyz_gradient_accum.shape == (P, 1) and model.max_radii2D.shape == (P,)


def test_densify_stats_after_two_rendered_frames(gpu_device):
    """The train-loop use (train.py:107,130-131): render -> loss -> backward -> stats, two views in a row."""
    from mvs_gaussian_splatting_amd import render, l1_loss, add_densification_stats
    from mvs_gaussian_splatting_amd.synthetic import PipelineParams
    dev = gpu_device
    model, cam, bg, target = small_scene(P=3000, sh_degree=1, width=160, height=96)
    _, cam2, _, _ = small_scene(P=3000, sh_degree=1, width=160, height=96, view=3)
    model.to(dev); cam.to(dev); cam2.to(dev)
    for p in model.parameters():
        p.requires_grad_(True)
    ref = (model.xyz_gradient_accum.clone(), model.denom.clone(), model.max_radii2D.clone())
    seen = torch.zeros(3000, dtype=torch.bool, device=dev)
    for c in (cam, cam2):
        pkg = render(c, model, PipelineParams(), bg.to(dev))
        l1_loss(pkg["render"], target.to(dev)).backward()
        ref = _reference_update(*ref, pkg["viewspace_points"].grad, pkg["radii"])
        add_densification_stats(model, pkg["viewspace_points"], pkg["radii"])
        seen |= pkg["visibility_filter"]
    assert 0 < int(seen.sum()) < 3000
    assert torch.equal(model.denom, ref[1]) and torch.equal(model.max_radii2D, ref[2])
    assert float(((model.xyz_gradient_accum - ref[0]).abs() / ref[0].abs().clamp(min=1e-12)).max()) <= 1e-6
    assert float(model.xyz_gradient_accum[~seen].abs().max()) == 0.0 and float(model.denom[~seen].abs().max()) == 0.0
    assert float(model.xyz_gradient_accum[seen].max()) > 0.0
