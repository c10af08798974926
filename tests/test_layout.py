"""The optional Morton layout step (``mvs_gaussian_splatting_amd/layout.py``): the permutation, and the optimizer surgery
(the pattern of the reference's ``_prune_optimizer``, ``scene/gaussian_model.py:401-417``).  CPU only."""
import torch
from torch import nn

from mvs_gaussian_splatting_amd.densify import GROUP_ATTR
from mvs_gaussian_splatting_amd.layout import morton_permutation, reorder_gaussians_


def test_morton_permutation_is_a_stable_permutation_that_groups_octants():
    g = torch.Generator().manual_seed(3)
    xyz = torch.rand(5000, 3, generator=g) * torch.tensor([6.0, 3.4, 3.0]) + torch.tensor([-3.0, -1.7, 4.5])
    perm = morton_permutation(xyz)
    assert torch.equal(torch.sort(perm).values, torch.arange(5000))
    lo, hi = xyz.min(0).values, xyz.max(0).values
    q = ((xyz - lo) / (hi - lo) * 1023.0).clamp(0, 1023).long()
    octant = ((q[:, 2] >> 9) << 2 | (q[:, 1] >> 9) << 1 | (q[:, 0] >> 9))[perm]
    assert torch.all(octant[1:] >= octant[:-1])                  # the top bit of each axis is the coarsest split
    # neighbours along the curve are neighbours in space: mean distance of consecutive points far below random order's
    d_sorted = (xyz[perm][1:] - xyz[perm][:-1]).norm(dim=1).mean()
    d_random = (xyz[1:] - xyz[:-1]).norm(dim=1).mean()
    assert d_sorted < 0.15 * d_random
    # stable: equal cells keep index order; degenerate clouds do not divide by zero
    same = torch.zeros(7, 3)
    assert torch.equal(morton_permutation(same), torch.arange(7))
    assert morton_permutation(torch.zeros(0, 3)).numel() == 0


class _Model:
    pass


def _model(P, seed, with_optimizer):
    g = torch.Generator().manual_seed(seed)
    m = _Model()
    shapes = {"xyz": (P, 3), "f_dc": (P, 1, 3), "f_rest": (P, 15, 3), "opacity": (P, 1), "scaling": (P, 3), "rotation": (P, 4)}
    for k, a in GROUP_ATTR.items():
        setattr(m, a, nn.Parameter(torch.randn(*shapes[k], generator=g)))
    m.xyz_gradient_accum = torch.rand(P, 1, generator=g)
    m.denom = torch.rand(P, 1, generator=g)
    m.max_radii2D = torch.rand(P, generator=g)
    if with_optimizer:
        m.optimizer = torch.optim.Adam([{"params": [getattr(m, a)], "lr": 1e-2 * (i + 1), "name": k}
                                        for i, (k, a) in enumerate(GROUP_ATTR.items())], lr=0.0, eps=1e-15)
    return m


def _step(m, grads):
    for k, a in GROUP_ATTR.items():
        getattr(m, a).grad = grads[k].clone()
    m.optimizer.step()


def test_reordering_moves_adam_moments_with_their_gaussians():
    P = 257
    a, b = _model(P, 5, True), _model(P, 5, True)
    g = torch.Generator().manual_seed(9)
    g1 = {k: torch.randn_like(getattr(a, at), ) for k, at in GROUP_ATTR.items()}
    g2 = {k: torch.randn_like(getattr(a, at)) for k, at in GROUP_ATTR.items()}
    _step(a, g1); _step(b, g1)                                   # both hold moments now
    before = {at: getattr(b, at).detach().clone() for at in GROUP_ATTR.values()}
    stats = {at: getattr(b, at).clone() for at in ("xyz_gradient_accum", "denom", "max_radii2D")}
    perm = reorder_gaussians_(b)
    assert not torch.equal(perm, torch.arange(P))
    for at in GROUP_ATTR.values():
        t = getattr(b, at)
        assert isinstance(t, nn.Parameter) and t.requires_grad and t.is_contiguous()
        assert torch.equal(t.detach(), before[at][perm])
        assert any(t is grp["params"][0] for grp in b.optimizer.param_groups)      # the optimizer owns the new tensor
    for at, t in stats.items():
        assert torch.equal(getattr(b, at), t[perm])
    assert len(b.optimizer.state) == len(GROUP_ATTR)
    _step(a, g2); _step(b, {k: v[perm] for k, v in g2.items()})   # the second step sees the first step's moments
    for at in GROUP_ATTR.values():
        assert torch.equal(getattr(b, at).detach(), getattr(a, at).detach()[perm])


def test_reordering_a_model_without_optimizer_keeps_tensor_kinds_and_rejects_bad_permutations():
    import pytest
    m = _model(64, 1, False)
    m._opacity = m._opacity.detach().clone().requires_grad_(True)         # a plain tensor, as synthetic.py's model
    m._rotation = m._rotation.detach().clone()                             # frozen
    ref = m._xyz.detach().clone()
    perm = torch.randperm(64, generator=torch.Generator().manual_seed(2))
    assert torch.equal(reorder_gaussians_(m, perm), perm)
    assert torch.equal(m._xyz.detach(), ref[perm]) and isinstance(m._xyz, nn.Parameter)
    assert not isinstance(m._opacity, nn.Parameter) and m._opacity.requires_grad and not m._rotation.requires_grad
    with pytest.raises(ValueError):
        reorder_gaussians_(m, torch.zeros(64, dtype=torch.int64))
    with pytest.raises(ValueError):
        reorder_gaussians_(m, torch.arange(63))
