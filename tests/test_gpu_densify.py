"""Densification bookkeeping on the HIP path against the CPU restatement of the reference's
``densify_and_prune`` (oracle/densify_ref.py), with the same normal draws injected into both.

Copied rows (kept originals, clones, SH / opacity / rotation of children, Adam moments) must be bit-identical; the
children's computed xyz / scaling agree to 1e-6 relative (exp / log / 3x3 product in float32).  Parity unpinned: the
reference's model class does not import here and ships no fixtures (see the oracle's header).
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling",
        "rotation": "_rotation"}


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


class _Model:
    """The attributes of scene/gaussian_model.py's GaussianModel that densification touches."""

    def __init__(self, P, seed, dev, with_optimizer, sh_rest=15):
        g = torch.Generator().manual_seed(seed)
        self.percent_dense = 0.01
        cpu = {
            "xyz": torch.randn(P, 3, generator=g) * 2.0,
            "f_dc": torch.randn(P, 1, 3, generator=g),
            "f_rest": 0.1 * torch.randn(P, sh_rest, 3, generator=g),
            "opacity": 2.5 * torch.randn(P, 1, generator=g) - 1.0,           # some below sigmoid^-1(0.005) = -5.3
            "scaling": math.log(0.05) + 1.2 * torch.randn(P, 3, generator=g),  # straddles 0.01*extent and 0.1*extent
            "rotation": torch.randn(P, 4, generator=g),
        }
        self.cpu_params = cpu
        for k, a in ATTR.items():
            setattr(self, a, torch.nn.Parameter(cpu[k].to(dev).requires_grad_(True)))
        denom = torch.randint(0, 4, (P, 1), generator=g).float()              # zeros -> 0/0 = NaN -> 0 (:751-752)
        accum = torch.rand(P, 1, generator=g) * 0.0006 * denom
        self.cpu_accum, self.cpu_denom = accum, denom
        self.cpu_radii = torch.rand(P, generator=g) * 40
        self.xyz_gradient_accum, self.denom, self.max_radii2D = accum.to(dev), denom.to(dev), self.cpu_radii.to(dev)
        self.cpu_moments = None
        if with_optimizer:
            groups = [{"params": [getattr(self, ATTR[k])], "lr": 1e-3, "name": k} for k in GROUPS]
            self.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
            for k in GROUPS:                                                   # one step creates exp_avg / exp_avg_sq
                p = getattr(self, ATTR[k])
                p.grad = torch.randn(p.shape, generator=g).to(dev)
            self.optimizer.step()
            self.cpu_params = {k: getattr(self, ATTR[k]).detach().cpu() for k in GROUPS}
            self.cpu_moments = {k: (self.optimizer.state[getattr(self, ATTR[k])]["exp_avg"].cpu(),
                                    self.optimizer.state[getattr(self, ATTR[k])]["exp_avg_sq"].cpu()) for k in GROUPS}


def _run_both(dev, P, seed, with_optimizer, max_screen_size, max_grad=0.0002, min_opacity=0.005, extent=5.0):
    from mvs_gaussian_splatting_amd.densify import densify_and_prune
    from oracle.densify_ref import densify_and_prune_ref, count_split_selected_ref
    m = _Model(P, seed, dev, with_optimizer)
    n_sel = count_split_selected_ref(m.cpu_params, m.cpu_accum.clone(), m.cpu_denom, m.percent_dense, max_grad, extent)
    noise = torch.randn(2 * n_sel, 3, generator=torch.Generator().manual_seed(seed + 100))
    ref = densify_and_prune_ref(m.cpu_params, m.cpu_moments, m.cpu_accum, m.cpu_denom, m.cpu_radii, m.percent_dense,
                                max_grad, min_opacity, extent, max_screen_size, noise)
    info = densify_and_prune(m, max_grad, min_opacity, extent, max_screen_size, noise=noise.to(dev))
    return m, ref, info, n_sel


def _check(m, ref, info, n_sel, with_optimizer):
    rp, rm, raccum, rdenom, rradii, rinfo = ref
    n_out = rp["xyz"].shape[0]
    assert info["points"] == n_out and info["split_selected"] == n_sel == rinfo["split"]
    n_new_computed = 2 * info["children_per_copy"]
    first_child = n_out - n_new_computed
    for k in GROUPS:
        got = getattr(m, ATTR[k]).detach().cpu()
        assert got.shape == rp[k].shape, k
        if k in ("xyz", "scaling"):
            assert torch.equal(got[:first_child], rp[k][:first_child]), k            # copies: bit-identical
            err = (got[first_child:] - rp[k][first_child:]).abs() / rp[k][first_child:].abs().clamp(min=1.0)
            assert n_new_computed == 0 or float(err.max()) <= 1e-6, k
        else:
            assert torch.equal(got, rp[k]), k
        assert isinstance(getattr(m, ATTR[k]), torch.nn.Parameter) and getattr(m, ATTR[k]).requires_grad
    if with_optimizer:
        for group in m.optimizer.param_groups:
            k = group["name"]
            p = group["params"][0]
            assert p is getattr(m, ATTR[k])
            st = m.optimizer.state[p]
            assert torch.equal(st["exp_avg"].cpu(), rm[k][0]) and torch.equal(st["exp_avg_sq"].cpu(), rm[k][1]), k
            assert len(m.optimizer.state) == len(GROUPS)
    for got, want in ((m.xyz_gradient_accum, raccum), (m.denom, rdenom), (m.max_radii2D, rradii)):
        assert got.shape == want.shape and float(got.abs().max()) == 0.0 if got.numel() else got.shape == want.shape


@pytest.mark.parametrize("P,with_optimizer,max_screen_size", [
    (20000, True, 20), (20000, True, None), (5000, False, 20), (257, True, 20), (1, False, None)])
def test_densify_and_prune_matches_reference_restatement(dev, P, with_optimizer, max_screen_size):
    m, ref, info, n_sel = _run_both(dev, P, 7 + P, with_optimizer, max_screen_size)
    _check(m, ref, info, n_sel, with_optimizer)
    if P >= 5000:       # the scene exercises every class
        assert info["cloned"] > 0 and info["children_per_copy"] > 0 and info["kept"] < P
        assert info["split_selected"] >= info["children_per_copy"]


def test_densify_nothing_selected_and_everything_pruned(dev):
    # threshold above every gradient: pure prune
    m, ref, info, n_sel = _run_both(dev, 3000, 3, True, 20, max_grad=1.0)
    assert n_sel == 0 and info["cloned"] == 0
    _check(m, ref, info, n_sel, True)
    # min_opacity above every opacity: nothing survives, tensors become empty
    m, ref, info, n_sel = _run_both(dev, 3000, 4, True, 20, min_opacity=2.0)
    assert info["points"] == 0
    _check(m, ref, info, n_sel, True)
    # the optimizer still steps on the empty / regrown model
    m, ref, info, n_sel = _run_both(dev, 3000, 5, True, None)
    for p in (g["params"][0] for g in m.optimizer.param_groups):
        p.grad = torch.ones_like(p)
    m.optimizer.step()


def test_densify_rejects_cpu_and_bad_noise(dev):
    from mvs_gaussian_splatting_amd import _lib
    from mvs_gaussian_splatting_amd.densify import densify_and_prune
    m = _Model(100, 1, torch.device("cpu"), False)
    with pytest.raises(_lib.GsrError):
        densify_and_prune(m, 0.0002, 0.005, 5.0, 20)
    m = _Model(2000, 1, dev, False)
    with pytest.raises(ValueError, match="noise must be"):
        densify_and_prune(m, 0.0002, 0.005, 5.0, 20, noise=torch.zeros(1, 3, device=dev))


def test_densify_in_spatial_order_is_the_reference_result_permuted(dev):
    """``spatial_order=True`` (this build's extension): the Gaussians, their Adam moments and the reset statistics of the
    reference-order result, stored along the Morton curve of the new positions."""
    from mvs_gaussian_splatting_amd.densify import densify_and_prune
    from mvs_gaussian_splatting_amd.layout import morton_permutation
    from oracle.densify_ref import count_split_selected_ref
    a, b = _Model(3000, 21, dev, True), _Model(3000, 21, dev, True)
    n_sel = count_split_selected_ref(a.cpu_params, a.cpu_accum.clone(), a.cpu_denom, a.percent_dense, 0.0002, 5.0)
    noise = torch.randn(2 * n_sel, 3, generator=torch.Generator().manual_seed(5)).to(dev)
    ia = densify_and_prune(a, 0.0002, 0.005, 5.0, 20, noise=noise)
    ib = densify_and_prune(b, 0.0002, 0.005, 5.0, 20, noise=noise, spatial_order=True)
    assert ia == ib and ia["cloned"] > 0 and ia["split_selected"] > 0
    perm = morton_permutation(a._xyz)
    assert not torch.equal(perm, torch.arange(perm.numel(), device=dev))
    for k in GROUPS:
        pa, pb = getattr(a, ATTR[k]), getattr(b, ATTR[k])
        assert isinstance(pb, torch.nn.Parameter) and pb.requires_grad
        assert torch.equal(pb.detach(), pa.detach()[perm])
        sa, sb = a.optimizer.state[pa], b.optimizer.state[pb]
        assert torch.equal(sb["exp_avg"], sa["exp_avg"][perm]) and torch.equal(sb["exp_avg_sq"], sa["exp_avg_sq"][perm])
    assert b.xyz_gradient_accum.shape == a.xyz_gradient_accum.shape and not b.xyz_gradient_accum.any()
