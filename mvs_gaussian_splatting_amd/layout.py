"""Spatial ordering of the Gaussians in memory (an optional layout step; not part of the reference's path).

The kernels stream the per-Gaussian tensors by index: lane i of a wave handles Gaussian i.  Whether a Gaussian is inside
the frustum, and which tiles it reaches, depends on where it IS -- so when neighbours in memory are neighbours in space,
whole waves are culled together (their 192-byte SH rows are skipped as whole lines instead of leaving half-used lines
behind), the instance lists of a tile point at adjacent 64-byte records and the gradient rows of a wave's Gaussians are
adjacent too.  The reference appends clones and split children at the end of its tensors
(``scene/gaussian_model.py:466-504``), so a model under training is in no spatial order.  ``reorder_gaussians_`` stores
it along a 30-bit Morton curve of the positions, with the Adam-moment surgery of the reference's ``_prune_optimizer``
(``:401-417``: same groups, same state keys); a trainer calls it after ``densify_and_prune`` (every 100 iterations,
``train.py:130-134``).

Results are the unpermuted model's up to the permutation: every per-Gaussian quantity is computed from that Gaussian's
row alone, and a tile's list is ordered by depth.  Only equal-depth ties inside a tile (broken by index, as in the
reference's stable sort) can blend in another order; a frame without such ties is bit-identical, gradients included
(``tests/test_gpu_layout.py``; at 6 M Gaussians the sorted keys are the same array, 882 of 8.2 M list entries swap places
inside equal-key runs and 279 of 2 M pixels differ, by at most 1.2e-3).  Measured on the 6 M-Gaussian bench cloud, whose index order is uniformly random:
forward 0.912 -> 0.846 ms, train step 2.252 -> 2.056 ms (``profiles/r04/layout_morton.txt``).
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from .densify import GROUP_ATTR

STATS_ATTR = ("xyz_gradient_accum", "denom", "max_radii2D")


def _spread3(v: torch.Tensor) -> torch.Tensor:
    """10-bit integers -> their bits spread to every third position (int64)."""
    v = v & 0x3ff
    v = (v | (v << 16)) & 0x030000ff
    v = (v | (v << 8)) & 0x0300f00f
    v = (v | (v << 4)) & 0x030c30c3
    v = (v | (v << 2)) & 0x09249249
    return v


@torch.no_grad()
def morton_permutation(xyz: torch.Tensor) -> torch.Tensor:
    """Indices that sort the points along a 30-bit Morton (Z-order) curve over their bounding box; points of one cell keep
    their order (stable).  ``xyz``: ``[P, 3]`` on any device."""
    p = xyz.detach().float()
    if p.dim() != 2 or p.shape[1] != 3:
        raise ValueError(f"xyz must be [P, 3], got {tuple(p.shape)}")
    if p.shape[0] == 0:
        return torch.empty(0, dtype=torch.int64, device=p.device)
    lo, hi = p.min(dim=0).values, p.max(dim=0).values
    q = ((p - lo) / (hi - lo).clamp(min=1e-12) * 1023.0).clamp(0.0, 1023.0).to(torch.int64)
    code = _spread3(q[:, 0]) | (_spread3(q[:, 1]) << 1) | (_spread3(q[:, 2]) << 2)
    return torch.sort(code, stable=True).indices


@torch.no_grad()
def reorder_gaussians_(model, perm: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Permute every per-Gaussian tensor of a model in place: row i of the result is row ``perm[i]`` of the input
    (default: ``morton_permutation(model._xyz)``).  The model is duck-typed as in ``densify.py``: the six parameter
    tensors, the densification statistics if present, and ``model.optimizer`` (one parameter per named group) if present,
    whose ``exp_avg`` / ``exp_avg_sq`` rows move with their Gaussians.  Returns the permutation."""
    xyz = model._xyz
    P = int(xyz.shape[0])
    if perm is None:
        perm = morton_permutation(xyz)
    perm = perm.to(device=xyz.device, dtype=torch.int64)
    if perm.shape != (P,) or (P and not torch.equal(torch.sort(perm).values, torch.arange(P, device=perm.device))):
        raise ValueError(f"perm must be a permutation of range({P})")
    new = {k: getattr(model, a).detach()[perm].contiguous() for k, a in GROUP_ATTR.items()}
    optimizer = getattr(model, "optimizer", None)
    owned = set()
    if optimizer is not None:
        for group in optimizer.param_groups:
            name = group.get("name")
            if name not in new:
                continue
            old = group["params"][0]
            stored = optimizer.state.get(old, None)
            group["params"][0] = nn.Parameter(new[name].requires_grad_(True))
            if stored is not None:
                for key in ("exp_avg", "exp_avg_sq"):
                    if key in stored:
                        stored[key] = stored[key][perm].contiguous()
                del optimizer.state[old]
                optimizer.state[group["params"][0]] = stored
            new[name] = group["params"][0]
            owned.add(name)
    for k, a in GROUP_ATTR.items():
        t, old = new[k], getattr(model, a)
        if k not in owned:
            t = nn.Parameter(t, requires_grad=old.requires_grad) if isinstance(old, nn.Parameter) \
                else t.requires_grad_(old.requires_grad)
        setattr(model, a, t)
    for a in STATS_ATTR:
        t = getattr(model, a, None)
        if isinstance(t, torch.Tensor) and t.dim() >= 1 and t.shape[0] == P:
            setattr(model, a, t[perm].contiguous())
    return perm
