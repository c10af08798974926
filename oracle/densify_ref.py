"""CPU restatement of the reference's densification bookkeeping -- TEST INFRASTRUCTURE ONLY.

Follows ``scene/gaussian_model.py`` of the reference step by step, with the same tensor operations (boolean-mask
indexing, ``cat``, ``repeat``) on plain dicts instead of ``nn.Parameter`` / ``torch.optim.Adam`` objects:

* ``densify_and_prune``      :750-772   -> :func:`densify_and_prune_ref`
* ``densify_and_clone``      :580-610   (plain branch: ``grow_dir`` / ``continous_dir`` / ``learn_split_*`` are the
  fork's default-off options, SURVEY §2, and are not restated)
* ``densify_and_split``      :506-578   (``symmetric_split`` False, ``arguments/__init__.py:62``)
* ``densification_postfix``  :466-504   (statistics are reset to zeros there)
* ``prune_points`` / ``_prune_optimizer`` / ``cat_tensors_to_optimizer``  :401-464
* ``build_rotation``         ``utils/general_utils.py:78-99``

PARITY STATUS: unpinned (the reference model class cannot be imported here -- ``plyfile`` / ``simple_knn`` are absent --
and the reference holds no fixtures for it); the restatement is line-by-line from the text.

The reference draws ``torch.normal(mean=0, std=stds)`` inside ``densify_and_split`` (:537-539); here the standard-normal
draws are an argument (``noise [N * n_selected, 3]``, used as ``stds * noise``) so that two implementations can be
compared on the same samples.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")     # training_setup order, :245-252


def build_rotation_ref(r: torch.Tensor) -> torch.Tensor:
    norm = torch.sqrt(r[:, 0] * r[:, 0] + r[:, 1] * r[:, 1] + r[:, 2] * r[:, 2] + r[:, 3] * r[:, 3])
    q = r / norm[:, None]
    R = torch.zeros((q.size(0), 3, 3), dtype=r.dtype)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R[:, 0, 0] = 1 - 2 * (y * y + z * z)
    R[:, 0, 1] = 2 * (x * y - r * z)
    R[:, 0, 2] = 2 * (x * z + r * y)
    R[:, 1, 0] = 2 * (x * y + r * z)
    R[:, 1, 1] = 1 - 2 * (x * x + z * z)
    R[:, 1, 2] = 2 * (y * z - r * x)
    R[:, 2, 0] = 2 * (x * z - r * y)
    R[:, 2, 1] = 2 * (y * z + r * x)
    R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


class _State:
    """The slice of GaussianModel the densification touches: raw parameters, Adam moments (or None), statistics."""

    def __init__(self, params: Dict[str, torch.Tensor], moments: Optional[Dict[str, Tuple[torch.Tensor, torch.Tensor]]],
                 accum: torch.Tensor, denom: torch.Tensor, max_radii2D: torch.Tensor, percent_dense: float):
        self.p = {k: v.clone() for k, v in params.items()}
        self.m = None if moments is None else {k: (a.clone(), b.clone()) for k, (a, b) in moments.items()}
        self.accum, self.denom, self.max_radii2D = accum.clone(), denom.clone(), max_radii2D.clone()
        self.percent_dense = percent_dense

    # activations, :27-42,151-183
    def scaling(self):
        return torch.exp(self.p["scaling"])

    def opacity(self):
        return torch.sigmoid(self.p["opacity"])

    def n(self):
        return self.p["xyz"].shape[0]

    def prune_points(self, mask):                                   # :424-449 with _prune_optimizer :401-422
        valid = ~mask
        for k in GROUPS:
            self.p[k] = self.p[k][valid]
            if self.m is not None:
                self.m[k] = (self.m[k][0][valid], self.m[k][1][valid])
        self.accum = self.accum[valid]
        self.denom = self.denom[valid]
        self.max_radii2D = self.max_radii2D[valid]

    def postfix(self, new):                                         # :466-504 with cat_tensors_to_optimizer :451-472
        for k in GROUPS:
            ext = new[k]
            if self.m is not None:
                self.m[k] = (torch.cat((self.m[k][0], torch.zeros_like(ext)), dim=0),
                             torch.cat((self.m[k][1], torch.zeros_like(ext)), dim=0))
            self.p[k] = torch.cat((self.p[k], ext), dim=0)
        n = self.n()
        dt = self.accum.dtype
        self.accum = torch.zeros((n, 1), dtype=dt)
        self.denom = torch.zeros((n, 1), dtype=dt)
        self.max_radii2D = torch.zeros((n,), dtype=dt)

    def densify_and_clone(self, grads, grad_threshold, scene_extent):           # :580-610
        sel = torch.where(torch.norm(grads, dim=-1) >= grad_threshold, True, False)
        sel = torch.logical_and(sel, torch.max(self.scaling(), dim=1).values <= self.percent_dense * scene_extent)
        self.postfix({k: self.p[k][sel] for k in GROUPS})
        return int(sel.sum())

    def densify_and_split(self, grads, grad_threshold, scene_extent, noise, N=2):   # :506-578
        n_init = self.n()
        padded = torch.zeros((n_init,), dtype=grads.dtype)
        padded[:grads.shape[0]] = grads.squeeze()
        sel = torch.where(padded >= grad_threshold, True, False)
        sel = torch.logical_and(sel, torch.max(self.scaling(), dim=1).values > self.percent_dense * scene_extent)
        stds = self.scaling()[sel].repeat(N, 1)
        if noise.shape[0] != stds.shape[0]:
            raise ValueError(f"noise must have {stds.shape[0]} rows (N * selected), got {noise.shape[0]}")
        samples = stds * noise.to(stds.dtype)                       # torch.normal(mean=0, std=stds), :537-539
        rots = build_rotation_ref(self.p["rotation"][sel]).repeat(N, 1, 1)
        new = {
            "xyz": torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + self.p["xyz"][sel].repeat(N, 1),
            "scaling": torch.log(self.scaling()[sel].repeat(N, 1) / (0.8 * N)),
            "rotation": self.p["rotation"][sel].repeat(N, 1),
            "f_dc": self.p["f_dc"][sel].repeat(N, 1, 1),
            "f_rest": self.p["f_rest"][sel].repeat(N, 1, 1),
            "opacity": self.p["opacity"][sel].repeat(N, 1),
        }
        self.postfix(new)
        prune_filter = torch.cat((sel, torch.zeros(N * int(sel.sum()), dtype=torch.bool)))
        self.prune_points(prune_filter)
        return int(sel.sum())


def count_split_selected_ref(params, accum, denom, percent_dense, max_grad, extent) -> int:
    """How many Gaussians ``densify_and_split`` will select (the row count of ``noise`` is twice this)."""
    grads = accum / denom
    grads[grads.isnan()] = 0.0
    sc = torch.exp(params["scaling"]).max(dim=1).values
    return int(((grads.squeeze(-1) >= max_grad) & (sc > percent_dense * extent)).sum())


def densify_and_prune_ref(params, moments, accum, denom, max_radii2D, percent_dense, max_grad, min_opacity, extent,
                          max_screen_size, noise):
    """:750-772 (plain branch).  Returns (params, moments, accum, denom, max_radii2D, info)."""
    st = _State(params, moments, accum, denom, max_radii2D, percent_dense)
    grads = st.accum / st.denom
    grads[grads.isnan()] = 0.0
    n_clone = st.densify_and_clone(grads, max_grad, extent)
    n_split = st.densify_and_split(grads, max_grad, extent, noise)
    prune_mask = (st.opacity() < min_opacity).squeeze(-1)
    if max_screen_size:
        big_points_vs = st.max_radii2D > max_screen_size        # max_radii2D was reset by the postfix above (:503)
        big_points_ws = st.scaling().max(dim=1).values > 0.1 * extent
        prune_mask = torch.logical_or(torch.logical_or(prune_mask, big_points_vs), big_points_ws)
    n_pruned = int(prune_mask.sum())
    st.prune_points(prune_mask)
    return st.p, st.m, st.accum, st.denom, st.max_radii2D, {"cloned": n_clone, "split": n_split, "pruned": n_pruned}
