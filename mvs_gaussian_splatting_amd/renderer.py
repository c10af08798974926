"""Render host: the plain path of the reference's ``gaussian_renderer.render``
(``gaussian_renderer/__init__.py:19-90,256-313``) on duck-typed camera / model objects.

The fork's default-off grow / learned-split branch (``:91-253``) is out of scope.
"""
from __future__ import annotations

import math

import torch

from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians_fused
from .sh import eval_sh


def _can_fuse(pc, pipe, override_color) -> bool:
    """The raw-parameter fast path applies when the model is the reference's parameterisation
    (``scene/gaussian_model.py:27-42``: exp / sigmoid / normalize activations, degree-3 SH storage split
    into ``_features_dc`` / ``_features_rest``) and no Python-side fallback was requested."""
    if override_color is not None or getattr(pipe, "convert_SHs_python", False) or \
            getattr(pipe, "compute_cov3D_python", False) or not getattr(pipe, "fuse_activations", True):
        return False
    need = ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity")
    if not all(hasattr(pc, k) for k in need):
        return False
    if getattr(pc, "scaling_activation", None) is not torch.exp or \
            getattr(pc, "opacity_activation", None) is not torch.sigmoid or \
            getattr(pc, "rotation_activation", None) is not torch.nn.functional.normalize:
        return False
    fr = pc._features_rest
    # degree-3 storage (15 rest coefficients: split SH rows) or degree-0 storage (none: f_dc IS the SH tensor)
    return fr.dim() == 3 and fr.shape[1] in (0, 15) and pc._features_dc.shape[1] == 1 and pc._xyz.is_cuda


_ZEROS = {}


def _zero_leaf(like: torch.Tensor) -> torch.Tensor:
    """A leaf tensor of zeros shaped like ``like`` with ``requires_grad`` (its ``.grad`` is a fresh tensor per
    backward).  The zeros are cached per (shape, dtype, device) and shared read-only between the leaves."""
    key = (tuple(like.shape), like.dtype, like.device)
    z = _ZEROS.get(key)
    if z is None:
        if len(_ZEROS) > 8:
            _ZEROS.clear()
        z = _ZEROS[key] = torch.zeros(like.shape, dtype=like.dtype, device=like.device)
    return z.detach().requires_grad_(True)


def _fused_stats(pc, pipe, xyz):
    """(xyz_gradient_accum, denom, max_radii2D) when the caller asked for the densification statistics to be taken
    inside the backward (``pipe.fuse_densify_stats``) and the model carries float32 accumulators."""
    if not getattr(pipe, "fuse_densify_stats", False) or not (torch.is_grad_enabled() and xyz.requires_grad):
        return None
    trio = tuple(getattr(pc, k, None) for k in ("xyz_gradient_accum", "denom", "max_radii2D"))
    P = int(xyz.shape[0])
    if any(t is None or not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != P for t in trio):
        return None
    return trio


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier: float = 1.0,
           override_color=None, **_fork_kwargs):
    """Render the scene; ``bg_color`` must be on the GPU.  Returns the reference's result dict
    (``gaussian_renderer/__init__.py:309-313``).

    ``pipe.fuse_densify_stats = True`` (this build's extension) makes the backward of this frame also run
    ``add_densification_stats`` (``scene/gaussian_model.py:775-777``) and the ``max_radii2D`` update of ``train.py:130``
    on the model's accumulators; the ``add_densification_stats`` of this package then recognises the frame and does
    nothing, so the reference's call sequence (render, backward, add_densification_stats) stays as it is."""
    xyz = pc.get_xyz
    # zero tensor whose .grad receives dL/d(mean2D) for the densification statistics.  The reference builds it as
    # `zeros_like(...) + 0` + retain_grad() (gaussian_renderer/__init__.py:32-36); a leaf with requires_grad gets
    # its .grad populated all the same and saves a 72 MB copy kernel per frame at 6 M Gaussians.
    # The operator never reads (or writes) its values, so every frame's leaf aliases one cached block of zeros: a
    # fresh 72 MB memset per frame is 20 us of the 6 M-Gaussian forward.
    screenspace_points = _zero_leaf(xyz)
    stats = _fused_stats(pc, pipe, xyz)
    if stats is not None:
        screenspace_points._gsr_stats_fused = True      # read by losses.add_densification_stats

    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height),
        image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5),
        tanfovy=math.tan(viewpoint_camera.FoVy * 0.5),
        bg=bg_color,
        scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform,
        sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center,
        prefiltered=False,
        debug=bool(getattr(pipe, "debug", False)),
    )
    if _can_fuse(pc, pipe, override_color):
        # same result as the getter path below, without materialising cat(f_dc, f_rest), exp, normalize, sigmoid (and
        # without building an nn.Module per frame: the operator is called as a function)
        # visibility_filter (= radii > 0, gaussian_renderer/__init__.py:311) is stored by the preprocess kernel itself:
        # a torch compare over 6 M radii is a 9-us kernel per frame
        visible = torch.empty(xyz.shape[0], dtype=torch.bool, device=xyz.device)
        rendered_image, radii = rasterize_gaussians_fused(xyz, screenspace_points, pc._features_dc, pc._features_rest,
                                                          pc._opacity, pc._scaling, pc._rotation, raster_settings,
                                                          densify_stats=stats, visible=visible)
        return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": visible,
                "radii": radii, "selected_pts_mask": None}

    rasterizer = GaussianRasterizer(raster_settings=raster_settings)
    scales = rotations = cov3D_precomp = None
    if getattr(pipe, "compute_cov3D_python", False):
        cov3D_precomp = pc.get_covariance(scaling_modifier)
    else:
        scales, rotations = pc.get_scaling, pc.get_rotation

    shs = colors_precomp = None
    if override_color is None:
        if getattr(pipe, "convert_SHs_python", False):
            feats = pc.get_features
            shs_view = feats.transpose(1, 2).reshape(-1, 3, (pc.max_sh_degree + 1) ** 2)
            dir_pp = xyz - viewpoint_camera.camera_center.repeat(feats.shape[0], 1)
            dir_pp = dir_pp / dir_pp.norm(dim=1, keepdim=True)
            colors_precomp = torch.clamp_min(eval_sh(pc.active_sh_degree, shs_view, dir_pp) + 0.5, 0.0)
        else:
            shs = pc.get_features
    else:
        colors_precomp = override_color

    # exactly the reference's eight keyword arguments (gaussian_renderer/__init__.py:257-265); the statistics request of
    # this build travels as a ninth only when the caller asked for it
    extra = {} if stats is None else {"densify_stats": stats}
    rendered_image, radii = rasterizer(means3D=xyz, means2D=screenspace_points, shs=shs,
                                       colors_precomp=colors_precomp, opacities=pc.get_opacity, scales=scales,
                                       rotations=rotations, cov3D_precomp=cov3D_precomp, **extra)
    return {"render": rendered_image,
            "viewspace_points": screenspace_points,
            "visibility_filter": radii > 0,
            "radii": radii,
            "selected_pts_mask": None}
