"""Helpers for the -m gpu parity tests: run the HIP path through the C ABI and read its
internal state back with the gsr_debug_read_* entry points."""
import ctypes as C
import math

import numpy as np
import torch

from mvs_gaussian_splatting_amd import _lib
from mvs_gaussian_splatting_amd.rasterizer import GaussianRasterizationSettings, _make_params, _ptr


def product_settings(cam, bg, sh_degree, dev, scale_modifier=1.0, debug=False):
    return GaussianRasterizationSettings(
        int(cam.image_height), int(cam.image_width), math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5),
        bg.to(dev), scale_modifier, cam.world_view_transform.to(dev), cam.full_proj_transform.to(dev), sh_degree,
        cam.camera_center.to(dev), False, debug)


def forward_with_state(dev, settings, means3D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                       cov3D_precomp=None, binning_mode=None, want_stats=False, sync_free_capacity=None,
                       depth_span_lt24=False):
    """Forward through the C ABI keeping the workspaces; returns a dict of numpy/torch results.
    sync_free_capacity: run gsr_forward (no count read-back) with that instance capacity instead of the two calls;
    depth_span_lt24: ... and without the depth sort's fourth pass (GsrParams.depth_span_lt24; "depth_keys" in the result
    is what the caller has to check)."""
    lib = _lib.load()
    e = torch.empty(0, dtype=torch.float32, device=dev)
    f = lambda t: e if t is None else t.to(dev).float().contiguous()  # noqa: E731
    means3D, opacities = f(means3D), f(opacities)
    shs, colors_precomp, scales, rotations, cov3D_precomp = map(f, (shs, colors_precomp, scales, rotations, cov3D_precomp))
    P = means3D.shape[0]
    H, W = settings.image_height, settings.image_width
    with torch.cuda.device(dev):
        params, keep = _make_params(dev, settings, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp)
        stream = torch.cuda.current_stream(dev).cuda_stream
        geom = torch.empty(lib.gsr_geom_bytes(P), dtype=torch.uint8, device=dev)
        img = torch.empty(lib.gsr_image_bytes(W, H), dtype=torch.uint8, device=dev)
        radii = torch.zeros(P, dtype=torch.int32, device=dev)
        color = torch.empty(3, H, W, dtype=torch.float32, device=dev)
        R, V = C.c_uint32(0), C.c_uint32(0)
        depth_keys = None
        # the list-level parity tests speak about the un-culled lists unless a mode is asked for
        params.binning_mode = _lib.BINNING_TWO_LEVEL if binning_mode is None else binning_mode
        if sync_free_capacity is None:
            _lib.check(lib.gsr_forward_preprocess(C.byref(params), geom.data_ptr(), radii.data_ptr(), stream, C.byref(R),
                                                  C.byref(V)), "pre")
            R, V = int(R.value), int(V.value)
            nb = lib.gsr_binning_bytes(R, V, W, H, params.binning_mode)
            binning = torch.empty(nb, dtype=torch.uint8, device=dev)
            _lib.check(lib.gsr_forward_render(C.byref(params), geom.data_ptr(), binning.data_ptr(), nb, img.data_ptr(), R, V,
                                              color.data_ptr(), stream), "render")
            lay_R, lay_V = R, V
        else:
            cap = int(sync_free_capacity)
            pinned = torch.zeros(16, dtype=torch.int32).pin_memory()
            params.counts_pinned = pinned.data_ptr()
            params.depth_span_lt24 = 1 if depth_span_lt24 else 0
            nb = lib.gsr_binning_bytes(cap, P, W, H, params.binning_mode)
            binning = torch.empty(nb, dtype=torch.uint8, device=dev)
            _lib.check(lib.gsr_forward(C.byref(params), geom.data_ptr(), binning.data_ptr(), nb, cap, img.data_ptr(),
                                       radii.data_ptr(), color.data_ptr(), None, stream), "gsr_forward")
            torch.cuda.synchronize(dev)
            R, V = int(pinned[0]) & 0xffffffff, int(pinned[1]) & 0xffffffff
            depth_keys = (int(pinned[2]) & 0xffffffff, int(pinned[3]) & 0xffffffff)
            assert R <= cap, "test asked for a capacity below the instance count"
            lay_R, lay_V = cap, P       # what the workspaces are laid out for
        xy = torch.empty(P, 2, device=dev)
        con = torch.empty(P, 4, device=dev)
        rgb = torch.empty(P, 3, device=dev)
        depth = torch.empty(P, device=dev)
        tiles = torch.empty(P, dtype=torch.int32, device=dev)
        offs = torch.empty(P, dtype=torch.int32, device=dev)
        rect = torch.empty(P, 4, dtype=torch.int32, device=dev)
        clamped = torch.empty(P, dtype=torch.int32, device=dev)
        _lib.check(lib.gsr_debug_read_geom(geom.data_ptr(), P, xy.data_ptr(), con.data_ptr(), rgb.data_ptr(),
                                           depth.data_ptr(), tiles.data_ptr(), offs.data_ptr(), rect.data_ptr(),
                                           clamped.data_ptr(), stream), "read_geom")
        keys = torch.empty(max(R, 1), dtype=torch.int64, device=dev)
        plist = torch.empty(max(R, 1), dtype=torch.int32, device=dev)
        if lay_R != R:       # capacity-sized layout: the debug reader copies R_layout entries; read into capacity-sized buffers
            keys_l = torch.empty(max(lay_R, 1), dtype=torch.int64, device=dev)
            plist_l = torch.empty(max(lay_R, 1), dtype=torch.int32, device=dev)
        else:
            keys_l, plist_l = keys, plist
        _lib.check(lib.gsr_debug_read_binning(geom.data_ptr(), P, binning.data_ptr(), lay_R, lay_V, W, H, params.binning_mode,
                                              keys_l.data_ptr(), plist_l.data_ptr(), stream), "read_bin")
        keys, plist = keys_l, plist_l
        gx, gy = (W + 15) // 16, (H + 15) // 16
        final_T = torch.empty(H, W, device=dev)
        n_contrib = torch.empty(H, W, dtype=torch.int32, device=dev)
        ranges = torch.empty(gx * gy, 2, dtype=torch.int32, device=dev)
        _lib.check(lib.gsr_debug_read_image(img.data_ptr(), W, H, final_T.data_ptr(), n_contrib.data_ptr(),
                                            ranges.data_ptr(), stream), "read_img")
        cnt = (C.c_uint32 * 8)()
        _lib.check(lib.gsr_debug_read_counts(geom.data_ptr(), P, cnt, stream), "read_counts")
        stats = None
        if want_stats and R > 0:      # work counters of the forward compositing kernel (re-runs it with counting on)
            st8 = torch.zeros(8, dtype=torch.int64, device=dev)
            scratch = torch.empty_like(color)
            _lib.check(lib.gsr_debug_render_stats(C.byref(params), geom.data_ptr(), binning.data_ptr(), img.data_ptr(), lay_R, lay_V,
                                                  scratch.data_ptr(), st8.data_ptr(), stream), "stats")
            torch.cuda.synchronize(dev)
            v = st8.cpu().tolist()
            stats = {"instances": v[0], "staged": v[1], "pairs": v[2], "wave_evals": v[3]}
            assert torch.equal(scratch, color)
        torch.cuda.synchronize(dev)
    del keep
    return {"counts": [int(v) for v in cnt], "stats": stats, "depth_keys": depth_keys, "color": color.cpu(), "radii": radii.cpu(), "R": R, "V": V, "xy": xy.cpu(), "conic_opacity": con.cpu(),
            "rgb": rgb.cpu(), "depth": depth.cpu(), "tiles": tiles.cpu().numpy().astype(np.int64),
            "offsets": offs.cpu().numpy().view(np.uint32), "rect": rect.cpu().numpy().astype(np.int64),
            "clamped": clamped.cpu().numpy(), "keys": keys[:R].cpu().numpy().view(np.uint64),
            "point_list": plist[:R].cpu().numpy().view(np.uint32), "final_T": final_T.cpu(),
            "n_contrib": n_contrib.cpu(), "ranges": ranges.cpu().numpy().astype(np.int64)}


def oracle_inputs(model):
    """(means3D, opacity, shs, scales, rotations) exactly as render() hands them to the operator."""
    return model.get_xyz, model.get_opacity, model.get_features, model.get_scaling, model.get_rotation


def grads_product(dev, model, settings, target, weight, use_cov=False, use_colors=None, loss_kind="l1"):
    """HIP operator forward + backward of the masked L1 loss of tests/grad_util.py (same leaves, same weights);
    loss_kind="linear": of grad_util.weighted_sum (the all-pixel run)."""
    from grad_util import loss_of
    from mvs_gaussian_splatting_amd import GaussianRasterizer
    leaves = {}

    def leaf(name, t):
        leaves[name] = t.detach().to(dev).requires_grad_(True)
        return leaves[name]

    xyz = leaf("xyz", model._xyz)
    op = leaf("opacity", model._opacity)
    m2 = torch.zeros(xyz.shape[0], 3, device=dev, requires_grad=True)
    leaves["means2D"] = m2
    kw = {}
    if use_colors is not None:
        kw["colors_precomp"] = leaf("colors", use_colors)
    else:
        fdc, fr = leaf("f_dc", model._features_dc), leaf("f_rest", model._features_rest)
        kw["shs"] = torch.cat((fdc, fr), dim=1)
    if use_cov:
        kw["cov3D_precomp"] = leaf("cov3D", model.get_covariance(1.0))
    else:
        kw["scales"] = torch.exp(leaf("scaling", model._scaling))
        kw["rotations"] = torch.nn.functional.normalize(leaf("rotation", model._rotation))
    col, radii = GaussianRasterizer(settings)(means3D=xyz, means2D=m2, opacities=torch.sigmoid(op), **kw)
    loss_of(col, target, weight, loss_kind).backward()
    return {k: v.grad.detach().cpu() for k, v in leaves.items()}, col.detach().cpu()
