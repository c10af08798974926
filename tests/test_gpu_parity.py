"""HIP path vs CPU oracle on identical seeded inputs (run on the GPU box: pytest -m gpu).

Tolerances (BASELINE.json north_star: 1e-5 relative fp32):
* per-Gaussian preprocess results: the kernel mirrors the oracle's float32 op order with fma contraction
  off, so integers (radii, tile rects, offsets, sort keys, point lists, ranges) are compared EXACTLY and
  floats to 1e-6 relative;
* pixels: 1e-5 * max(1, |ref|) on every pixel whose compositing decisions are not within 1e-4 (relative) of
  a threshold in the oracle (alpha vs 1/255, T vs 1e-4).  A pixel that sits on a threshold may legitimately
  flip between two float32 exp implementations; those are bounded to a small fraction and to the size of one
  flipped contribution;
* gradients: against float64 autograd of the oracle at 1e-5, max-norm relative per parameter tensor
  (max|got - ref| / max|ref|), UNCONDITIONALLY: the threshold-fragile pixels get loss weight 0 on both sides
  (tests/grad_util.py) instead of loosening the bar; the only escape is conditioning -- 2 x the error of the
  oracle itself run in float32 where that independent float32 implementation misses 1e-5.  Every test prints
  the per-tensor errors that ran.
"""
import math

import numpy as np
import pytest
import torch

from conftest import make_settings, small_scene

pytestmark = pytest.mark.gpu


def _oracle_forward(model, cam, bg, deg, dtype=torch.float32, **kw):
    from oracle import rasterize_ref
    c = lambda t: t.to(dtype)  # noqa: E731
    st = make_settings(cam, bg, deg)
    return rasterize_ref(c(model.get_xyz), None, c(model.get_opacity), st, shs=c(model.get_features),
                         scales=c(model.get_scaling), rotations=c(model.get_rotation), want_aux=True, want_margin=True, **kw)


@pytest.mark.parametrize("mode", [0, 1], ids=["two_level", "keys64"])
@pytest.mark.parametrize("deg,P,w,h", [(3, 3000, 320, 176), (0, 2000, 200, 200), (1, 1500, 97, 131), (2, 800, 64, 48)])
def test_forward_stages_match_oracle(gpu_device, deg, P, w, h, mode):
    from gpu_util import forward_with_state, product_settings
    model, cam, bg, _ = small_scene(P=P, sh_degree=deg, width=w, height=h)
    bg = torch.tensor([0.1, 0.2, 0.3])
    col, radii, aux = _oracle_forward(model, cam, bg, deg)
    st = product_settings(cam, bg, deg, gpu_device)
    out = forward_with_state(gpu_device, st, model.get_xyz, model.get_opacity, shs=model.get_features,
                             scales=model.get_scaling, rotations=model.get_rotation, binning_mode=mode)
    pre = aux["pre"]
    assert out["V"] == int((radii > 0).sum())
    # ---- preprocess: integers exact ------------------------------------------------------
    assert torch.equal(out["radii"], radii)
    assert np.array_equal(out["tiles"], pre["tiles_touched"].numpy())
    assert np.array_equal(out["offsets"].astype(np.int64), np.cumsum(pre["tiles_touched"].numpy()))
    vis = (radii > 0).numpy()
    keep = pre["keep"].numpy()
    gid = pre["idx"].numpy()[keep]
    assert np.array_equal(np.nonzero(vis)[0], gid)
    assert np.array_equal(out["rect"][gid], pre["v_rect"].numpy()[keep])
    # ---- preprocess: floats --------------------------------------------------------------
    def close(a, b, tol=1e-6):
        a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
        return np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)), initial=0.0) <= tol
    assert close(out["xy"].numpy()[gid], pre["v_xy"].numpy()[keep])
    assert close(out["depth"].numpy()[gid], pre["v_depth"].numpy()[keep])
    assert close(out["conic_opacity"].numpy()[gid, :3], pre["v_conic"].numpy()[keep])
    assert close(out["conic_opacity"].numpy()[gid, 3], pre["v_opacity"].numpy()[keep])
    assert close(out["rgb"].numpy()[gid], pre["v_rgb"].numpy()[keep])
    cl = pre["v_clamped"].numpy()[keep]
    assert np.array_equal(out["clamped"][gid], cl[:, 0] * 1 + cl[:, 1] * 2 + cl[:, 2] * 4)
    # ---- binning: exact ------------------------------------------------------------------
    assert out["R"] == aux["keys"].shape[0]
    assert np.array_equal(out["keys"], aux["keys"])
    assert np.array_equal(out["point_list"], aux["point_list"])
    assert np.array_equal(out["ranges"], aux["ranges"])
    # ---- pixels ----------------------------------------------------------------------------
    margin = aux["margin"]
    robust = margin > 1e-4
    err = (out["color"] - col).abs() / col.abs().clamp(min=1.0)
    err_px = err.max(dim=0).values
    assert float(err_px[robust].max()) <= 1e-5, f"robust pixels differ by {float(err_px[robust].max())}"
    fragile = ~robust
    assert int(fragile.sum()) <= 0.01 * robust.numel()
    assert float(err_px.max()) <= 2.0 / 255.0          # a flipped pixel moves by at most ~one min-alpha contribution
    assert torch.equal(out["n_contrib"][robust], aux["n_contrib"][robust])
    tdiff = (out["final_T"] - aux["final_T"]).abs()
    assert float(tdiff[robust].max()) <= 1e-5


def test_colors_precomp_and_cov3d_precomp(gpu_device):
    from gpu_util import forward_with_state, product_settings
    from oracle import rasterize_ref
    model, cam, bg, _ = small_scene(P=1500, sh_degree=0, width=160, height=96)
    g = torch.Generator().manual_seed(5)
    colors = torch.rand(1500, 3, generator=g)
    cov = model.get_covariance(1.0)
    st = make_settings(cam, bg, 0)
    col, radii, aux = rasterize_ref(model.get_xyz, None, model.get_opacity, st, colors_precomp=colors,
                                    cov3D_precomp=cov, want_aux=True, want_margin=True)
    out = forward_with_state(gpu_device, product_settings(cam, bg, 0, gpu_device), model.get_xyz, model.get_opacity,
                             colors_precomp=colors, cov3D_precomp=cov)
    assert torch.equal(out["radii"], radii)
    assert np.array_equal(out["point_list"], aux["point_list"])
    robust = aux["margin"] > 1e-4
    err = ((out["color"] - col).abs() / col.abs().clamp(min=1.0)).max(dim=0).values
    assert float(err[robust].max()) <= 1e-5


def test_empty_and_degenerate_inputs(gpu_device):
    """No Gaussians, all culled (behind the camera), zero-opacity: image == background."""
    from mvs_gaussian_splatting_amd import GaussianRasterizer
    from gpu_util import product_settings
    model, cam, _, _ = small_scene(P=64, sh_degree=0, width=50, height=34)
    bg = torch.tensor([0.25, 0.5, 0.75])
    st = product_settings(cam, bg, 0, gpu_device)
    dev = gpu_device
    expect = bg.view(3, 1, 1).expand(3, 34, 50)
    # P == 0
    z = torch.zeros(0, 3, device=dev)
    col, radii = GaussianRasterizer(st)(means3D=z, means2D=z, opacities=torch.zeros(0, 1, device=dev),
                                        shs=torch.zeros(0, 1, 3, device=dev), scales=z, rotations=torch.zeros(0, 4, device=dev))
    assert radii.numel() == 0 and torch.allclose(col.cpu(), expect)
    # all behind the camera
    xyz = model.get_xyz.clone()
    xyz[:, 2] = -xyz[:, 2]
    args = dict(means2D=torch.zeros_like(xyz).to(dev), opacities=model.get_opacity.to(dev), shs=model.get_features.to(dev),
                scales=model.get_scaling.to(dev), rotations=model.get_rotation.to(dev))
    col, radii = GaussianRasterizer(st)(means3D=xyz.to(dev), **args)
    assert int((radii > 0).sum()) == 0 and torch.allclose(col.cpu(), expect)
    # opacity 0 -> background
    args["opacities"] = torch.zeros_like(args["opacities"])
    col, radii = GaussianRasterizer(st)(means3D=model.get_xyz.to(dev), **args)
    assert int((radii > 0).sum()) > 0 and torch.allclose(col.cpu(), expect)


def _masked_grad_parity(dev, model, cam, bg, target, deg, label, use_cov=False, use_colors=None, smod=1.0, unmasked=True):
    """float64 oracle defines the loss weights; float32 oracle bounds the conditioning; HIP must meet the bar.
    unmasked: also run a discontinuity-free loss over EVERY pixel (the threshold-fragile ones included): at the same bar
    when the scene has no fragile pixel, at the per-scene fragile-share bar when it has (tests/grad_util.py)."""
    from gpu_util import grads_product, product_settings
    from grad_util import grads_oracle, compare_grads, compare_grads_unmasked
    st_o = make_settings(cam, bg, deg, scale_modifier=smod)
    ref, weight, aux, col64 = grads_oracle(model, st_o, target, use_cov=use_cov, use_colors=use_colors)
    ref32, _, _, _ = grads_oracle(model, st_o, target, dtype=torch.float32, use_cov=use_cov, use_colors=use_colors,
                                  weight=weight)
    got, col = grads_product(dev, model, product_settings(cam, bg, deg, dev, scale_modifier=smod), target, weight,
                             use_cov, use_colors)
    n_fragile = int((aux["margin"] <= 1e-4).sum())
    n_masked = int((weight == 0).sum())
    assert n_masked <= 0.3 * weight.numel(), "the mask must leave most of the image in the loss"
    compare_grads(got, ref, ref32, f"{label} (fragile pixels {n_fragile}, masked elements {n_masked}/{weight.numel()})")
    # the z component of the screen-space gradient is never written to
    assert float(got["means2D"][:, 2].abs().max()) == 0.0
    for k, r in ref.items():          # every compared tensor carries a real signal
        if k != "means2D" and r.numel():
            assert float(r.abs().max()) > 0.0, k
    assert float(ref["means2D"][:, :2].abs().max()) > 0.0
    if unmasked:
        # EVERY pixel in the loss, through a loss that has no discontinuity of its own (grad_util.weighted_sum)
        from grad_util import linear_weights
        wts = linear_weights(weight.shape)
        kw_o = dict(use_cov=use_cov, use_colors=use_colors, loss_kind="linear")
        ref_u, _, _, _ = grads_oracle(model, st_o, target, weight=wts, **kw_o)
        got_u, _ = grads_product(dev, model, product_settings(cam, bg, deg, dev, scale_modifier=smod), target, wts,
                                 use_cov, use_colors, loss_kind="linear")
        if n_fragile == 0:
            # no pixel on which float32 and float64 may decide differently: the bar of the masked run, on every pixel
            ref_u32, _, _, _ = grads_oracle(model, st_o, target, dtype=torch.float32, weight=wts, **kw_o)
            compare_grads(got_u, ref_u, ref_u32, f"{label}, every pixel in the loss (none threshold-fragile)")
        else:
            # what the same loss over the ROBUST pixels gives in float64 (and float32: the conditioning of that part): the
            # difference to ref_u is the share of the gradient the fragile pixels carry -- it sets the bar of this scene
            robust_w = wts * (aux["margin"] > 1e-4)[None].to(wts.dtype)
            ref_r, _, _, _ = grads_oracle(model, st_o, target, weight=robust_w, **kw_o)
            ref_r32, _, _, _ = grads_oracle(model, st_o, target, dtype=torch.float32, weight=robust_w, **kw_o)
            compare_grads_unmasked(got_u, ref_u, n_fragile, label, ref_masked=ref_r, ref_masked32=ref_r32)
            # ... and with ONLY those few pixels out of the loss (no sign mask, no tile mask) the bar is the strict one,
            # with 3 x (not 2 x) the float32 oracle's error as the conditioning escape: random-sign weights make the
            # per-Gaussian sums cancel (the float32 oracle itself is 10-40 x further from float64 under this loss than
            # under the masked L1 of the same scene), and the extreme errors of two float32 implementations of such a sum
            # differ by a small factor either way (rare-branches scene: HIP 2.7 x the oracle's on means2D, 0.1 x on
            # scaling).  The cap (2e-4) is unchanged.
            got_r, _ = grads_product(dev, model, product_settings(cam, bg, deg, dev, scale_modifier=smod), target, robust_w,
                                     use_cov, use_colors, loss_kind="linear")
            compare_grads(got_r, ref_r, ref_r32, f"{label}, every pixel but the {n_fragile} threshold-fragile ones in the loss",
                          e32_factor=3.0)
    return got, ref, weight, aux


@pytest.mark.parametrize("deg,use_cov,colors", [(3, False, False), (1, False, False), (0, True, True)])
def test_backward_matches_fp64_oracle(gpu_device, deg, use_cov, colors):
    model, cam, _, target = small_scene(P=2500, sh_degree=deg, width=208, height=120, scale=0.06)
    bg = torch.tensor([0.3, 0.1, 0.2])
    use_colors = torch.rand(2500, 3, generator=torch.Generator().manual_seed(3)) if colors else None
    _masked_grad_parity(gpu_device, model, cam, bg, target, deg, f"backward deg={deg} cov={use_cov} colors={colors}",
                        use_cov, use_colors)


@pytest.mark.parametrize("deg", [3, 0])
def test_fused_raw_parameter_path_matches_unfused_and_oracle(gpu_device, deg):
    """render() fed with raw parameters (split SH for degree-3 storage, f_dc alone for degree-0 storage;
    activations inside the kernels) must give the pixels and raw-parameter gradients of the getter path / the fp64 oracle."""
    from mvs_gaussian_splatting_amd import render
    from mvs_gaussian_splatting_amd.synthetic import PipelineParams
    from grad_util import grads_oracle, compare_grads, masked_l1
    dev = gpu_device
    from mvs_gaussian_splatting_amd.renderer import _can_fuse
    model, cam, _, target = small_scene(P=2500, sh_degree=deg, width=208, height=120, scale=0.06)
    bg = torch.tensor([0.3, 0.1, 0.2])
    st_o = make_settings(cam, bg, deg)
    ref, weight, aux, _ = grads_oracle(model, st_o, target)
    ref32, _, _, _ = grads_oracle(model, st_o, target, dtype=torch.float32, weight=weight)
    model.to(dev); cam.to(dev)
    out = {}
    for fused in (True, False):
        for p in model.parameters():
            p.grad = None
            p.requires_grad_(True)
        pipe = PipelineParams()
        pipe.fuse_activations = fused
        assert _can_fuse(model, pipe, None) == fused
        pkg = render(cam, model, pipe, bg.to(dev))
        masked_l1(pkg["render"], target, weight).backward()
        out[fused] = (pkg["render"].detach().cpu(), pkg["radii"].cpu(),
                      {"xyz": model._xyz.grad.cpu(), "f_dc": model._features_dc.grad.cpu(),
                       "f_rest": (model._features_rest.grad.cpu() if model._features_rest.grad is not None
                                  else torch.zeros_like(model._features_rest).cpu()),
                       "opacity": model._opacity.grad.cpu(),
                       "scaling": model._scaling.grad.cpu(), "rotation": model._rotation.grad.cpu(),
                       "means2D": pkg["viewspace_points"].grad.cpu()})
    assert int((out[True][1] != out[False][1]).sum()) <= 2          # expf vs torch.exp may flip a ceil()
    assert float((out[True][0] - out[False][0]).abs().max()) <= 2.0 / 255.0
    n_fragile = int((aux["margin"] <= 1e-4).sum())
    for fused in (True, False):
        compare_grads(out[fused][2], ref, ref32, f"render() fused={fused} (fragile pixels {n_fragile})")


@pytest.mark.parametrize("n,end_bit", [(1, 45), (63, 45), (4097, 45), (1_000_003, 45), (300_000, 64), (50_000, 17)])
def test_sort_pairs_u64_matches_numpy_stable_argsort(gpu_device, n, end_bit):
    """stable LSD radix sort of (u64 key, u32 value) on bits [0, end_bit): bit-exact against
    numpy.argsort(kind='stable') of the masked keys (values carry the original index, so stability is checked)."""
    import ctypes as C
    from mvs_gaussian_splatting_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(n)
    keys = rng.integers(0, 2 ** 63, size=n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=n, dtype=np.uint64)
    if n > 1000:
        keys[: n // 4] = keys[0]                      # long runs of duplicates
    mask = np.uint64((1 << end_bit) - 1) if end_bit < 64 else np.uint64(2 ** 64 - 1)
    order = np.argsort(keys & mask, kind="stable")
    dev = gpu_device
    k = torch.from_numpy(keys.view(np.int64)).to(dev)
    v = torch.arange(n, dtype=torch.int32, device=dev)
    kt, vt = torch.empty_like(k), torch.empty_like(v)
    scratch = torch.empty(lib.gsr_sort_scratch_bytes(n), dtype=torch.uint8, device=dev)
    in_tmp = C.c_int32(0)
    with torch.cuda.device(dev):
        _lib.check(lib.gsr_sort_pairs_u64(k.data_ptr(), v.data_ptr(), kt.data_ptr(), vt.data_ptr(), n, end_bit,
                                          scratch.data_ptr(), torch.cuda.current_stream(dev).cuda_stream,
                                          C.byref(in_tmp)), "sort")
        torch.cuda.synchronize()
    ko, vo = (kt, vt) if in_tmp.value else (k, v)
    assert np.array_equal(vo.cpu().numpy().astype(np.int64), order)
    assert np.array_equal(ko.cpu().numpy().view(np.uint64), keys[order])


# ---------------------------------------------------------------------------------------------------
# Full-size checks (BASELINE config C4: 6 M Gaussians, SH 3, 1920x1080) through size-independent properties
# ---------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def c4_scene(gpu_device):
    from mvs_gaussian_splatting_amd.synthetic import CONFIGS, make_scene
    model, cam, bg, target = make_scene(CONFIGS["C4"])
    return model, cam, bg, target


def test_full_size_binning_properties_and_mode_equivalence(gpu_device, c4_scene):
    """Sortedness, range partition, count identities; two-level and 64-bit-key binning give identical lists."""
    from gpu_util import forward_with_state, product_settings
    model, cam, bg, _ = c4_scene
    st = product_settings(cam, bg, 3, gpu_device)
    outs = []
    for mode in (0, 1):
        with torch.no_grad():
            o = forward_with_state(gpu_device, st, model.get_xyz, model.get_opacity, shs=model.get_features,
                                   scales=model.get_scaling, rotations=model.get_rotation, binning_mode=mode)
        keys, plist, ranges, tiles = o["keys"], o["point_list"], o["ranges"], o["tiles"]
        assert o["R"] == int(tiles.sum()) == keys.size and o["V"] == int((o["radii"] > 0).sum())
        assert np.all(keys[1:] >= keys[:-1])                                   # sorted by (tile, depth)
        same = keys[1:] == keys[:-1]
        assert np.all(plist[1:][same] > plist[:-1][same])                      # ties in Gaussian-index order
        ne = ranges[:, 1] > ranges[:, 0]
        assert int((ranges[ne, 1] - ranges[ne, 0]).sum()) == o["R"]            # ranges partition the list
        starts = np.sort(ranges[ne, 0])
        assert starts[0] == 0 and np.all(np.diff(starts) > 0)
        tile_of = (keys >> np.uint64(32)).astype(np.int64)
        assert np.array_equal(np.bincount(tile_of, minlength=ranges.shape[0]), ranges[:, 1] - ranges[:, 0])
        assert np.array_equal(np.sort(np.bincount(plist, minlength=tiles.size)), np.sort(tiles))  # one instance per overlapped tile
        assert np.array_equal(np.bincount(plist, minlength=tiles.size), tiles)
        outs.append(o)
    assert np.array_equal(outs[0]["keys"], outs[1]["keys"])
    assert np.array_equal(outs[0]["point_list"], outs[1]["point_list"])
    assert np.array_equal(outs[0]["ranges"], outs[1]["ranges"])
    assert torch.equal(outs[0]["color"], outs[1]["color"])                    # same lists -> bitwise same pixels


def test_full_size_compositing_invariants(gpu_device, c4_scene):
    """colour == 1, bg == 1 -> every pixel is sum(w) + T_final == 1; colour scaling by 2 is exact; bg = 0 image of
    a zero-opacity cloud is 0."""
    from mvs_gaussian_splatting_amd import GaussianRasterizer
    from gpu_util import product_settings
    model, cam, _, _ = c4_scene
    dev = gpu_device
    P = model.get_xyz.shape[0]
    with torch.no_grad():
        args = dict(means3D=model.get_xyz.to(dev), means2D=torch.zeros(P, 3, device=dev),
                    scales=model.get_scaling.to(dev), rotations=model.get_rotation.to(dev))
        op = model.get_opacity.to(dev)
        ones = torch.ones(P, 3, device=dev)
        st1 = product_settings(cam, torch.ones(3), 0, dev)
        col, radii = GaussianRasterizer(st1)(opacities=op, colors_precomp=ones, **args)
        assert float((col - 1.0).abs().max()) < 1.2e-4                         # <= T_STOP of mass lost at saturation
        st0 = product_settings(cam, torch.zeros(3), 0, dev)
        c = torch.rand(P, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
        a, _ = GaussianRasterizer(st0)(opacities=op, colors_precomp=c, **args)
        b, _ = GaussianRasterizer(st0)(opacities=op, colors_precomp=2.0 * c, **args)
        assert torch.equal(b, 2.0 * a)                                         # linear in colour, power of two exact
        z, _ = GaussianRasterizer(st0)(opacities=torch.zeros_like(op), colors_precomp=c, **args)
        assert float(z.abs().max()) == 0.0


def test_full_size_train_step_is_deterministic_and_fused_matches_getters(gpu_device, c4_scene):
    """The backward has no atomics: two runs are bitwise identical.  The raw-parameter (fused) path agrees with
    the getter path up to expf/sigmoid rounding."""
    from mvs_gaussian_splatting_amd import render, l1_loss
    from mvs_gaussian_splatting_amd.synthetic import PipelineParams, CONFIGS, make_scene
    dev = gpu_device
    model, cam, bg, target = make_scene(CONFIGS["C4"])
    model.to(dev); cam.to(dev)
    bg, target = bg.to(dev), target.to(dev)
    for p in model.parameters():
        p.requires_grad_(True)

    def step(fused):
        for p in model.parameters():
            p.grad = None
        pipe = PipelineParams()
        pipe.fuse_activations = fused
        pkg = render(cam, model, pipe, bg)
        loss = l1_loss(pkg["render"], target)
        loss.backward()
        return pkg["render"].detach().clone(), [p.grad.detach().clone() for p in model.parameters()], \
            pkg["viewspace_points"].grad.detach().clone()

    img1, g1, m1 = step(True)
    img2, g2, m2 = step(True)
    assert torch.equal(img1, img2) and torch.equal(m1, m2)
    assert all(torch.equal(a, b) for a, b in zip(g1, g2))
    img3, g3, m3 = step(False)
    assert float((img1 - img3).abs().max()) <= 2.0 / 255.0
    assert float((img1 - img3).abs().mean()) <= 1e-6
    for a, b in zip(g1, g3):
        assert float((a - b).abs().max()) <= 1e-3 * float(b.abs().max()) + 1e-12
    assert all(torch.isfinite(a).all() for a in g1)


def test_l1_and_dssim_loss_match_reference_golden(gpu_device):
    """fused L1 and L1 + D-SSIM kernels against the fixtures generated from the reference's
    utils/loss_utils.py (value and gradient), and against torch on a 1080p-shaped input."""
    import os
    from mvs_gaussian_splatting_amd import l1_loss, l1_dssim_loss
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loss.npz"))
    dev = gpu_device
    a = torch.tensor(g["a"], device=dev, requires_grad=True)
    b = torch.tensor(g["b"], device=dev)
    l = l1_loss(a, b)
    l.backward()
    assert math.isclose(l.item(), float(g["l1"]), rel_tol=1e-6)
    assert torch.allclose(a.grad.cpu(), torch.tensor(g["l1_grad"]))
    lam = 0.2
    a.grad = None
    tot = l1_dssim_loss(a, b, lam)
    tot.backward()
    want = (1 - lam) * float(g["l1"]) + lam * (1 - float(g["ssim"]))
    assert math.isclose(tot.item(), want, rel_tol=1e-5)
    want_grad = (1 - lam) * torch.tensor(g["l1_grad"]) - lam * torch.tensor(g["ssim_grad"])
    err = float((a.grad.cpu() - want_grad).abs().max()) / float(want_grad.abs().max())
    assert err <= 1e-4, err
    # odd, non-multiple-of-16 shape vs a torch restatement on the device
    x = torch.rand(3, 131, 77, device=dev, requires_grad=True)
    y = torch.rand(3, 131, 77, device=dev)
    l1_dssim_loss(x, y, lam).backward()
    gx = x.grad.clone(); x.grad = None
    import torch.nn.functional as F
    w1 = torch.tensor([math.exp(-(i - 5) ** 2 / (2 * 1.5 ** 2)) for i in range(11)], device=dev)
    w1 = (w1 / w1.sum()).unsqueeze(1)
    win = (w1 @ w1.t()).expand(3, 1, 11, 11).contiguous()
    def ssim_t(p, q):
        p, q = p[None], q[None]
        m1, m2 = F.conv2d(p, win, padding=5, groups=3), F.conv2d(q, win, padding=5, groups=3)
        s1 = F.conv2d(p * p, win, padding=5, groups=3) - m1 * m1
        s2 = F.conv2d(q * q, win, padding=5, groups=3) - m2 * m2
        s12 = F.conv2d(p * q, win, padding=5, groups=3) - m1 * m2
        return (((2 * m1 * m2 + 1e-4) * (2 * s12 + 9e-4)) / ((m1 * m1 + m2 * m2 + 1e-4) * (s1 + s2 + 9e-4))).mean()
    ref = (1 - lam) * (x - y).abs().mean() + lam * (1 - ssim_t(x, y))
    ref.backward()
    assert float((gx - x.grad).abs().max()) <= 1e-4 * float(x.grad.abs().max())


def test_mark_visible_matches_preprocess_near_plane(gpu_device):
    from mvs_gaussian_splatting_amd import GaussianRasterizer
    from gpu_util import product_settings
    model, cam, bg, _ = small_scene(P=4000, sh_degree=0, width=64, height=64, view=3)
    st = product_settings(cam, bg, 0, gpu_device)
    vis = GaussianRasterizer(st).markVisible(model.get_xyz.to(gpu_device)).cpu()
    z = model.get_xyz @ cam.world_view_transform[:3, 2] + cam.world_view_transform[3, 2]
    robust = (z - 0.2).abs() > 1e-5
    assert torch.equal(vis[robust], (z > 0.2)[robust]) and 0 < int(vis.sum()) < 4000


# ---------------------------------------------------------------------------------------------------
# BASELINE configs 2 and 3 at their real sizes against the oracle
# ---------------------------------------------------------------------------------------------------
def test_config_C2_full_forward_matches_oracle(gpu_device):
    """C2: 100 k Gaussians, SH degree 0, 800x800, forward only -- every integer exact, every robust pixel 1e-5."""
    from gpu_util import forward_with_state, product_settings
    from mvs_gaussian_splatting_amd.synthetic import CONFIGS, make_scene
    torch.set_num_threads(min(16, len(__import__("os").sched_getaffinity(0))))
    cfg = CONFIGS["C2"]
    model, cam, bg, _ = make_scene(cfg)
    col, radii, aux = _oracle_forward(model, cam, bg, cfg.sh_degree)
    out = forward_with_state(gpu_device, product_settings(cam, bg, cfg.sh_degree, gpu_device), model.get_xyz,
                             model.get_opacity, shs=model.get_features, scales=model.get_scaling,
                             rotations=model.get_rotation)
    assert torch.equal(out["radii"], radii)
    assert np.array_equal(out["keys"], aux["keys"]) and np.array_equal(out["point_list"], aux["point_list"])
    assert np.array_equal(out["ranges"], aux["ranges"])
    robust = aux["margin"] > 1e-4
    err = ((out["color"] - col).abs() / col.abs().clamp(min=1.0)).max(dim=0).values
    assert float(err[robust].max()) <= 1e-5
    assert int((~robust).sum()) <= 0.01 * robust.numel()
    assert torch.equal(out["n_contrib"][robust], aux["n_contrib"][robust])


def _masked_train_step_vs_fp64_oracle(dev, cfg_name, view, tile_step, label, check_stats=False, select_from=None):
    """One train step of a BASELINE config at its real size.  The L1 loss is restricted to ~40 tiles spread over the
    image (mask), so that the float64 autograd oracle only has to composite those tiles; the HIP path runs the whole frame
    and must produce the same pixels there and the same parameter gradients (tests/grad_util.py bar).
    select_from = (step_x, step_y): the 40 tiles are not the fixed lattice `tile_step` but the 40 best-conditioned tiles of
    a denser lattice, ranked by the ORACLE alone (grad_util.pick_well_conditioned_tiles: float32 against float64
    compositing gradients per tile) -- for the views whose fixed lattice meets a tile no float32 implementation
    resolves to the 2e-4 cap."""
    from mvs_gaussian_splatting_amd import render, add_densification_stats
    from mvs_gaussian_splatting_amd.synthetic import CONFIGS, make_scene, PipelineParams
    import os
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    cfg = CONFIGS[cfg_name]
    model, cam, bg, target = make_scene(cfg, view=view)
    gx, gy = (cfg.width + 15) // 16, (cfg.height + 15) // 16
    tiles = [ty * gx + tx for ty in range(3, gy, tile_step[1]) for tx in range(5, gx, tile_step[0])]
    from grad_util import (grads_oracle, compare_grads, masked_l1, full_frame_lists, members_of_tiles, SubModel,
                           expand_grads, pick_well_conditioned_tiles)
    st = make_settings(cam, bg, cfg.sh_degree)
    P = model._xyz.shape[0]
    # The masked loss only reaches the Gaussians in the lists of the masked tiles (tens of thousands of the 6 M): the
    # whole model goes through the oracle's preprocess + binning WITHOUT autograd (radii of all P, the tile lists), the
    # differentiable float64 / float32 oracle then runs on exactly those Gaussians -- same per-Gaussian arithmetic, same
    # lists for those tiles (a subset keeps the index order, so depth ties sort alike); every other Gaussian's gradient is
    # exactly zero on both sides, which is asserted of the HIP path below.
    lists = full_frame_lists(model, st)
    if select_from is not None:
        cands = [ty * gx + tx for ty in range(1, gy, select_from[1]) for tx in range(2, gx, select_from[0])]
        sub_c = SubModel(model, members_of_tiles(lists, cands))
        tiles, info = pick_well_conditioned_tiles(sub_c, st, target, cands, len(tiles))
        print(f"[tiles] {label}: {len(tiles)} of {info['n_candidates']} candidate tiles by float32-vs-float64 conditioning of "
              f"the compositing gradients: worst picked {info['scores_picked_max']:.2e}, median of all "
              f"{info['scores_all_median']:.2e}, worst of all {info['scores_all_max']:.2e}")
    mask = torch.zeros(1, cfg.height, cfg.width)
    for t in tiles:
        ty, tx = divmod(t, gx)
        mask[:, ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16] = 1.0
    # ---- oracle, float64 (defines the loss weights) and float32 (conditioning bound), only the masked tiles ---------
    idx = members_of_tiles(lists, tiles)
    sub = SubModel(model, idx)
    ref_s, weight, aux, col = grads_oracle(sub, st, target, tiles=tiles, tile_mask=mask)
    ref32_s, _, aux32, col32 = grads_oracle(sub, st, target, dtype=torch.float32, tiles=tiles, weight=weight)
    ref, ref32 = expand_grads(ref_s, idx, P), expand_grads(ref32_s, idx, P)
    in_loss = torch.zeros(P, dtype=torch.bool)
    in_loss[idx] = True
    aux["radii"], aux32["radii"] = lists[torch.float64][0], lists[torch.float32][0]
    print(f"[oracle] {label}: {idx.numel()} of {P} Gaussians reach the {len(tiles)} tiles in the loss")
    # ---- HIP path, whole frame -------------------------------------------------------------------------
    model.to(dev); cam.to(dev)
    for p in model.parameters():
        p.requires_grad_(True)
    pkg = render(cam, model, PipelineParams(), bg.to(dev))
    img = pkg["render"]
    masked_l1(img, target, weight).backward()
    m = mask[0].bool()
    robust = (aux["margin"] > 1e-4) & m
    # pixels, relative to max(1, |ref|) (colours of a degree-3 cloud exceed 1) as in the C2 test.  Two bars:
    #  (a) against the float32 oracle 1e-5 -- north_star's bar is float32 against float32 (the reference rasterizer
    #      computes in float32);
    #  (b) against float64 1e-5 wherever float32 can deliver it: a pixel with several hundred contributors accumulates
    #      more float32 rounding than that, so the bar of a pixel is 2 x the float32 oracle's own error there, never more
    #      than 1e-4 (the rule of grad_util.compare_grads, per pixel).
    rel = lambda x, r: ((x.double() - r.double()).abs() / r.double().abs().clamp(min=1.0)).max(dim=0).values  # noqa: E731
    got = img.detach().cpu()
    err, err32, err_vs32 = rel(got, col), rel(col32, col), rel(got, col32)
    worst = int(torch.where(robust, err, torch.zeros_like(err)).argmax())
    wy, wx = divmod(worst, cfg.width)
    print(f"[pixels] {label}: vs float64: worst robust pixel ({wx},{wy}) err {float(err[wy, wx]):.2e} (float32 oracle there "
          f"{float(err32[wy, wx]):.2e}, its own worst {float(err32[robust].max()):.2e}), |ref| {float(col[:, wy, wx].abs().max()):.3f}, "
          f"contributors {int(aux['n_contrib'][wy, wx])}, mean {float(err[robust].mean()):.2e}; vs the float32 oracle: worst "
          f"{float(err_vs32[robust].max()):.2e}, mean {float(err_vs32[robust].mean()):.2e}")
    assert float(err_vs32[robust].max()) <= 1e-5
    bar = (2.0 * err32).clamp(min=1e-5)
    assert float(bar[robust].max()) <= 1e-4, "float32 itself is too far from float64 on a pixel of this scene"
    assert bool((err[robust] <= bar[robust]).all())
    n_fragile = int(((aux["margin"] <= 1e-4) & m).sum())
    assert n_fragile <= 0.02 * int(m.sum())
    got = {"xyz": model._xyz.grad, "f_dc": model._features_dc.grad, "f_rest": model._features_rest.grad,
           "opacity": model._opacity.grad, "scaling": model._scaling.grad, "rotation": model._rotation.grad,
           "means2D": pkg["viewspace_points"].grad}
    got = {k: v.detach().cpu() for k, v in got.items()}
    for k, v in got.items():      # a Gaussian in none of the masked tiles' lists gets no gradient at all
        assert not v[~in_loss].any(), f"{label}: {k} has a gradient for a Gaussian that reaches no tile in the loss"
    compare_grads(got, ref, ref32, f"{label} ({len(tiles)} tiles, fragile pixels {n_fragile})")
    if check_stats:
        # densification statistics of the step (scene/gaussian_model.py:775-777, train.py:130) against the oracle's
        # ||dL/dmeans2D[:, :2]|| -- same max-norm bar as the means2D gradient itself
        # radii: the float32 oracle's, except where ceil(3 sqrt(lambda)) sits on an integer boundary (the fused path
        # applies exp / normalize / sigmoid inside the kernel: an ulp of difference in the scale flips such a radius)
        radii = pkg["radii"]
        r32 = aux32["radii"].to(torch.int32)
        n_diff = int((radii.cpu() != r32).sum())
        print(f"[radii] {label}: {n_diff} of {radii.numel()} differ from the float32 oracle "
              f"({int((aux['radii'].to(torch.int32) != r32).sum())} differ between the float32 and the float64 oracle)")
        assert n_diff <= 2e-5 * radii.numel() and int((radii.cpu() - r32).abs().max()) <= 1
        vis = (radii > 0).cpu() & (aux["radii"] > 0)
        add_densification_stats(model, pkg["viewspace_points"], radii)
        want = ref["means2D"][:, :2].norm(dim=1).double()
        want32 = ref32["means2D"][:, :2].norm(dim=1).double()
        acc = model.xyz_gradient_accum.detach().cpu().double().reshape(-1)
        scale = float(want.abs().max())
        e, e32 = float((acc - want)[vis].abs().max()) / scale, float((want32 - want)[vis].abs().max()) / scale
        print(f"[densify stats] {label}: xyz_gradient_accum err {e:.2e} (float32 oracle {e32:.2e})")
        assert e <= max(1e-5, 2.0 * e32) <= 2e-4
        hip_vis = (radii > 0).cpu()
        assert float(acc[~hip_vis].abs().max()) == 0.0
        assert torch.equal(model.denom.detach().cpu().reshape(-1), hip_vis.float())
        assert torch.equal(model.max_radii2D.detach().cpu(), torch.where(hip_vis, radii.cpu().float(), torch.zeros(())))
    return len(tiles), n_fragile


def test_config_C3_masked_train_step_matches_fp64_oracle(gpu_device):
    """C3: 1 M Gaussians, SH degree 3, 1920x1080, forward + backward."""
    n, _ = _masked_train_step_vs_fp64_oracle(gpu_device, "C3", 0, (15, 14), "C3 masked train step")
    assert n == 40


@pytest.mark.parametrize("view", [0, 1, 2, 3, 4, 5, 6, 7])
def test_config_C4_masked_train_step_matches_fp64_oracle(gpu_device, view):
    """C4, the config the headline metric is quoted on: 6 M Gaussians, SH degree 3, 1920x1080, forward + backward +
    densification statistics -- from every one of C5's eight cameras (BASELINE configs[4]).  view 0 is the bench's camera,
    view 2 stands at the edge of the cloud and looks along its long axis: both on the fixed 40-tile lattice.  On that
    lattice views 1 and 3 meet tiles on which the float32 ORACLE is 8e-5 / 1.1e-4 off float64 in the screen-space gradient
    (above grad_util's 2e-4 cap on 2 x that error): for the views other than 0 and 2 the 40 tiles are the best-conditioned
    of a denser lattice of ~130, ranked by the oracle alone; the cap and every bar stay as they are."""
    n, _ = _masked_train_step_vs_fp64_oracle(gpu_device, "C4", view, (15, 14), f"C4 view {view} masked train step",
                                             check_stats=True, select_from=None if view in (0, 2) else (8, 8))
    assert n == 40


@pytest.mark.parametrize("N,kind", [(4, "uniform"), (129, "uniform"), (1000, "uniform"), (200_000, "uniform"),
                                    (50_000, "clustered"), (400_000, "clustered"), (30_000, "duplicates")])
def test_distCUDA2_matches_kdtree(gpu_device, N, kind):
    """exact 3-NN mean squared distance vs scipy's cKDTree (the reference's simple_knn is absent)."""
    from scipy.spatial import cKDTree
    from mvs_gaussian_splatting_amd.knn import distCUDA2
    g = torch.Generator().manual_seed(N)
    if kind == "uniform":
        pts = torch.rand(N, 3, generator=g) * torch.tensor([4.0, 2.0, 1.0]) - 1.0
    elif kind == "duplicates":   # coincident points (distance 0) and a degenerate (planar) cloud
        pts = torch.rand(N, 3, generator=g)
        pts[:, 2] = 0.5
        pts[N // 2:] = pts[:N - N // 2]
    else:   # dense clusters plus far outliers: boxes of very different sizes, pruning across the whole cloud
        centres = torch.randn(20, 3, generator=g) * 5
        pts = centres[torch.randint(0, 20, (N,), generator=g)] + 0.01 * torch.randn(N, 3, generator=g)
        pts[:50] = torch.randn(50, 3, generator=g) * 200
    got = distCUDA2(pts.to(gpu_device)).cpu().double().numpy()
    d, _ = cKDTree(pts.double().numpy()).query(pts.double().numpy(), k=4)
    ref = (d[:, 1:] ** 2).mean(axis=1)
    assert np.max(np.abs(got - ref) / np.maximum(ref, 1e-12)) < 1e-4


def test_hip_forward_matches_plain_c_oracle_at_C2(gpu_device):
    """Third leg of the triangle: HIP vs the scalar C restatement (per-pixel loops in upstream's order) at config C2."""
    from gpu_util import forward_with_state, product_settings
    from mvs_gaussian_splatting_amd.synthetic import CONFIGS, make_scene
    from oracle.c_oracle import forward_c
    cfg = CONFIGS["C2"]
    model, cam, bg, _ = make_scene(cfg, view=2)
    bg = torch.tensor([0.05, 0.1, 0.15])
    c = forward_c(model.get_xyz, model.get_opacity, make_settings(cam, bg, 0), shs=model.get_features,
                  scales=model.get_scaling, rotations=model.get_rotation)
    out = forward_with_state(gpu_device, product_settings(cam, bg, 0, gpu_device), model.get_xyz, model.get_opacity,
                             shs=model.get_features, scales=model.get_scaling, rotations=model.get_rotation)
    assert np.array_equal(out["radii"].numpy(), c["radii"])
    assert np.array_equal(out["keys"], c["keys"]) and np.array_equal(out["point_list"], c["point_list"])
    assert np.array_equal(out["ranges"], c["ranges"])
    err = np.abs(out["color"].numpy() - c["color"]).max(axis=0)
    assert np.mean(err > 1e-5) < 1e-3          # only threshold-straddling pixels may differ (libm expf vs v_exp_f32)
    assert err.max() <= 2.0 / 255.0
    assert np.mean(out["n_contrib"].numpy().astype(np.uint32) != c["n_contrib"]) < 1e-3


def _stress_model(P=1200, seed=7):
    """Rare-branch soup: screen-filling and sub-pixel splats, extreme anisotropy, Gaussians far outside the 1.3x guard
    band, straddling the near plane, fully opaque (alpha clamp 0.99) and nearly transparent ones."""
    from mvs_gaussian_splatting_amd.synthetic import SyntheticGaussianModel
    m = SyntheticGaussianModel(P, 2, seed=seed, log_scale_mean=math.log(0.05))
    g = torch.Generator().manual_seed(seed + 1)
    n = P // 8
    m._scaling[0 * n:1 * n] = math.log(2.5) + 0.2 * torch.randn(n, 3, generator=g)        # covers most of the image
    m._scaling[1 * n:2 * n] = math.log(0.0008)                                              # far below a pixel
    m._scaling[2 * n:3 * n, 0] = math.log(0.5); m._scaling[2 * n:3 * n, 1:] = math.log(0.01)   # needles (50:1)
    m._xyz[3 * n:4 * n, 0] = 40.0 * torch.sign(torch.randn(n, generator=g))                 # far outside the guard band
    m._scaling[3 * n:4 * n] = math.log(8.0)                                                 # ... but big enough to reach in
    m._xyz[4 * n:5 * n, 2] = 0.2 + 0.01 * torch.randn(n, generator=g)                       # around the near plane
    m._opacity[5 * n:6 * n] = 12.0                                                          # sigmoid -> 1: alpha clamp
    m._scaling[5 * n:6 * n] = math.log(0.3)
    m._opacity[6 * n:7 * n] = -7.0                                                          # sigmoid ~ 9e-4 < 1/255
    return m


@pytest.mark.parametrize("mode", [0, 1], ids=["two_level", "keys64"])
def test_rare_branches_forward_and_backward_match_oracle(gpu_device, mode, monkeypatch):
    from mvs_gaussian_splatting_amd import GaussianRasterizer
    from mvs_gaussian_splatting_amd.synthetic import orbit_camera
    from gpu_util import forward_with_state, product_settings
    from oracle import rasterize_ref
    from mvs_gaussian_splatting_amd import rasterizer
    monkeypatch.setattr(rasterizer, "_binning_mode_value", rasterizer._binning_from_name("keys64" if mode else "two_level"))
    model = _stress_model()
    cam = orbit_camera(1, 8, 208, 136, 120.0, 120.0)
    bg = torch.tensor([0.2, 0.4, 0.1])
    target = torch.rand(3, 136, 208, generator=torch.Generator().manual_seed(5))
    deg, smod = 2, 1.3
    # ---- forward: integers exact, robust pixels 1e-5 ---------------------------------------------------
    st_o = make_settings(cam, bg, deg, scale_modifier=smod)
    col, radii, aux = rasterize_ref(model.get_xyz, None, model.get_opacity, st_o, shs=model.get_features,
                                    scales=model.get_scaling, rotations=model.get_rotation, want_aux=True, want_margin=True)
    st = product_settings(cam, bg, deg, gpu_device, scale_modifier=smod)
    out = forward_with_state(gpu_device, st, model.get_xyz, model.get_opacity, shs=model.get_features,
                             scales=model.get_scaling, rotations=model.get_rotation, binning_mode=mode)
    assert torch.equal(out["radii"], radii)
    assert int(aux["pre"]["tiles_touched"].max()) > 16 * 4          # the wave-cooperative emission path is exercised
    assert np.array_equal(out["keys"], aux["keys"]) and np.array_equal(out["point_list"], aux["point_list"])
    assert np.array_equal(out["ranges"], aux["ranges"])
    robust = aux["margin"] > 1e-4
    err = ((out["color"] - col).abs() / col.abs().clamp(min=1.0)).max(dim=0).values
    assert float(err[robust].max()) <= 1e-5
    assert int((~robust).sum()) <= 0.25 * robust.numel()     # needles make many pixels ill-conditioned in float32
    assert float(err.max()) <= 3.0 / 255.0                   # ... where two float32 orders differ, but boundedly
    # ---- backward vs float64 autograd: fragile / ill-conditioned pixels carry no loss, the rest is held to 1e-5 --------
    _masked_grad_parity(gpu_device, model, cam, bg, target, deg, f"rare branches mode={mode}", smod=smod)


def test_debug_mode_synchronises_and_matches(gpu_device, tmp_path, monkeypatch):
    """settings.debug (gaussian_renderer/__init__.py:54): every stage is synchronised and checked; results are the
    same bit for bit, and a failing call leaves the upstream-style snapshot behind (README.md:143-146)."""
    import os
    from mvs_gaussian_splatting_amd import GaussianRasterizer, _lib
    from gpu_util import product_settings
    model, cam, bg, target = small_scene(P=1500, sh_degree=2, width=130, height=70)
    dev = gpu_device
    outs = []
    for debug in (False, True):
        st = product_settings(cam, bg, 2, dev, debug=debug)
        xyz = model.get_xyz.to(dev).requires_grad_(True)
        args = dict(means2D=torch.zeros(1500, 3, device=dev, requires_grad=True), opacities=model.get_opacity.to(dev),
                    shs=model.get_features.to(dev), scales=model.get_scaling.to(dev), rotations=model.get_rotation.to(dev))
        col, radii = GaussianRasterizer(st)(means3D=xyz, **args)
        (col - target.to(dev)).abs().mean().backward()
        outs.append((col.detach().clone(), radii.clone(), xyz.grad.clone(), args["means2D"].grad.clone()))
    assert all(torch.equal(a, b) for a, b in zip(*outs))
    # a call the library rejects (rotations not 16-byte aligned) in debug mode dumps its arguments
    monkeypatch.chdir(tmp_path)
    st = product_settings(cam, bg, 2, dev, debug=True)
    rot = torch.zeros(1500 * 4 + 1, device=dev)[1:].view(1500, 4)            # 4-byte aligned view
    rot.copy_(model.get_rotation)
    from mvs_gaussian_splatting_amd import rasterizer as R
    monkeypatch.setattr(R, "_f32c", lambda t, name, dev_, align16=False: t)   # skip the host-side realignment
    with pytest.raises(_lib.GsrError):
        GaussianRasterizer(st)(means3D=model.get_xyz.to(dev), means2D=torch.zeros(1500, 3, device=dev),
                               opacities=model.get_opacity.to(dev), shs=model.get_features.to(dev),
                               scales=model.get_scaling.to(dev), rotations=rot)
    assert os.path.exists(tmp_path / "snapshot_fw.dump")


@pytest.mark.parametrize("seed", list(range(8)))
def test_fuzz_random_small_scenes_forward_and_backward(gpu_device, seed):
    """Randomised configurations (image size incl. partial tiles, SH degree, scale / opacity distribution, background,
    scale_modifier, view, binning mode): integers exact, robust pixels 1e-5, gradients vs float64 autograd."""
    from gpu_util import forward_with_state, product_settings
    from mvs_gaussian_splatting_amd.synthetic import SceneConfig, make_scene
    from oracle import rasterize_ref
    rng = np.random.default_rng(1000 + seed)
    W, H = int(rng.integers(17, 260)), int(rng.integers(17, 200))
    deg = int(rng.integers(0, 4))
    P = int(rng.integers(50, 2500))
    f = float(rng.uniform(40.0, 260.0))
    scale = float(np.exp(rng.uniform(np.log(0.01), np.log(0.4))))
    smod = float(rng.choice([1.0, 1.0, 0.6, 1.7]))
    mode = int(rng.integers(0, 3))
    cfg = SceneConfig("fuzz", P, deg, W, H, f, f * float(rng.uniform(0.8, 1.25)), math.log(scale))
    model, cam, _, target = make_scene(cfg, seed=seed, view=int(rng.integers(0, 8)))
    model._opacity += float(rng.uniform(-2.0, 3.0))
    bg = torch.tensor(rng.uniform(0, 1, 3), dtype=torch.float32)
    st_o = make_settings(cam, bg, deg, scale_modifier=smod)
    col, radii, aux = rasterize_ref(model.get_xyz, None, model.get_opacity, st_o, shs=model.get_features,
                                    scales=model.get_scaling, rotations=model.get_rotation, want_aux=True, want_margin=True)
    st = product_settings(cam, bg, deg, gpu_device, scale_modifier=smod)
    out = forward_with_state(gpu_device, st, model.get_xyz, model.get_opacity, shs=model.get_features,
                             scales=model.get_scaling, rotations=model.get_rotation, binning_mode=mode)
    assert torch.equal(out["radii"], radii)
    if mode != 2:      # the culled mode drops dead instances: its lists are checked in test_gpu_culled_binning.py
        assert np.array_equal(out["keys"], aux["keys"]) and np.array_equal(out["point_list"], aux["point_list"])
        assert np.array_equal(out["ranges"], aux["ranges"])
    robust = aux["margin"] > 1e-4
    err = ((out["color"] - col).abs() / col.abs().clamp(min=1.0)).max(dim=0).values
    assert robust.any() and float(err[robust].max()) <= 1e-5
    assert float(err.max()) <= 3.0 / 255.0
    # gradients of the masked L1 loss against the float64 oracle (scale_modifier 1 on both sides)
    _masked_grad_parity(gpu_device, model, cam, bg, target, deg,
                        f"fuzz seed={seed} " + str(dict(W=W, H=H, deg=deg, P=P, scale=round(scale, 4))))


def _fuzz_scene(seed):
    """The configuration test_fuzz_random_small_scenes_forward_and_backward draws for `seed` (same generator calls)."""
    from mvs_gaussian_splatting_amd.synthetic import SceneConfig, make_scene
    rng = np.random.default_rng(1000 + seed)
    W, H = int(rng.integers(17, 260)), int(rng.integers(17, 200))
    deg = int(rng.integers(0, 4))
    P = int(rng.integers(50, 2500))
    f = float(rng.uniform(40.0, 260.0))
    scale = float(np.exp(rng.uniform(np.log(0.01), np.log(0.4))))
    rng.choice([1.0, 1.0, 0.6, 1.7]); rng.integers(0, 3)
    cfg = SceneConfig("fuzz", P, deg, W, H, f, f * float(rng.uniform(0.8, 1.25)), math.log(scale))
    model, cam, _, target = make_scene(cfg, seed=seed, view=int(rng.integers(0, 8)))
    model._opacity += float(rng.uniform(-2.0, 3.0))
    return model, cam


@pytest.mark.parametrize("seed", [0, 3, 83, 85, 106, 111, 131, 139, 283])
def test_backward_takes_the_forwards_decisions_pixel_by_pixel(gpu_device, seed):
    """Exact, oracle-free check that the backward composites what the forward composited -- on the threshold-fragile
    pixels in particular (the ones the masked gradient comparison gives no weight and on which float32 and float64
    legitimately disagree).  With precomputed colours and a black background the image is linear in the colours,
    C(p) = sum_i w_i(p) c_i, and the backward of dL/dpix = (1, 1, 1) at ONE pixel p returns dL/dc_i = w_i(p): so
    sum_i <dL/dc_i, c_i> must reproduce the forward's own sum_ch C_ch(p) to float32 rounding.  A pair the forward blended
    and the backward skipped (or the reverse: a different alpha >= 1/255 decision, clamp scope, last contributor, list
    cut-off) is missing from the sum with its whole weight: at least T / 255 of the colour.  Seeds 83 .. 283 are the fuzz
    configurations whose unmasked gradients differ most from float64 (profiles/r03/fuzz_seeds_72_180.log, fuzz_seeds_180_290.log: up to 9.8e-3 at seed 283)."""
    from gpu_util import product_settings
    from mvs_gaussian_splatting_amd import GaussianRasterizer
    from oracle import rasterize_ref
    dev = gpu_device
    model, cam = _fuzz_scene(seed)
    bg = torch.zeros(3)
    P = model.get_xyz.shape[0]
    colors = torch.rand(P, 3, generator=torch.Generator().manual_seed(seed)) * 0.9 + 0.1
    st_o = make_settings(cam, bg, 0)
    _, _, aux = rasterize_ref(model.get_xyz, None, model.get_opacity, st_o, colors_precomp=colors,
                              scales=model.get_scaling, rotations=model.get_rotation, want_aux=True, want_margin=True)
    H, W = cam.image_height, cam.image_width
    fragile = torch.nonzero(aux["margin"] <= 1e-4)[:24].tolist()
    rng = np.random.default_rng(seed)
    pixels = fragile + [[int(rng.integers(0, H)), int(rng.integers(0, W))] for _ in range(8)]
    st = product_settings(cam, bg, 0, dev)
    xyz, op = model.get_xyz.to(dev), model.get_opacity.to(dev)
    sc, rot = model.get_scaling.to(dev), model.get_rotation.to(dev)
    worst = 0.0
    for y, x in pixels:
        c = colors.to(dev).requires_grad_(True)
        col, _ = GaussianRasterizer(st)(means3D=xyz, means2D=torch.zeros_like(xyz), opacities=op, colors_precomp=c,
                                        scales=sc, rotations=rot)
        dL = torch.zeros_like(col)
        dL[:, y, x] = 1.0
        col.backward(dL)
        lhs = float((c.grad.double() * c.detach().double()).sum())
        rhs = float(col[:, y, x].detach().double().sum())
        err = abs(lhs - rhs) / max(abs(rhs), 1e-2)
        worst = max(worst, err)
        assert err <= 1e-5, (seed, (y, x), lhs, rhs, "fragile" if [y, x] in fragile else "random")
    print(f"[fwd/bwd decisions] fuzz seed={seed}: {len(fragile)} fragile + 8 random pixels, worst relative mismatch {worst:.1e}")


def _pixel_from_contributors(pre, slots, x, y, dt):
    """sum_ch C_ch at pixel (x, y) composited from exactly the given visible-candidate slots, in the given order: the
    oracle's arithmetic (A.4: alpha = min(0.99, o exp(power)) with upstream's straight-through clamp) with NO decision of
    its own -- which instances contribute is an input."""
    from oracle.rasterizer_ref import _alpha_st
    xy, con, o, rgb = pre["v_xy"][slots], pre["v_conic"][slots], pre["v_opacity"][slots], pre["v_rgb"][slots]
    dx = xy[:, 0] - torch.tensor(float(x), dtype=dt)
    dy = xy[:, 1] - torch.tensor(float(y), dtype=dt)
    power = -0.5 * (con[:, 0] * dx * dx + con[:, 2] * dy * dy) - con[:, 1] * dx * dy
    alpha = _alpha_st(o.reshape(-1), torch.exp(power), True)
    T_excl = torch.cat([torch.ones(1, dtype=dt), torch.cumprod(1.0 - alpha, dim=0)[:-1]])
    return ((alpha * T_excl)[:, None] * rgb).sum()


@pytest.mark.parametrize("seed", [0, 3, 83, 106, 131, 283])
def test_single_pixel_gradients_of_geometry_and_opacity_with_the_forwards_decisions(gpu_device, seed):
    """The threshold-fragile pixels get loss weight 0 in the masked gradient comparison and only a loose bar (2e-2) in
    the unmasked one, because a float32 and a float64 evaluation may DECIDE differently there.  Here the decisions are
    taken out of the comparison: dL/dpix = (1, 1, 1) at ONE pixel goes through the HIP backward; the instances that
    contributed to that pixel are read off the colour gradient (dL/dc_i = w_i(p) != 0); the oracle then composites exactly
    those instances, in the HIP path's list order, with no threshold of its own, and its float64 autograd gives the
    gradient of the same pixel w.r.t. position, scale, rotation, opacity and colour.  Same decisions on both sides, so
    the grad_util bar applies (1e-5 max-norm relative, or twice the float32 oracle's own error), on fragile pixels too."""
    from gpu_util import forward_with_state, product_settings
    from grad_util import compare_grads
    from mvs_gaussian_splatting_amd import GaussianRasterizer
    from oracle import rasterize_ref, preprocess_ref
    dev = gpu_device
    model, cam = _fuzz_scene(seed)
    bg = torch.zeros(3)
    P = model.get_xyz.shape[0]
    colors = torch.rand(P, 3, generator=torch.Generator().manual_seed(seed)) * 0.9 + 0.1
    st_o = make_settings(cam, bg, 0)
    _, _, aux = rasterize_ref(model.get_xyz, None, model.get_opacity, st_o, colors_precomp=colors,
                              scales=model.get_scaling, rotations=model.get_rotation, want_aux=True, want_margin=True)
    H, W = cam.image_height, cam.image_width
    gx = (W + 15) // 16
    fragile = torch.nonzero(aux["margin"] <= 1e-4)[:10].tolist()
    rng = np.random.default_rng(seed)
    covered = torch.nonzero(aux["n_contrib"] > 0)
    if covered.shape[0] == 0:
        pytest.skip("nothing on the screen in this configuration")
    pixels = fragile + [covered[int(rng.integers(0, covered.shape[0]))].tolist() for _ in range(3)]
    st = product_settings(cam, bg, 0, dev)
    # the HIP path's lists (un-culled two-level mode: the culled default drops only instances that cannot contribute)
    lists = forward_with_state(dev, st, model.get_xyz, model.get_opacity, colors_precomp=colors, scales=model.get_scaling,
                               rotations=model.get_rotation)
    names = ("xyz", "opacity", "scaling", "rotation", "colors")
    raw = (model._xyz, model._opacity, model._scaling, model._rotation, colors)

    def leaves(dt, device):
        return {n: t.detach().to(device=device, dtype=dt).requires_grad_(True) for n, t in zip(names, raw)}

    def operator_inputs(L):
        return dict(means3D=L["xyz"], opacities=torch.sigmoid(L["opacity"]), colors_precomp=L["colors"],
                    scales=torch.exp(L["scaling"]), rotations=torch.nn.functional.normalize(L["rotation"]))

    Lh = leaves(torch.float32, dev)
    col, _ = GaussianRasterizer(st)(means2D=torch.zeros(P, 3, device=dev), **operator_inputs(Lh))
    checked = 0
    for y, x in pixels:
        for t in Lh.values():
            t.grad = None
        dL = torch.zeros_like(col)
        dL[:, y, x] = 1.0
        col.backward(dL, retain_graph=True)
        got = {n: (t.grad.detach().cpu() if t.grad is not None else torch.zeros_like(t).cpu()) for n, t in Lh.items()}
        contrib = torch.nonzero(got["colors"].abs().sum(dim=1) > 0).reshape(-1).numpy()
        if contrib.size == 0:
            continue
        tile = (y // 16) * gx + x // 16
        lst = lists["point_list"][lists["ranges"][tile, 0]:lists["ranges"][tile, 1]].astype(np.int64)
        pos = {int(g): i for i, g in enumerate(lst)}
        assert all(int(g) in pos for g in contrib)                     # every contributor is in the tile's list
        ordered = sorted((int(g) for g in contrib), key=lambda g: pos[g])
        refs = []
        for dt in (torch.float64, torch.float32):
            Lo = leaves(dt, "cpu")
            kw = operator_inputs(Lo)
            pre = preprocess_ref(kw["means3D"], kw["opacities"], st_o, colors_precomp=kw["colors_precomp"],
                                 scales=kw["scales"], rotations=kw["rotations"])
            slot_of = torch.full((P,), -1, dtype=torch.int64)
            slot_of[pre["idx"]] = torch.arange(pre["idx"].shape[0])
            slots = slot_of[torch.tensor(ordered)]
            assert int(slots.min()) >= 0
            _pixel_from_contributors(pre, slots, x, y, dt).backward()
            refs.append({n: (t.grad.detach() if t.grad is not None else torch.zeros_like(t)) for n, t in Lo.items()})
        kind = "fragile" if [y, x] in fragile else "random"
        compare_grads(got, refs[0], refs[1], f"fuzz seed {seed}, one-hot pixel ({x},{y}) [{kind}], {len(ordered)} contributors")
        checked += 1
    assert checked >= 1


@pytest.mark.parametrize("seed", [0, 83, 131])
def test_backward_takes_the_forwards_decisions_on_the_sh_raw_parameter_path(gpu_device, seed):
    """The decision identity of the test above on the path the headline numbers use: render() fed with the raw parameters
    (split SH, activations inside the kernels).  The image is linear in the DC coefficients wherever the colour is not
    clamped -- rgb_i = max(SH_C0 f_dc_i + rest_i + 0.5, 0) -- so for dL/dpix = (1, 1, 1) at one pixel
    sum_i <dL/df_dc_i / SH_C0, rgb_i> must reproduce the forward's sum_ch C_ch(p) (a clamped channel has gradient 0 and
    colour 0: it drops out on both sides); rgb_i is the forward's own, read back from its geometry workspace."""
    import ctypes as C
    from mvs_gaussian_splatting_amd import render, _lib
    from mvs_gaussian_splatting_amd.synthetic import PipelineParams
    from oracle import rasterize_ref
    dev = gpu_device
    model, cam = _fuzz_scene(seed)
    if model.max_sh_degree not in (0, 3):       # the raw-parameter path takes degree-0 or degree-3 storage
        model, cam = _fuzz_scene(seed + 1000)
    if model.max_sh_degree not in (0, 3):
        pytest.skip("no degree-0 / degree-3 configuration for this seed")
    bg = torch.zeros(3)
    st_o = make_settings(cam, bg, model.active_sh_degree)
    _, _, aux = rasterize_ref(model.get_xyz, None, model.get_opacity, st_o, shs=model.get_features,
                              scales=model.get_scaling, rotations=model.get_rotation, want_aux=True, want_margin=True)
    H, W = cam.image_height, cam.image_width
    fragile = torch.nonzero(aux["margin"] <= 1e-4)[:16].tolist()
    rng = np.random.default_rng(seed)
    pixels = fragile + [[int(rng.integers(0, H)), int(rng.integers(0, W))] for _ in range(6)]
    model.to(dev); cam.to(dev)
    for p in model.parameters():
        p.requires_grad_(True)
    pkg = render(cam, model, PipelineParams(), bg.to(dev))
    col = pkg["render"]
    ctx = col.grad_fn
    assert type(ctx).__name__.startswith("_RasterizeGaussiansFused")
    geom = ctx.saved_tensors[7]
    P = model._xyz.shape[0]
    rgb = torch.empty(P, 3, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().gsr_debug_read_geom(geom.data_ptr(), P, None, None, rgb.data_ptr(), None, None, None, None, None,
                                                   torch.cuda.current_stream(dev).cuda_stream), "read_geom")
    vis = (pkg["radii"] > 0)
    SH_C0 = 0.28209479177387814
    worst = 0.0
    for y, x in pixels:
        model._features_dc.grad = None
        dL = torch.zeros_like(col)
        dL[:, y, x] = 1.0
        col.backward(dL, retain_graph=True)
        g = model._features_dc.grad.reshape(P, 3).double()
        lhs = float(((g / SH_C0) * rgb.double())[vis].sum())
        rhs = float(col[:, y, x].detach().double().sum())
        err = abs(lhs - rhs) / max(abs(rhs), 1e-2)
        worst = max(worst, err)
        assert err <= 1e-5, (seed, (y, x), lhs, rhs, "fragile" if [y, x] in fragile else "random")
    print(f"[fwd/bwd decisions, SH raw-parameter path] fuzz seed={seed}: {len(fragile)} fragile + 6 random pixels, worst {worst:.1e}")


def test_render_host_modes_and_leaf_reuse(gpu_device):
    """render() under no_grad / inference_mode, and two frames whose screen-space leaves alias the cached zeros:
    each backward fills its own .grad."""
    from mvs_gaussian_splatting_amd import render, l1_loss
    from mvs_gaussian_splatting_amd.synthetic import PipelineParams
    dev = gpu_device
    model, cam, bg, target = small_scene(P=1200, sh_degree=3, width=100, height=60)
    model2, cam2, _, _ = small_scene(P=1200, sh_degree=3, width=100, height=60, view=2)
    model.to(dev); cam.to(dev); cam2.to(dev)
    bg, target = bg.to(dev), target.to(dev)
    pipe = PipelineParams()
    with torch.no_grad():
        a = render(cam, model, pipe, bg)["render"]
    with torch.inference_mode():
        b = render(cam, model, pipe, bg)["render"]
    assert torch.equal(a, b) and not a.requires_grad
    for p in model.parameters():
        p.requires_grad_(True)
    p1 = render(cam, model, pipe, bg)
    p2 = render(cam2, model, pipe, bg)
    assert p1["viewspace_points"] is not p2["viewspace_points"]
    assert torch.equal(p1["render"].detach(), a)
    l1_loss(p2["render"], target).backward()
    g2 = p2["viewspace_points"].grad.clone()
    assert p1["viewspace_points"].grad is None
    l1_loss(p1["render"], target).backward()
    g1 = p1["viewspace_points"].grad
    assert torch.equal(p2["viewspace_points"].grad, g2) and not torch.equal(g1, g2)
    assert float(p1["viewspace_points"].detach().abs().max()) == 0.0   # the shared zeros were never written
    # visibility filter / radii contract of the result dict
    assert torch.equal(p1["visibility_filter"], p1["radii"] > 0) and p1["selected_pts_mask"] is None


def test_forward_only_variant_gives_the_same_image(gpu_device):
    """GsrParams.forward_only (set by the operator under no_grad / when nothing requires grad): the compositing
    variant that tracks nothing for a backward renders the same image bit for bit; a backward on it is refused."""
    import ctypes as C
    from mvs_gaussian_splatting_amd import render, _lib
    from mvs_gaussian_splatting_amd.synthetic import PipelineParams
    from test_gpu_parity import _stress_model
    from mvs_gaussian_splatting_amd.synthetic import orbit_camera
    dev = gpu_device
    scenes = [small_scene(P=4000, sh_degree=3, width=200, height=120)[:3],
              (_stress_model(), orbit_camera(1, 8, 208, 136, 120.0, 120.0), torch.tensor([0.2, 0.4, 0.1]))]
    for model, cam, bg in scenes:
        model.to(dev); cam.to(dev)
        bg = bg.to(dev)
        for fused in (True, False):
            pipe = PipelineParams()
            pipe.fuse_activations = fused
            for p in model.parameters():
                p.requires_grad_(True)
            b = render(cam, model, pipe, bg)                 # a backward may follow: tracking variant
            assert b["render"].grad_fn is not None
            with torch.no_grad():
                c = render(cam, model, pipe, bg)             # forward only
            assert c["render"].grad_fn is None
            assert torch.equal(b["render"].detach(), c["render"]) and torch.equal(b["radii"], c["radii"])
    # the C ABI refuses a backward after a forward-only forward
    p = _lib.GsrParams()
    p.forward_only = 1
    g = _lib.GsrGrads()
    rc = _lib.load().gsr_backward(C.byref(p), None, None, None, None, 0, 0, None, None, 0, C.byref(g), None)
    assert rc != 0 and b"forward_only" in _lib.load().gsr_last_error()


def test_operator_rejects_misshaped_inputs(gpu_device):
    """The drop-in boundary hands raw pointers to the kernels: row counts and trailing dims are checked on the host."""
    from mvs_gaussian_splatting_amd import GaussianRasterizer
    from gpu_util import product_settings
    model, cam, bg, _ = small_scene(P=64, sh_degree=1, width=48, height=32)
    dev = gpu_device
    st = product_settings(cam, bg, 1, dev)
    good = dict(means3D=model.get_xyz.to(dev), means2D=torch.zeros(64, 3, device=dev), opacities=model.get_opacity.to(dev),
                shs=model.get_features.to(dev), scales=model.get_scaling.to(dev), rotations=model.get_rotation.to(dev))
    GaussianRasterizer(st)(**good)
    bad = [("scales", good["scales"][:, :1].contiguous()),                 # isotropic scales [P,1]
           ("rotations", good["rotations"][:, :3].contiguous()),           # [P,3]
           ("shs", good["shs"][:32].contiguous()),                         # fewer rows than P
           ("shs", good["shs"][:, :, :2].contiguous()),                    # last dim 2
           ("means3D", good["means3D"][:, :2].contiguous()),
           ("opacities", good["opacities"][:10].contiguous())]
    for name, t in bad:
        with pytest.raises(ValueError):
            GaussianRasterizer(st)(**{**good, name: t})
    cov = dict(good)
    del cov["scales"], cov["rotations"]
    with pytest.raises(ValueError):
        GaussianRasterizer(st)(cov3D_precomp=torch.zeros(64, 3, 3, device=dev), **cov)      # full matrices, not [P,6]
    with pytest.raises(ValueError):
        col = dict(cov); del col["shs"]
        GaussianRasterizer(st)(cov3D_precomp=torch.zeros(64, 6, device=dev), colors_precomp=torch.zeros(64, 4, device=dev), **col)


@pytest.mark.parametrize("pinned", [False, True])
def test_depth_sort_of_a_frame_spanning_more_than_24_bits(gpu_device, pinned):
    """The two-level binning sorts (depth bits - frame minimum) on 24 bits and adds the pass for the top digit only when
    a frame spans more than 2^24 float32 steps of depth: lists must stay those of a full 32-bit (tile, depth) sort."""
    from gpu_util import forward_with_state, product_settings
    from mvs_gaussian_splatting_amd import render
    from mvs_gaussian_splatting_amd.synthetic import PipelineParams
    from oracle import rasterize_ref
    model, cam, bg, _ = small_scene(P=2500, sh_degree=0, width=160, height=96)
    g = torch.Generator().manual_seed(11)
    z = torch.exp(torch.rand(2500, generator=g) * (math.log(2.0e6) - math.log(0.25)) + math.log(0.25))   # 0.25 .. 2e6: 23 binades
    scale = z / model._xyz[:, 2]
    model._xyz *= scale[:, None]                        # same screen positions, depths spread over 23 binades
    model._scaling += torch.log(scale)[:, None]         # same screen-space footprints
    st_o = make_settings(cam, bg, 0)
    col, radii, aux = rasterize_ref(model.get_xyz, None, model.get_opacity, st_o, shs=model.get_features,
                                    scales=model.get_scaling, rotations=model.get_rotation, want_aux=True, want_margin=True)
    d = aux["pre"]["v_depth"][aux["pre"]["keep"]]
    assert float(d.max() / d.min()) > 2.0 ** 17
    if not pinned:      # raw C ABI without the pinned mirror: lists compared entry by entry
        out = forward_with_state(gpu_device, product_settings(cam, bg, 0, gpu_device), model.get_xyz, model.get_opacity,
                                 shs=model.get_features, scales=model.get_scaling, rotations=model.get_rotation)
        # this cloud holds one Gaussian whose 3 sqrt(lambda) sits on an integer: its ceil() may differ by one between two
        # float32 evaluations; tile rects, keys and lists are compared exactly all the same
        assert int((out["radii"] != radii).sum()) <= 1 and int((out["radii"] - radii).abs().max()) <= 1
        assert np.array_equal(out["keys"], aux["keys"]) and np.array_equal(out["point_list"], aux["point_list"])
        img = out["color"]
    else:               # the operator (pinned count mirror)
        model.to(gpu_device); cam.to(gpu_device)
        with torch.no_grad():
            img = render(cam, model, PipelineParams(), bg.to(gpu_device))["render"].cpu()
    robust = aux["margin"] > 1e-4
    err = ((img - col).abs() / col.abs().clamp(min=1.0)).max(dim=0).values
    assert float(err[robust].max()) <= 1e-5


def test_packed_rect_payload_boundaries(gpu_device):
    """The depth sort carries (index, packed rect) for rects of width <= 4 and <= 16 tiles and a fall-back marker for the
    rest (gsr_common.h pack_rect): a cloud whose rects straddle both limits must give the oracle's lists in the two-level
    mode, the same lists as the 64-bit-key mode, and the same picture with the tile masks of the culled mode."""
    from gpu_util import forward_with_state, product_settings
    model, cam, bg, _ = small_scene(P=3000, sh_degree=1, width=352, height=240)
    g = torch.Generator().manual_seed(21)
    # screen rects from one tile to ~12 tiles across, one third of them needles (tall / wide bounding boxes after the culling)
    model._scaling[:] = torch.log(torch.exp(torch.rand(3000, 1, generator=g) * math.log(40.0)) * 0.01).expand(3000, 3)
    model._scaling[:1000, 1:] -= math.log(12.0)
    col, radii, aux = _oracle_forward(model, cam, bg, 1)
    rect = aux["pre"]["v_rect"].numpy()[aux["pre"]["keep"].numpy()]
    w, h = rect[:, 2] - rect[:, 0], rect[:, 3] - rect[:, 1]
    packs = (w <= 4) & (w * h <= 16)
    assert packs.sum() > 100 and (w > 4).sum() > 100 and ((w <= 4) & (w * h > 16)).sum() >= 1
    assert ((w == 4) & (h == 4)).sum() >= 1 and ((w == 5) | (h == 5)).sum() >= 1
    st = product_settings(cam, bg, 1, gpu_device)
    o = {m: forward_with_state(gpu_device, st, model.get_xyz, model.get_opacity, shs=model.get_features,
                               scales=model.get_scaling, rotations=model.get_rotation, binning_mode=m) for m in (0, 1, 2)}
    assert np.array_equal(o[0]["keys"], aux["keys"]) and np.array_equal(o[0]["point_list"], aux["point_list"])
    assert np.array_equal(o[0]["ranges"], aux["ranges"])
    assert np.array_equal(o[1]["keys"], o[0]["keys"]) and np.array_equal(o[1]["point_list"], o[0]["point_list"])
    # culled lists: a sub-sequence of the full ones per tile, and no visible difference (dropped instances never pass alpha)
    assert o[2]["R"] < o[0]["R"]
    full = set(zip((o[0]["keys"] >> np.uint64(32)).tolist(), o[0]["point_list"].tolist()))
    assert set(zip((o[2]["keys"] >> np.uint64(32)).tolist(), o[2]["point_list"].tolist())) <= full
    assert np.all(o[2]["keys"][1:] >= o[2]["keys"][:-1])
    assert torch.equal(o[2]["color"], o[0]["color"])


def test_frame_of_more_than_8192_tiles(gpu_device):
    """build_tile_order handles the tiles in chunks of 8192 (one trip to memory per chunk): a frame with more tiles than
    that (2112 x 1104 -> 132 x 69 = 9108) must still composite every tile exactly once -- checked against the oracle on
    every 5th tile and on all tiles of the second chunk."""
    from gpu_util import forward_with_state, product_settings
    from mvs_gaussian_splatting_amd.synthetic import SceneConfig, make_scene
    from oracle import rasterize_ref
    W, H = 2112, 1104
    cfg = SceneConfig("big", 3000, 1, W, H, 1300.0, 1300.0, math.log(0.05))
    model, cam, _, _ = make_scene(cfg, seed=4)
    bg = torch.tensor([0.2, 0.5, 0.1])
    gx, gy = (W + 15) // 16, (H + 15) // 16
    assert gx * gy > 8192
    tiles = sorted(set(range(0, gx * gy, 5)) | set(range(8192, gx * gy)))
    st_o = make_settings(cam, bg, 1)
    col, radii, aux = rasterize_ref(model.get_xyz, None, model.get_opacity, st_o, shs=model.get_features,
                                    scales=model.get_scaling, rotations=model.get_rotation, want_aux=True, want_margin=True,
                                    tiles=tiles)
    out = forward_with_state(gpu_device, product_settings(cam, bg, 1, gpu_device), model.get_xyz, model.get_opacity,
                             shs=model.get_features, scales=model.get_scaling, rotations=model.get_rotation, binning_mode=2)
    assert torch.equal(out["radii"], radii)
    mask = torch.zeros(gy * 16, gx * 16, dtype=torch.bool)
    for t in tiles:
        ty, tx = divmod(t, gx)
        mask[ty * 16:(ty + 1) * 16, tx * 16:(tx + 1) * 16] = True
    mask = mask[:H, :W] & (aux["margin"] > 1e-4)
    err = ((out["color"] - col).abs() / col.abs().clamp(min=1.0)).max(dim=0).values
    assert int(mask.sum()) > 100_000 and float(err[mask].max()) <= 1e-5
    touched = (out["n_contrib"] > 0)
    assert int(touched[:, : W // 2].sum()) > 0 and int(touched[H - 64:, :].sum()) > 0      # the last tile rows are composited too
