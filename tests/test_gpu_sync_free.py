"""The forward without a host round-trip (ABI v11 ``gsr_forward``: caller-side capacity, counts read on the device,
overflow visible in pinned memory) against the two-call forward it replaces after the first frame
(``gsr_forward_preprocess`` + ``gsr_forward_render``, i.e. upstream's per-forward ``num_rendered`` read-back,
``gaussian_renderer/__init__.py:257-265``): images, radii, per-pixel state and every gradient must be BIT-IDENTICAL;
a frame that does not fit its capacity is issued again before the operator returns (default, "verified" mode: the
operator never raises and never hands out an incomplete image, whatever the camera sequence -- ``train.py:81-107``,
``render.py:32-35``) or reported loudly (opt-in "deferred" mode); frames whose depths span more than 24 bits take the
device-predicated fourth sort pass.  Also the fused densification statistics (``GsrGrads.stats_*``) and the
roctx switch."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from conftest import small_scene

pytestmark = pytest.mark.gpu


def _scene(dev, P=6000, wide_depth=False, seed=0):
    model, cam, bg, target = small_scene(P=P, sh_degree=2, width=352, height=208, scale=0.03, seed=seed)
    if wide_depth:
        g = torch.Generator().manual_seed(11)
        z = torch.exp(torch.rand(P, generator=g) * (math.log(2.0e6) - math.log(0.25)) + math.log(0.25))
        s = z / model._xyz[:, 2]
        model._xyz *= s[:, None]
        model._scaling += torch.log(s)[:, None]
    model.to(dev); cam.to(dev)
    for p in model.parameters():
        p.requires_grad_(True)
    return model, cam, bg.to(dev), target.to(dev)


def _step(model, cam, bg, target, fused=True, stats=False):
    from mvs_gaussian_splatting_amd import render, l1_loss
    from mvs_gaussian_splatting_amd.synthetic import PipelineParams
    for p in model.parameters():
        p.grad = None
    pipe = PipelineParams()
    pipe.fuse_activations = fused
    pipe.fuse_densify_stats = stats
    pkg = render(cam, model, pipe, bg)
    l1_loss(pkg["render"], target).backward()
    return pkg, [p.grad.detach().clone() for p in model.parameters()] + [pkg["viewspace_points"].grad.detach().clone()]


@pytest.fixture()
def fresh_state(monkeypatch):
    """Every test starts without a remembered capacity and leaves none behind."""
    from mvs_gaussian_splatting_amd import rasterizer
    rasterizer.synchronize_counts()
    rasterizer._states.clear()
    monkeypatch.setattr(rasterizer, "_sync_free_value", rasterizer.SYNC_VERIFIED)
    yield rasterizer
    try:
        rasterizer.synchronize_counts()
    except Exception:
        pass
    rasterizer._states.clear()


@pytest.mark.parametrize("mode", ["verified", "deferred"])
@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("wide_depth", [False, True])
def test_second_frame_runs_without_readback_and_is_bit_identical(gpu_device, fresh_state, fused, wide_depth, mode):
    rz = fresh_state
    rz.set_sync_free(mode)
    model, cam, bg, target = _scene(gpu_device, wide_depth=wide_depth)
    pkg0, g0 = _step(model, cam, bg, target, fused)                 # first frame: two-call path, learns the capacity
    ctx0 = pkg0["render"].grad_fn
    assert ctx0.frame_pending is None
    R0, V0 = rz.frame_counts(pkg0["render"])
    assert R0 > V0 > 1000 and ctx0.layout == (R0, V0)
    pkg1, g1 = _step(model, cam, bg, target, fused)                 # second frame: gsr_forward, laid out for a capacity
    ctx1 = pkg1["render"].grad_fn
    assert ctx1.layout[0] >= int(1.5 * R0) and ctx1.layout[1] == model._xyz.shape[0]
    if mode == "deferred":
        pend = ctx1.frame_pending
        assert pend is not None and pend.done and pend.capacity == ctx1.layout[0]
    else:
        assert ctx1.frame_pending is None and ctx1.counts == (R0, V0)
    assert rz.frame_counts(pkg1["render"]) == (R0, V0)
    assert torch.equal(pkg0["render"], pkg1["render"]) and torch.equal(pkg0["radii"], pkg1["radii"])
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)
    # forward-only frames (no autograd node)
    with torch.no_grad():
        from mvs_gaussian_splatting_amd import render
        from mvs_gaussian_splatting_amd.synthetic import PipelineParams
        img = render(cam, model, PipelineParams(), bg)["render"]
    rz.synchronize_counts()
    assert torch.equal(img, pkg0["render"].detach())
    assert rz.last_counts(gpu_device, model._xyz.shape[0], cam.image_width, cam.image_height) == (R0, V0)
    assert rz.reissued_frames(gpu_device, model._xyz.shape[0], cam.image_width, cam.image_height) == 0


def test_raw_abi_lists_and_state_are_those_of_the_two_call_path(gpu_device):
    """gsr_forward at several capacities (exact, generous) leaves the same sorted lists, ranges, final_T and n_contrib
    as the two-call forward; the top-digit pass is exercised by the wide-depth cloud."""
    from gpu_util import forward_with_state, product_settings
    for wide in (False, True):
        model, cam, bg, _ = small_scene(P=5000, sh_degree=1, width=320, height=176, scale=0.04, seed=3)
        if wide:
            z = torch.exp(torch.rand(5000, generator=torch.Generator().manual_seed(2)) * math.log(4.0e6) + math.log(0.3))
            s = z / model._xyz[:, 2]
            model._xyz *= s[:, None]
            model._scaling += torch.log(s)[:, None]
        st = product_settings(cam, bg, 1, gpu_device)
        kw = dict(shs=model.get_features, scales=model.get_scaling, rotations=model.get_rotation)
        for mode in (0, 2):
            ref = forward_with_state(gpu_device, st, model.get_xyz, model.get_opacity, binning_mode=mode, **kw)
            for cap in (ref["R"], ref["R"] + 1, 3 * ref["R"]):
                out = forward_with_state(gpu_device, st, model.get_xyz, model.get_opacity, binning_mode=mode,
                                         sync_free_capacity=cap, **kw)
                assert (out["R"], out["V"]) == (ref["R"], ref["V"])
                assert np.array_equal(out["point_list"], ref["point_list"]) and np.array_equal(out["keys"], ref["keys"])
                assert np.array_equal(out["ranges"], ref["ranges"])
                for k in ("color", "final_T", "n_contrib", "radii"):
                    assert torch.equal(out[k], ref[k]), k
            # GsrParams.depth_span_lt24 (ABI v13): the frame goes without the depth sort's fourth pass.  A narrow frame
            # is the same frame; a wide one is reported through the two depth keys (the caller discards it) and touches
            # nothing out of bounds: same counts, the same Gaussians in every tile, only their order is not the depth order.
            out = forward_with_state(gpu_device, st, model.get_xyz, model.get_opacity, binning_mode=mode,
                                     sync_free_capacity=ref["R"] + 7, depth_span_lt24=True, **kw)
            lo, hi = out["depth_keys"]
            assert (out["R"], out["V"]) == (ref["R"], ref["V"]) and hi >= lo
            assert ((hi - lo) >> 24 != 0) == wide
            assert np.array_equal(out["ranges"], ref["ranges"]) and torch.equal(out["radii"], ref["radii"])
            if not wide:
                assert np.array_equal(out["point_list"], ref["point_list"]) and np.array_equal(out["keys"], ref["keys"])
                for k in ("color", "final_T", "n_contrib"):
                    assert torch.equal(out[k], ref[k]), k
            else:
                assert not np.array_equal(out["point_list"], ref["point_list"])
                tile_of = (ref["keys"] >> np.uint64(32)).astype(np.int64)
                a = np.lexsort((ref["point_list"], tile_of))
                b = np.lexsort((out["point_list"], tile_of))
                assert np.array_equal(ref["point_list"][a], out["point_list"][b])


def test_a_frame_that_spans_more_depth_than_the_frames_before_is_issued_again(gpu_device, fresh_state):
    """Verified mode issues frames without the depth sort's fourth pass while every frame seen stayed below 0.9 x 2^24
    depth-key steps.  The first frame that spans more is caught by the counts check and issued again before the operator
    returns (bit-identical to the per-frame read-back); from then on the pass is enqueued and nothing is re-issued."""
    rz = fresh_state
    narrow = _scene(gpu_device)
    wide = _scene(gpu_device, wide_depth=True)
    P, (W, H) = narrow[0]._xyz.shape[0], (narrow[1].image_width, narrow[1].image_height)
    prev = rz.set_sync_free(False)
    ref_n, gref_n = _step(*narrow)
    ref_w, gref_w = _step(*wide)
    rz.set_sync_free(prev)
    rz._states.clear()
    _step(*narrow)                                               # two-call path: learns capacity and span
    st = next(iter(rz._states.values()))
    assert 0 < st.depth_span < rz._DEPTH_SPAN_TRUSTED
    st.capacity = 1 << 22                                        # room for the wide cloud: only the span is at stake
    pkg, g = _step(*narrow)                                      # issued without the fourth pass
    assert rz.reissued_frames(gpu_device, P, W, H) == 0
    assert torch.equal(pkg["render"], ref_n["render"]) and all(torch.equal(a, b) for a, b in zip(g, gref_n))
    pkg, g = _step(*wide)                                        # spans 23 binades: caught, issued again
    assert rz.reissued_frames(gpu_device, P, W, H) == 1 and st.depth_span >> 24
    assert rz.frame_counts(pkg["render"]) == rz.frame_counts(ref_w["render"])
    assert torch.equal(pkg["render"], ref_w["render"]) and torch.equal(pkg["radii"], ref_w["radii"])
    assert all(torch.equal(a, b) for a, b in zip(g, gref_w))
    for scene, ref, gref in ((wide, ref_w, gref_w), (narrow, ref_n, gref_n), (wide, ref_w, gref_w)):
        pkg, g = _step(*scene)                                   # the pass is enqueued from now on: any frame is right
        assert torch.equal(pkg["render"], ref["render"]) and all(torch.equal(a, b) for a, b in zip(g, gref))
    assert rz.reissued_frames(gpu_device, P, W, H) == 1
    with torch.no_grad():                                        # forward-only frames take the same route
        from mvs_gaussian_splatting_amd import render
        from mvs_gaussian_splatting_amd.synthetic import PipelineParams
        rz._states.clear()
        a = render(narrow[1], narrow[0], PipelineParams(), narrow[2])["render"]
        next(iter(rz._states.values())).capacity = 1 << 22
        b = render(wide[1], wide[0], PipelineParams(), wide[2])["render"]
        c = render(wide[1], wide[0], PipelineParams(), wide[2])["render"]
    assert torch.equal(a, ref_n["render"].detach()) and torch.equal(b, ref_w["render"].detach()) and torch.equal(c, b)
    assert rz.reissued_frames(gpu_device, P, W, H) == 1


def _views(P, n=4, **kw):
    return [small_scene(P=P, sh_degree=2, width=352, height=208, scale=0.03, view=v, **kw)[1] for v in range(n)]


def test_overflow_is_recovered_before_the_operator_returns(gpu_device, fresh_state):
    """Default mode: a frame that does not fit the capacity it was issued with is issued again, inside the forward --
    the caller sees the complete image, the backward the complete state; nothing raises; everything is bit-identical to
    the per-frame read-back."""
    rz = fresh_state
    model, cam, bg, target = _scene(gpu_device)
    P = model._xyz.shape[0]
    prev = rz.set_sync_free(False)
    pkg_ref, g_ref = _step(model, cam, bg, target)
    rz.set_sync_free(prev)
    rz._states.clear()
    pkg0, _ = _step(model, cam, bg, target)
    R0, _ = rz.frame_counts(pkg0["render"])
    key_state = next(iter(rz._states.values()))
    for fused in (True, False):
        key_state.capacity = max(1024, R0 // 3)                  # far too small for the next frame
        before = rz.reissued_frames(gpu_device, P, cam.image_width, cam.image_height)
        pkg, g = _step(model, cam, bg, target, fused)            # overflows on the device, is re-issued, then runs its backward
        assert rz.reissued_frames(gpu_device, P, cam.image_width, cam.image_height) == before + 1
        assert rz.frame_counts(pkg["render"])[0] == R0
        assert torch.equal(pkg["render"], pkg_ref["render"]) and torch.equal(pkg["radii"], pkg_ref["radii"])
        if fused:
            for a, b in zip(g, g_ref):
                assert torch.equal(a, b)
        assert key_state.capacity >= int(1.5 * R0)               # ... and the capacity has grown for the frames after it
    # forward-only frames too (they share workspaces from frame to frame: the re-issue must not leave a stale set behind)
    from mvs_gaussian_splatting_amd import render
    from mvs_gaussian_splatting_amd.synthetic import PipelineParams
    with torch.no_grad():
        a = render(cam, model, PipelineParams(), bg)["render"]
        key_state.capacity = max(1024, R0 // 3)
        b = render(cam, model, PipelineParams(), bg)["render"]
        c = render(cam, model, PipelineParams(), bg)["render"]
    for img in (a, b, c):
        assert torch.equal(img, pkg_ref["render"].detach())


def test_any_camera_sequence_is_bit_identical_to_the_per_frame_readback(gpu_device, fresh_state):
    """train.py:81-92 draws a different camera every iteration: the capacity learnt on one view says nothing about the
    next.  Four orbit views, visited in an order that makes the instance count jump, with a capacity forced below every
    view's count half of the time: image, radii, every gradient and the densification statistics of every step equal the
    two-call path's bit for bit, and nothing raises."""
    from mvs_gaussian_splatting_amd import add_densification_stats
    rz = fresh_state
    model, _, bg, target = _scene(gpu_device, P=8000)
    cams = [c.to(gpu_device) for c in _views(8000)]
    order = [0, 2, 1, 3, 3, 0, 2, 2, 1]

    def run(mode, squeeze):
        rz.set_sync_free(mode)
        rz._states.clear()
        model.xyz_gradient_accum.zero_(); model.denom.zero_(); model.max_radii2D.zero_()
        out = []
        for i, v in enumerate(order):
            if squeeze and i % 2 == 1:
                for st in rz._states.values():
                    st.capacity = 2048
            pkg, g = _step(model, cams[v], bg, target)
            add_densification_stats(model, pkg["viewspace_points"], pkg["radii"])
            out.append((pkg["render"].detach().clone(), pkg["radii"].clone(), g, rz.frame_counts(pkg["render"])))
        stats = (model.xyz_gradient_accum.clone(), model.denom.clone(), model.max_radii2D.clone())
        return out, stats

    ref, ref_stats = run(False, False)
    counts = [r[3][0] for r in ref]
    assert len(set(counts)) >= 3 and min(counts) > 0, counts      # the views really differ
    for squeeze in (False, True):
        got, got_stats = run(True, squeeze)
        for (img, radii, g, cnt), (img0, radii0, g0, cnt0) in zip(got, ref):
            assert cnt == cnt0 and torch.equal(img, img0) and torch.equal(radii, radii0)
            for a, b in zip(g, g0):
                assert torch.equal(a, b)
        for a, b in zip(got_stats, ref_stats):
            assert torch.equal(a, b)
    P = model._xyz.shape[0]
    assert rz.reissued_frames(gpu_device, P, cams[0].image_width, cams[0].image_height) >= 4


def test_deferred_mode_reports_an_overflow_instead_of_rendering_it_silently(gpu_device, fresh_state):
    rz = fresh_state
    rz.set_sync_free("deferred")
    model, cam, bg, target = _scene(gpu_device)
    pkg0, _ = _step(model, cam, bg, target)
    R0, _ = rz.frame_counts(pkg0["render"])
    key_state = next(iter(rz._states.values()))
    key_state.capacity = max(1024, R0 // 3)                      # far too small for the next frame
    from mvs_gaussian_splatting_amd import _lib, render, l1_loss
    from mvs_gaussian_splatting_amd.synthetic import PipelineParams
    pkg = render(cam, model, PipelineParams(), bg)               # issued without a read-back: overflows on the device
    loss = l1_loss(pkg["render"], target)
    with pytest.raises(_lib.GsrError, match="overflowed its binning capacity"):
        loss.backward()                                          # the frame's own backward refuses to run
    with pytest.raises(_lib.GsrError, match="overflowed"):
        rz.frame_counts(pkg["render"])
    assert key_state.capacity >= int(1.5 * R0)                   # ... and the capacity has grown for the frames after it
    pkg2, g2 = _step(model, cam, bg, target)                     # which are complete again
    assert torch.equal(pkg2["render"], pkg0["render"])
    # a forward-only frame that overflowed is reported by the next call into the operator / by synchronize_counts
    key_state.capacity = max(1024, R0 // 3)
    with torch.no_grad():
        render(cam, model, PipelineParams(), bg)
    with pytest.raises(_lib.GsrError, match="overflowed"):
        rz.synchronize_counts()
    rz.synchronize_counts()                                      # reported once per drain; nothing left pending


def test_fused_densification_statistics_equal_the_stand_alone_kernel(gpu_device, fresh_state):
    from mvs_gaussian_splatting_amd import add_densification_stats
    for fused_inputs in (True, False):
        model, cam, bg, target = _scene(gpu_device, P=4000, seed=5)
        _, cam2, _, _ = small_scene(P=4000, sh_degree=2, width=352, height=208, scale=0.03, view=2)
        cam2.to(gpu_device)
        g = torch.Generator().manual_seed(3)
        base = (torch.rand(4000, 1, generator=g).to(gpu_device), torch.randint(0, 5, (4000, 1), generator=g).float().to(gpu_device),
                torch.randint(0, 6, (4000,), generator=g).float().to(gpu_device))
        results = []
        for stats_in_backward in (False, True):
            model.xyz_gradient_accum, model.denom, model.max_radii2D = (t.clone() for t in base)
            for c in (cam, cam2):
                pkg, _ = _step(model, c, bg, target, fused=fused_inputs, stats=stats_in_backward)
                assert bool(getattr(pkg["viewspace_points"], "_gsr_stats_fused", False)) == stats_in_backward
                add_densification_stats(model, pkg["viewspace_points"], pkg["radii"])     # no-op for a fused frame
            results.append((model.xyz_gradient_accum.clone(), model.denom.clone(), model.max_radii2D.clone()))
        for a, b in zip(*results):
            assert torch.equal(a, b)
        assert float(results[0][1].max()) >= base[1].max() + 1


def test_markers_switch(gpu_device, fresh_state):
    from mvs_gaussian_splatting_amd import _lib
    _lib.enable_markers(True)        # librocprofiler-sdk-roctx.so ships with the ROCm image
    try:
        model, cam, bg, target = _scene(gpu_device, P=2000)
        pkg, _ = _step(model, cam, bg, target)
        assert torch.isfinite(pkg["render"]).all()
    finally:
        _lib.enable_markers(False)


def test_full_size_C3_frame_without_readback_equals_the_two_call_frame(gpu_device):
    """BASELINE config 3 (1 M Gaussians, 1080p) through the raw C ABI: gsr_forward at 1.5 x capacity leaves the same image,
    per-pixel state, lists and ranges as the two-call forward -- the size-independent property the small scenes above
    check, at a size where every launch is really sized for a capacity beyond the count (blocks past the count exit)."""
    from gpu_util import forward_with_state, product_settings
    from mvs_gaussian_splatting_amd.synthetic import CONFIGS, make_scene
    model, cam, bg, _ = make_scene(CONFIGS["C3"])
    st = product_settings(cam, bg, 3, gpu_device)
    kw = dict(shs=model.get_features, scales=model.get_scaling, rotations=model.get_rotation, binning_mode=2)
    ref = forward_with_state(gpu_device, st, model.get_xyz, model.get_opacity, **kw)
    out = forward_with_state(gpu_device, st, model.get_xyz, model.get_opacity, sync_free_capacity=int(1.5 * ref["R"]), **kw)
    assert (out["R"], out["V"]) == (ref["R"], ref["V"]) and ref["R"] > 2_000_000
    for k in ("color", "final_T", "n_contrib", "radii"):
        assert torch.equal(out[k], ref[k]), k
    assert np.array_equal(out["point_list"], ref["point_list"]) and np.array_equal(out["ranges"], ref["ranges"])


# ---- a frame WITHOUT a single instance, issued into the state an ordinary frame left behind ------------
def _away_camera(cam, dev):
    """The camera of ``cam`` turned by 180 degrees about the vertical axis: every Gaussian of the cloud lies behind it."""
    from mvs_gaussian_splatting_amd.synthetic import SyntheticCamera
    f = lambda fov, px: px / (2.0 * math.tan(fov * 0.5))  # noqa: E731
    return SyntheticCamera(cam.image_width, cam.image_height, f(cam.FoVx, cam.image_width), f(cam.FoVy, cam.image_height),
                           R=np.diag([-1.0, 1.0, -1.0]), T=np.zeros(3), device=dev)


# 352 x 208 = 286 tiles: the tile sort is one pass (ranges = scan of its digit totals); 640 x 400 = 1000 tiles: two passes,
# the last one segmented by the first one's totals
@pytest.mark.parametrize("size", [(352, 208), (640, 400)])
@pytest.mark.parametrize("mode", ["verified", "deferred"])
def test_a_frame_without_instances_after_an_ordinary_one_is_the_background(gpu_device, fresh_state, size, mode):
    """The capacity (and every workspace word) is left by an ordinary view; the next view sees nothing.  The sort's row
    scan has no item to count -- its totals, the segment table of the second pass and the tile ranges must say so instead
    of repeating the frame before: image == background, every range (0, 0), every gradient zero."""
    rz = fresh_state
    rz.set_sync_free(mode)
    W, H = size
    model, cam, _, target = small_scene(P=6000, sh_degree=2, width=W, height=H, scale=0.03)
    model.to(gpu_device); cam.to(gpu_device)
    target = target.to(gpu_device)
    for p in model.parameters():
        p.requires_grad_(True)
    bg = torch.tensor([0.25, 0.5, 0.75], device=gpu_device)
    away = _away_camera(cam, gpu_device)
    pkg0, _ = _step(model, cam, bg, target)                       # learns the capacity
    assert rz.frame_counts(pkg0["render"])[0] > 1000
    _step(model, cam, bg, target)                                 # gsr_forward: leaves its totals in the workspaces
    for rep in range(2):
        pkg, g = _step(model, away, bg, target)
        assert rz.frame_counts(pkg["render"]) == (0, 0)
        assert torch.equal(pkg["render"], bg[:, None, None].expand(3, H, W))
        assert not pkg["visibility_filter"].any() and not pkg["radii"].any()
        for t in g:
            assert not t.any()
    pkg1, _ = _step(model, cam, bg, target)                       # and the ordinary view is itself again
    assert torch.equal(pkg1["render"], pkg0["render"])
    with torch.no_grad():                                         # forward-only frames share their workspaces outright
        from mvs_gaussian_splatting_amd import render
        from mvs_gaussian_splatting_amd.synthetic import PipelineParams
        a = render(cam, model, PipelineParams(), bg)["render"]
        b = render(away, model, PipelineParams(), bg)["render"]
        c = render(cam, model, PipelineParams(), bg)["render"]
    rz.synchronize_counts()
    assert torch.equal(a, pkg0["render"].detach()) and torch.equal(c, a)
    assert torch.equal(b, bg[:, None, None].expand(3, H, W))


@pytest.mark.parametrize("size", [(352, 208), (640, 400)])
def test_raw_gsr_forward_without_instances_into_used_workspaces(gpu_device, size):
    """The same through the raw C ABI: two ``gsr_forward`` calls into ONE set of workspaces, the second with every
    Gaussian behind the camera: ranges all (0, 0), image == background, counts 0."""
    from gpu_util import product_settings
    from mvs_gaussian_splatting_amd import _lib
    from mvs_gaussian_splatting_amd.rasterizer import _make_params
    lib = _lib.load()
    dev = gpu_device
    W, H = size
    model, cam, _, _ = small_scene(P=6000, sh_degree=1, width=W, height=H, scale=0.03)
    model.to(dev)
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    P = 6000
    e = torch.empty(0, device=dev)
    cap = 1 << 18
    with torch.cuda.device(dev), torch.no_grad():
        stream = torch.cuda.current_stream(dev).cuda_stream
        geom = torch.empty(lib.gsr_geom_bytes(P), dtype=torch.uint8, device=dev)
        img = torch.empty(lib.gsr_image_bytes(W, H), dtype=torch.uint8, device=dev)
        nb = lib.gsr_binning_bytes(cap, P, W, H, _lib.BINNING_TWO_LEVEL_CULLED)
        binning = torch.empty(nb, dtype=torch.uint8, device=dev)
        radii = torch.zeros(P, dtype=torch.int32, device=dev)
        color = torch.empty(3, H, W, device=dev)
        pinned = torch.zeros(16, dtype=torch.int32).pin_memory()
        gx, gy = (W + 15) // 16, (H + 15) // 16
        results = []
        for c in (cam, _away_camera(cam, dev), cam, _away_camera(cam, dev)):
            st = product_settings(c, bg, 1, dev)
            params, keep = _make_params(dev, st, model.get_xyz.contiguous(), model.get_features.contiguous(), e,
                                        model.get_opacity.contiguous(), model.get_scaling.contiguous(),
                                        model.get_rotation.contiguous(), e)
            params.binning_mode = _lib.BINNING_TWO_LEVEL_CULLED
            params.counts_pinned = pinned.data_ptr()
            _lib.check(lib.gsr_forward(C.byref(params), geom.data_ptr(), binning.data_ptr(), nb, cap, img.data_ptr(),
                                       radii.data_ptr(), color.data_ptr(), None, stream), "gsr_forward")
            final_T = torch.empty(H, W, device=dev)
            n_contrib = torch.empty(H, W, dtype=torch.int32, device=dev)
            ranges = torch.empty(gx * gy, 2, dtype=torch.int32, device=dev)
            _lib.check(lib.gsr_debug_read_image(img.data_ptr(), W, H, final_T.data_ptr(), n_contrib.data_ptr(),
                                                ranges.data_ptr(), stream), "read_img")
            torch.cuda.synchronize(dev)
            results.append((int(pinned[0]), int(pinned[1]), color.clone(), ranges.clone(), final_T.clone(), n_contrib.clone()))
            del keep
    assert results[0][0] > 1000 and results[0][0] <= cap
    for k in (1, 3):
        R, V, col, ranges, final_T, n_contrib = results[k]
        assert (R, V) == (0, 0)
        assert not ranges.any()
        assert torch.equal(col, bg[:, None, None].expand(3, H, W))
        assert torch.equal(final_T, torch.ones_like(final_T)) and not n_contrib.any()
    assert torch.equal(results[2][2], results[0][2]) and torch.equal(results[2][3], results[0][3])


@pytest.mark.parametrize("size", [(352, 208), (640, 400)])
def test_graphed_renderer_frame_without_instances(gpu_device, size):
    from mvs_gaussian_splatting_amd import render
    from mvs_gaussian_splatting_amd.graphed import GraphedRenderer
    from mvs_gaussian_splatting_amd.synthetic import PipelineParams
    dev = gpu_device
    W, H = size
    model, cam, _, _ = small_scene(P=6000, sh_degree=2, width=W, height=H, scale=0.03)
    model.to(dev); cam.to(dev)
    bg = torch.tensor([0.3, 0.1, 0.6], device=dev)
    away = _away_camera(cam, dev)
    gr = GraphedRenderer(model, PipelineParams(), bg)
    with torch.no_grad():
        want = render(cam, model, PipelineParams(), bg)["render"]
        for c, expect in ((cam, want), (cam, want), (away, None), (away, None), (cam, want)):
            got = gr.render(c)
            if expect is None:
                assert torch.equal(got["render"], bg[:, None, None].expand(3, H, W))
                assert not got["radii"].any() and not got["visibility_filter"].any()
            else:
                assert torch.equal(got["render"], expect)
    gr.check()
