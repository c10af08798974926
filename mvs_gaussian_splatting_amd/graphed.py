"""Forward-only rendering of a FIXED model through one captured HIP graph per camera format.

``render.py`` of the reference (``render.py:32-40``) renders a trained model from a list of cameras: the parameters do
not change, only the view does.  At 100 k Gaussians (BASELINE config 2) the ~27 kernels of a frame take 0.19 ms on
the GPU but ~0.27 ms to ISSUE from Python one by one.  ``GraphedRenderer`` issues them once: the whole frame
(``gsr_forward``: no host round-trip, every launch sized for a capacity, counts read on the device) is captured into a
HIP graph (``torch.cuda.CUDAGraph``) whose inputs live in static buffers; a frame is then three small copies (view
matrix, projection matrix, camera centre) and one graph launch.

    gr = GraphedRenderer(gaussians, pipe, background)
    for cam in cameras:
        img = gr.render(cam)["render"]          # valid until the next render() -- clone() to keep it

The graph fixes everything passed by value: image size, tan(FoV/2), SH degree, scale modifier.  One graph is kept per
such format (cameras of one dataset share it).  Training cannot use this class: the backward needs each frame's
workspaces, which the next replay overwrites.

Capacity: the first frame of a format runs eagerly (with the count read-back) and sizes the binning workspace at 1.5 x
its instance count.  Every later frame's real count lands in pinned memory and is compared with the capacity before
``render()`` returns (``verify=True``, the default: the host waits for that frame); an overflowed frame is re-rendered
with a larger workspace (nothing is lost: the inputs are still there), so the caller always receives a complete image.
``verify=False`` is for callers that take the verdict themselves (``MultiStreamRenderer`` does, per frame, before it
hands the frame over): the verdict then arrives with the next call for the same format or in ``check()``.

The model is FIXED: the operator inputs are snapshotted (cloned) at construction in both input modes; later in-place
updates of the model's parameters are not seen.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional

import torch

from . import _lib
from .rasterizer import GaussianRasterizationSettings, _make_params, _round_ws
from .renderer import _can_fuse


class _Format:
    """One captured graph: static inputs / outputs / workspaces for an (H, W, tanfovx, tanfovy, degree) combination."""

    def __init__(self):
        self.graph = None
        self.capacity = 0
        self.pending = False        # a replay whose count has not been compared with the capacity yet
        self.frames = 0


class GraphedRenderer:
    def __init__(self, pc, pipe, bg_color: torch.Tensor, scaling_modifier: float = 1.0, share_inputs_with=None):
        """``share_inputs_with``: another GraphedRenderer of the same model whose input snapshot this one reads too
        (the lanes of a MultiStreamRenderer: one copy of the parameters, not one per stream)."""
        if not pc.get_xyz.is_cuda:
            raise _lib.GsrError("GraphedRenderer needs the model on a ROCm GPU (no CPU path)")
        if getattr(pipe, "debug", False):
            raise ValueError("debug mode synchronises after every kernel: not capturable")
        self.lib = _lib.load()
        self.dev = pc.get_xyz.device
        self.scaling_modifier = float(scaling_modifier)
        self.sh_degree = int(pc.active_sh_degree)
        self.bg = bg_color.detach().to(self.dev, torch.float32).contiguous().clone()
        empty = torch.empty(0, dtype=torch.float32, device=self.dev)
        with torch.no_grad():
            # the operator inputs, taken ONCE (the model is fixed) as private copies: raw parameters where the kernels
            # can apply the activations themselves, the getters' results otherwise
            self.P = int(pc.get_xyz.shape[0])
            snap = lambda t: t.detach().contiguous().clone()  # noqa: E731
            if share_inputs_with is not None:
                o = share_inputs_with
                self.fused, self.inputs, self.act_flags = o.fused, o.inputs, o.act_flags
            elif _can_fuse(pc, pipe, None):
                self.fused = True
                self.inputs = dict(means3D=snap(pc.get_xyz), sh=snap(pc._features_dc), colors_precomp=empty,
                                   opacities=snap(pc._opacity), scales=snap(pc._scaling), rotations=snap(pc._rotation),
                                   cov3Ds_precomp=empty,
                                   sh_rest=snap(pc._features_rest) if pc._features_rest.shape[1] else None)
                self.act_flags = _lib.ACT_SCALE_EXP | _lib.ACT_ROT_NORMALIZE | _lib.ACT_OPACITY_SIGMOID
            else:
                self.fused = False
                self.inputs = dict(means3D=snap(pc.get_xyz), sh=snap(pc.get_features), colors_precomp=empty,
                                   opacities=snap(pc.get_opacity), scales=snap(pc.get_scaling),
                                   rotations=snap(pc.get_rotation), cov3Ds_precomp=empty, sh_rest=None)
                self.act_flags = 0
        self.formats: Dict[tuple, _Format] = {}

    # ---- one format ----------------------------------------------------------------------------------------------
    def _build(self, key, cam) -> _Format:
        H, W, tanx, tany = key
        lib, dev, P = self.lib, self.dev, self.P
        f = _Format()
        f.view = torch.empty(4, 4, dtype=torch.float32, device=dev)
        f.proj = torch.empty(4, 4, dtype=torch.float32, device=dev)
        f.campos = torch.empty(3, dtype=torch.float32, device=dev)
        f.settings = GaussianRasterizationSettings(H, W, tanx, tany, self.bg, self.scaling_modifier, f.view, f.proj,
                                                   self.sh_degree, f.campos, False, False)
        i = self.inputs
        f.params, f.keep = _make_params(dev, f.settings, i["means3D"], i["sh"], i["colors_precomp"], i["opacities"],
                                        i["scales"], i["rotations"], i["cov3Ds_precomp"], sh_rest=i["sh_rest"],
                                        act_flags=self.act_flags, forward_only=True)
        f.params.profile = None
        f.pinned = torch.zeros(16, dtype=torch.int32).pin_memory()
        f.params.counts_pinned = f.pinned.data_ptr()
        f.geom = torch.empty(lib.gsr_geom_bytes(P), dtype=torch.uint8, device=dev)
        f.img = torch.empty(lib.gsr_image_bytes(W, H), dtype=torch.uint8, device=dev)
        f.radii = torch.zeros(P, dtype=torch.int32, device=dev)
        f.visible = torch.zeros(P, dtype=torch.bool, device=dev)      # static: no per-frame allocation on a side stream
        f.color = torch.empty(3, H, W, dtype=torch.float32, device=dev)
        f.binning = None
        self._set_camera(f, cam)
        # first frame: eager, with the count read-back -> capacity
        stream = torch.cuda.current_stream(dev).cuda_stream
        R, V = C.c_uint32(0), C.c_uint32(0)
        _lib.check(lib.gsr_forward_preprocess(C.byref(f.params), f.geom.data_ptr(), f.radii.data_ptr(), stream,
                                              C.byref(R), C.byref(V)), "gsr_forward_preprocess")
        self._capture(f, max(int(R.value), 1))
        return f

    def _capture(self, f: _Format, need: int) -> None:
        lib, dev = self.lib, self.dev
        H, W = f.settings.image_height, f.settings.image_width
        f.capacity = (int(need * 1.5) + (1 << 20)) >> 20 << 20
        nbytes = lib.gsr_binning_bytes(f.capacity, self.P, W, H, f.params.binning_mode)
        f.binning = torch.empty(_round_ws(nbytes), dtype=torch.uint8, device=dev)
        f.nbytes = nbytes
        if f.params.binning_mode == _lib.BINNING_KEYS64:
            raise _lib.GsrError("GraphedRenderer needs a two-level binning mode (gsr_forward)")
        torch.cuda.synchronize(dev)
        f.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(f.graph):
            stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(lib.gsr_forward(C.byref(f.params), f.geom.data_ptr(), f.binning.data_ptr(), f.nbytes, f.capacity,
                                       f.img.data_ptr(), f.radii.data_ptr(), f.color.data_ptr(), None, stream),
                       "gsr_forward (graph capture)")
            torch.gt(f.radii, 0, out=f.visible)
        f.done = torch.cuda.Event()
        f.pending = False

    def _set_camera(self, f: _Format, cam) -> None:
        f.view.copy_(cam.world_view_transform, non_blocking=True)
        f.proj.copy_(cam.full_proj_transform, non_blocking=True)
        f.campos.copy_(cam.camera_center, non_blocking=True)

    def _verdict(self, f: _Format) -> bool:
        """Wait for the pending replay and compare its instance count with the capacity.  True: the frame is complete."""
        if not f.pending:
            return True
        f.done.synchronize()
        f.pending = False
        f.last_counts = (int(f.pinned[0]) & 0xffffffff, int(f.pinned[1]) & 0xffffffff)
        return f.last_counts[0] <= f.capacity

    # ---- public ----------------------------------------------------------------------------------------------------
    def _key(self, cam) -> tuple:
        return (int(cam.image_height), int(cam.image_width), math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5))

    def _result(self, f: _Format) -> dict:
        return {"render": f.color, "viewspace_points": None, "visibility_filter": f.visible, "radii": f.radii,
                "selected_pts_mask": None}

    def _replay(self, f: _Format) -> None:
        f.graph.replay()
        f.done.record()
        f.pending = True
        f.frames += 1

    def complete(self, viewpoint_camera) -> bool:
        """Take the verdict of the last frame issued for this camera's format (host waits for that frame) and, if it
        overflowed its capacity, render it again with a larger workspace on the current stream.  Returns True when the
        frame had to be re-rendered.  After this call the format's buffers hold a complete frame."""
        with torch.cuda.device(self.dev):
            f = self.formats[self._key(viewpoint_camera)]
            if self._verdict(f):
                return False
            self._capture(f, f.last_counts[0])
            self._set_camera(f, viewpoint_camera)
            self._replay(f)
            if not self._verdict(f):      # cannot happen: the capacity now exceeds this very frame's count
                raise _lib.GsrError("graphed frame overflowed twice")
            return True

    def render(self, viewpoint_camera, verify: bool = True) -> dict:
        """One frame.  The tensors of the result are STATIC buffers, overwritten by the next call for the same camera
        format.  ``verify=True`` (default) waits for the frame and re-renders it if its instance count exceeded the
        capacity; with ``verify=False`` the caller must call ``complete(camera)`` before using the frame."""
        key = self._key(viewpoint_camera)
        with torch.cuda.device(self.dev):
            f = self.formats.get(key)
            if f is None:
                f = self.formats[key] = self._build(key, viewpoint_camera)
            elif not self._verdict(f):
                need = f.last_counts[0]
                self._capture(f, need)
                raise _lib.GsrError(f"the previous graphed frame overflowed its binning capacity ({need} instances) and "
                                    "nobody took its verdict (render(verify=False) without complete()): its image was "
                                    "incomplete; the graph has been rebuilt with a larger workspace")
            self._set_camera(f, viewpoint_camera)
            self._replay(f)
        if verify:
            self.complete(viewpoint_camera)
        return self._result(f)

    def check(self) -> None:
        """Block until every issued frame has been checked; raises GsrError if the last frame of a format overflowed."""
        with torch.cuda.device(self.dev):
            for f in self.formats.values():
                if not self._verdict(f):
                    need = f.last_counts[0]
                    self._capture(f, need)
                    raise _lib.GsrError(f"a graphed frame overflowed its binning capacity ({need} instances); graph rebuilt")


class MultiStreamRenderer:
    """Forward-only rendering of a fixed model with several frames in flight, one per HIP stream.

    A frame is bound by HBM bandwidth for its first two thirds (preprocess, binning) and by vector instruction issue for
    the last (compositing), and the ~27 launches of a small frame leave the GPU waiting: frames on different streams fill
    each other's gaps.  With three streams 1.2x the frames per second at 6 M Gaussians / 1080p, 1.6x at 1 M, 2.4x at
    100 k / 800x800 (``profiles/r03/multi_stream.txt``; two streams gain little: frames issued together run in lockstep,
    HBM-bound phase against HBM-bound phase, and stream priorities meant to stagger them cost more than they give).  Frames are
    independent -- the model does not change -- which is the situation of the reference's ``render.py:32-40`` (a trained
    model, a list of cameras).  Every stream has its own ``GraphedRenderer`` (workspaces, static buffers, captured
    graph); the parameters are shared, read-only.

        mr = MultiStreamRenderer(gaussians, pipe, background, streams=3)
        for i, out in mr.render_views(cameras):        # in order; `out` is valid until `streams` more frames were issued
            save(out["render"])                        # (work issued on the current stream sees the finished frame)
        mr.check()

    ``render_views`` keeps ``streams`` frames in flight and hands each one over once its instance count has been
    checked against the lane's capacity (re-rendered first if it did not fit) and the CURRENT stream has been made to
    wait for it; whatever the caller enqueues on the current stream afterwards (a copy to the host, an encoder) runs
    after the frame and before the lane's buffers are reused.
    """

    def __init__(self, pc, pipe, bg_color: torch.Tensor, scaling_modifier: float = 1.0, streams: int = 3):
        if streams < 1:
            raise ValueError("streams must be >= 1")
        self.dev = pc.get_xyz.device
        if not pc.get_xyz.is_cuda:
            raise _lib.GsrError("MultiStreamRenderer needs the model on a ROCm GPU (no CPU path)")
        self.streams = [torch.cuda.Stream(self.dev) for _ in range(streams)]
        self.lanes = [GraphedRenderer(pc, pipe, bg_color, scaling_modifier)]
        self.lanes += [GraphedRenderer(pc, pipe, bg_color, scaling_modifier, share_inputs_with=self.lanes[0])
                       for _ in range(streams - 1)]
        self.done = [torch.cuda.Event() for _ in range(streams)]
        self.released = [None] * streams      # event on the consumer's stream after which a lane's buffers may be reused
        self.issued = 0

    def _issue(self, cam, inputs_ready):
        k = self.issued % len(self.lanes)
        s = self.streams[k]
        # the cameras (and, the first time, the parameters) may have been produced on the caller's stream: wait for the
        # point at which render_views() was called, not for the stream's tail -- the tail holds the hand-overs of the
        # frames before this one
        s.wait_event(inputs_ready)
        if self.released[k] is not None:
            s.wait_event(self.released[k])     # the consumer of this lane's previous frame has read it
        with torch.cuda.stream(s):
            out = self.lanes[k].render(cam, verify=False)
            self.done[k].record(s)
        self.issued += 1
        return k, out, cam

    def render_views(self, cameras):
        """Generator of (index, result) in camera order with ``len(self.streams)`` frames in flight."""
        cameras = list(cameras)
        n = len(self.lanes)
        inflight = []
        nxt = 0
        with torch.cuda.device(self.dev):
            inputs_ready = torch.cuda.Event()
            inputs_ready.record(torch.cuda.current_stream(self.dev))
            for i in range(len(cameras)):
                while nxt < len(cameras) and len(inflight) < n:
                    inflight.append(self._issue(cameras[nxt], inputs_ready))
                    nxt += 1
                k, out, cam = inflight.pop(0)
                # the frame's verdict BEFORE it is handed over (a host wait on this frame only; the later frames keep
                # the GPU busy): a frame that overflowed its capacity is rendered again on its lane first
                with torch.cuda.stream(self.streams[k]):
                    if self.lanes[k].complete(cam):
                        self.done[k].record(self.streams[k])
                cur = torch.cuda.current_stream(self.dev)
                cur.wait_event(self.done[k])
                yield i, out
                ev = torch.cuda.Event()
                ev.record(cur)                 # whatever the consumer enqueued on the current stream for this frame
                self.released[k] = ev

    def check(self) -> None:
        """Block until every issued frame has completed and been checked against its capacity (see GraphedRenderer)."""
        for s in self.streams:
            s.synchronize()
        for lane in self.lanes:
            lane.check()
