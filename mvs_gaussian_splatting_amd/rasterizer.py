"""Drop-in operator surface for the reference's ``diff_gaussian_rasterization`` import
(``gaussian_renderer/__init__.py:15``): ``GaussianRasterizationSettings`` (the 12 fields built at
``gaussian_renderer/__init__.py:42-55``) and ``GaussianRasterizer`` (constructed at ``:57``, called
with keyword arguments at ``:257-265``).  The arithmetic runs in ``libgsr_hip.so`` through the C ABI
of ``include/gsr.h``; PyTorch only owns the memory and the stream.

No CPU path exists: tensors must live on a ROCm device and the HIP library must be built.

Host synchronisation.  Upstream reads ``num_rendered`` back in every forward to size its sort buffers: the GPU drains
while the host round-trips, sizes the binning workspace and issues the second half of the frame.  Here only the first
frame of a (device, P, W, H) combination does that (``gsr_forward_preprocess`` + ``gsr_forward_render``).  Later frames
are ENQUEUED WHOLE by ``gsr_forward`` with a caller-side capacity (1.5 x the largest instance count seen); the host
then waits for the event behind the scan kernel only (a third of the way into the frame: the GPU keeps the rest of the
frame queued and never idles), compares the real count with the capacity and, if the frame did not fit, re-issues it on
the two-call path with a workspace of the right size BEFORE the operator returns.  The operator therefore never raises
and never hands out an incomplete image, whatever the camera sequence (``train.py:81-107`` draws a random camera per
iteration, ``render.py:32-35`` saves every image at once) -- it is bit-identical to the per-frame read-back.

``GSR_SYNC_FREE`` (read at import) / ``set_sync_free``:  ``1`` / ``True`` (default) the verified mode above;
``0`` / ``False`` every frame on the two-call path;  ``deferred`` the count is compared with the capacity at the
latest when the frame's backward starts, when the next frame is issued or in ``synchronize_counts()`` -- an overflowed
frame raises ``GsrError`` THERE (its image and gradients are incomplete).  Only for callers that can redo a frame.
"""
from __future__ import annotations

import collections
import ctypes as C
import os
import threading
from typing import NamedTuple, Optional

import torch
import torch.nn as nn

from . import _lib


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None or t.numel() == 0 else t.data_ptr()


def _f32c(t: torch.Tensor, name: str, dev: torch.device, align16: bool = False) -> torch.Tensor:
    if t.device != dev:
        raise ValueError(f"{name} is on {t.device}, expected {dev}")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    t = t.contiguous()
    if align16 and t.data_ptr() % 16 != 0:
        t = t.clone()
    return t


def _check_rows(t: torch.Tensor, name: str, P: int, *tail) -> None:
    """The kernels index raw pointers: a non-empty input must be [P, *tail] (tail entry None = any size >= 1)."""
    if t.numel() == 0:
        return
    ok = t.dim() == 1 + len(tail) and t.shape[0] == P and all(
        (d is None and int(s) >= 1) or (d is not None and int(s) == d) for s, d in zip(t.shape[1:], tail))
    if not ok:
        want = ", ".join("M" if d is None else str(d) for d in tail)
        raise ValueError(f"{name} must have shape [P={P}, {want}], got {list(t.shape)}")


def _require_gpu(t: torch.Tensor) -> torch.device:
    if not t.is_cuda:
        raise _lib.GsrError("GaussianRasterizer needs tensors on a ROCm GPU: this build has no CPU path "
                            f"(means3D is on {t.device})")
    return t.device


# ---- process-wide switches: read from the environment ONCE, at import; changed afterwards only through the setters ----
_BINNING = {"keys64": _lib.BINNING_KEYS64, "two_level": _lib.BINNING_TWO_LEVEL, "culled": _lib.BINNING_TWO_LEVEL_CULLED}


def _binning_from_name(name: str) -> int:
    name = name.lower()
    if name not in _BINNING:
        raise ValueError(f"binning mode must be one of {sorted(_BINNING)}, got {name!r}")
    return _BINNING[name]


_binning_mode_value = _binning_from_name(os.environ.get("GSR_BINNING", "culled"))
_debug_flags_value = 0
SYNC_OFF, SYNC_VERIFIED, SYNC_DEFERRED = 0, 1, 2


def _sync_mode_from(value) -> int:
    if isinstance(value, str):
        v = value.strip().lower()
        if v in ("0", "off", "false", "no"):
            return SYNC_OFF
        if v in ("1", "on", "true", "yes", "verified", ""):
            return SYNC_VERIFIED
        if v in ("2", "deferred"):
            return SYNC_DEFERRED
        raise ValueError(f"sync-free mode must be 0 / 1 / deferred, got {value!r}")
    if value is True or value is False:
        return SYNC_VERIFIED if value else SYNC_OFF
    if value in (SYNC_OFF, SYNC_VERIFIED, SYNC_DEFERRED):
        return int(value)
    raise ValueError(f"sync-free mode must be a bool, 0 / 1 / 2 or a name, got {value!r}")


_sync_free_value = _sync_mode_from(os.environ.get("GSR_SYNC_FREE", "1"))


def set_binning_mode(name: str) -> str:
    """culled (default: two_level minus the instances whose tile the alpha >= 1/255 ellipse cannot reach; colour, radii
    and gradients are bit-identical to the other modes) | two_level (upstream's lists via two 32-bit sorts) | keys64
    (upstream's 64-bit (tile, depth) key sort).  Initial value: GSR_BINNING at import.  Returns the previous name."""
    global _binning_mode_value
    prev = next(k for k, v in _BINNING.items() if v == _binning_mode_value)
    _binning_mode_value = _binning_from_name(name)
    return prev


def set_debug_flags(flags: int) -> int:
    """GsrParams.debug_flags for the calls that follow (``_lib.DEBUG_*``); returns the previous value."""
    global _debug_flags_value
    prev, _debug_flags_value = _debug_flags_value, int(flags)
    return prev


def set_sync_free(mode) -> int:
    """How frames after the first of a shape are issued (module docstring): False / 0 = two calls with the count
    read-back between them; True / 1 = enqueued whole, verified before the operator returns (default);
    "deferred" / 2 = enqueued whole, verified later (an overflowed frame raises GsrError).  Returns the previous
    setting (SYNC_OFF / SYNC_VERIFIED / SYNC_DEFERRED; pass it back to restore)."""
    global _sync_free_value
    prev, _sync_free_value = _sync_free_value, _sync_mode_from(mode)
    return prev


def sync_free_mode() -> int:
    return _sync_free_value


def _stream(dev: torch.device) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


def _make_params(dev, settings: GaussianRasterizationSettings, means3D, sh, colors_precomp, opacities, scales,
                 rotations, cov3Ds_precomp, sh_rest=None, act_flags: int = 0, forward_only: bool = False):
    """Returns (GsrParams, keepalive list).  ``counts_pinned`` is left NULL: the forward paths below attach theirs."""
    bg = _f32c(settings.bg, "bg", dev)
    view = _f32c(settings.viewmatrix, "viewmatrix", dev)
    proj = _f32c(settings.projmatrix, "projmatrix", dev)
    campos = _f32c(settings.campos, "campos", dev)
    if bg.numel() != 3 or view.numel() != 16 or proj.numel() != 16 or campos.numel() != 3:
        raise ValueError("bg/campos must have 3 elements and viewmatrix/projmatrix 16")
    P = int(means3D.shape[0])
    M = int(sh.shape[1]) if sh.numel() else 0
    if sh_rest is not None:
        M = 1 + int(sh_rest.shape[1])
    p = _lib.GsrParams()
    p.P, p.M, p.D = P, M, int(settings.sh_degree)
    p.width, p.height = int(settings.image_width), int(settings.image_height)
    p.tan_fovx, p.tan_fovy = float(settings.tanfovx), float(settings.tanfovy)
    p.scale_modifier = float(settings.scale_modifier)
    p.prefiltered, p.debug = int(bool(settings.prefiltered)), int(bool(settings.debug))
    p.means3D, p.shs, p.colors_precomp = _ptr(means3D), _ptr(sh), _ptr(colors_precomp)
    p.opacities, p.scales, p.rotations = _ptr(opacities), _ptr(scales), _ptr(rotations)
    p.cov3D_precomp = _ptr(cov3Ds_precomp)
    p.viewmatrix, p.projmatrix, p.campos, p.bg = view.data_ptr(), proj.data_ptr(), campos.data_ptr(), bg.data_ptr()
    p.profile = _lib.active_profile_handle()      # raw handle; the owning object travels in ctx.profile
    p.shs_rest = _ptr(sh_rest)
    p.act_flags = int(act_flags)
    p.binning_mode = _binning_mode_value
    p.counts_pinned = None
    p.forward_only = int(bool(forward_only))
    p.debug_flags = _debug_flags_value
    p.visible_out = None
    p.depth_span_lt24 = 0
    return p, [bg, view, proj, campos]


def _round_ws(nbytes: int) -> int:
    """Workspace sizes that depend on the per-view instance count are rounded up to 32 MiB steps, so that the caching
    allocator reuses one block from frame to frame instead of growing a new size class per view."""
    step = 1 << 25
    return max(step, (int(nbytes) + step - 1) // step * step)


# ---- instance capacity of the frames issued without a count read-back ------------------------------------------------
class _CapacityState:
    """Per (device, P, W, H, binning mode): the instance capacity later frames are issued with, the widest depth-key span
    seen (frames are issued without the depth sort's fourth pass while it stays clearly below 2^24), and the workspaces
    forward-only frames share (nothing reads them after the frame: a fresh allocation per frame is pure host time)."""
    __slots__ = ("capacity", "last_counts", "depth_span", "fo_ws", "reissued")

    def __init__(self):
        self.capacity = 0
        self.last_counts = (0, 0)
        self.depth_span = 0     # largest (max depth key - min depth key) of the frames seen
        self.fo_ws = {}         # stream handle -> (capacity, geom, img, binning) of the forward-only frames on that stream
        self.reissued = 0       # frames that did not fit their capacity and were issued again (verified mode)

    def observe(self, R: int, V: int, span: int = 0) -> None:
        self.last_counts = (R, V)
        if span > self.depth_span:
            self.depth_span = span
        want = (int(R * 1.5) + (1 << 20)) >> 20 << 20       # 1.5 x, in steps of 2^20 instances
        if R > 0 and want > self.capacity:
            self.capacity = want


class _Pending:
    """One DEFERRED frame whose instance count has not been compared with its capacity yet."""
    __slots__ = ("event", "slot", "capacity", "state", "done", "counts", "error", "dev_index")

    def __init__(self, event, slot, capacity, state, dev_index):
        self.event, self.slot, self.capacity, self.state, self.dev_index = event, slot, capacity, state, dev_index
        self.done, self.counts, self.error = False, None, None


_MAX_STATES = 16
_states: "collections.OrderedDict[tuple, _CapacityState]" = collections.OrderedDict()
_pending: "collections.deque[_Pending]" = collections.deque()
_free_slots: list = []
_parked_slots: list = []    # (slot, device index) of frames whose enqueue failed half-way: a kernel may still write them
_free_events: dict = {}     # device index -> HIP events created while that device was current
_defer_lock = threading.RLock()
_fo_owner = [None]          # the one state that keeps forward-only workspaces alive (~1 GB at 6 M Gaussians)


def _state_for(key) -> _CapacityState:
    with _defer_lock:
        st = _states.get(key)
        if st is None:
            st = _states[key] = _CapacityState()
            while len(_states) > _MAX_STATES:
                _states.popitem(last=False)
        else:
            _states.move_to_end(key)
        return st


def _keep_forward_only_ws(st: _CapacityState, stream: int, entry: tuple) -> None:
    """Forward-only frames of one stream reuse one set of workspaces; only the most recently used state holds any (a
    second resolution or a densified model takes the memory over instead of adding to it)."""
    with _defer_lock:
        owner = _fo_owner[0]
        if owner is not None and owner is not st:
            owner.fo_ws.clear()
        _fo_owner[0] = st
        if len(st.fo_ws) > 4:
            st.fo_ws.clear()
        st.fo_ws[stream] = entry


_RING = 64                  # deferred frames that may be in flight unchecked; the host waits for the oldest beyond that
_ring_store: list = []      # the one pinned allocation behind the slots (pin_memory() costs ~1 ms: never per frame)


def _pinned_slot() -> torch.Tensor:
    """A 64-byte slice of one pinned block for the counts of a deferred frame.  When all slots are out, the oldest
    pending frame is checked (blocking) to get its slot back."""
    while True:
        with _defer_lock:
            if not _ring_store:
                block = torch.zeros(_RING, 16, dtype=torch.int32).pin_memory()
                _ring_store.append(block)
                _free_slots.extend(block[i] for i in range(_RING))
            if _free_slots:
                return _free_slots.pop()
            oldest = _pending[0] if _pending else None
            parked = list(_parked_slots)
        if oldest is not None:
            _verify(oldest, block=True)
            continue
        if not parked:
            raise _lib.GsrError("pinned count slots exhausted with nothing pending")
        for slot, dev_index in parked:      # frames whose enqueue failed: safe again once their device has drained
            torch.cuda.synchronize(dev_index)
        with _defer_lock:
            for item in parked:
                if item in _parked_slots:
                    _parked_slots.remove(item)
                    _free_slots.append(item[0])


def _new_event(dev_index: int) -> int:
    """A HIP event of device ``dev_index`` (the current device): events are pooled per device -- recording an event
    on a stream of another device is an invalid-handle error."""
    with _defer_lock:
        pool = _free_events.get(dev_index)
        if pool:
            return pool.pop()
    ev = C.c_void_p()
    _lib.check(_lib.load().gsr_event_create(C.byref(ev)), "gsr_event_create")
    return ev.value


def _release_event(event: int, dev_index: int) -> None:
    with _defer_lock:
        _free_events.setdefault(dev_index, []).append(event)


def _verify(pend: _Pending, block: bool) -> bool:
    """Compare a deferred frame's real instance count with the capacity it ran with (waits for the scan kernel of that
    frame when ``block``).  Raises GsrError for an overflowed frame -- every time it is asked about."""
    with _defer_lock:
        if not pend.done:
            lib = _lib.load()
            if block:
                _lib.check(lib.gsr_event_wait(pend.event), "gsr_event_wait")
            else:
                done = C.c_int32(0)
                _lib.check(lib.gsr_event_query(pend.event, C.byref(done)), "gsr_event_query")
                if not done.value:
                    return False
            R, V = int(pend.slot[0]) & 0xffffffff, int(pend.slot[1]) & 0xffffffff
            pend.done, pend.counts = True, (R, V)
            pend.state.observe(R, V)
            _free_slots.append(pend.slot)
            _release_event(pend.event, pend.dev_index)
            pend.slot = pend.event = None
            try:
                _pending.remove(pend)
            except ValueError:
                pass
            if R > pend.capacity:
                pend.error = (f"frame issued in the DEFERRED sync-free mode overflowed its binning capacity: {R} instances "
                              f"> capacity {pend.capacity}; its image and gradients are incomplete and must be discarded "
                              "(later frames get a larger capacity; the default mode, set_sync_free(True), re-issues "
                              "such a frame by itself)")
        if pend.error:
            raise _lib.GsrError(pend.error)
        return True


def _drain_pending(block: bool = False) -> None:
    """Check every earlier deferred frame whose count has arrived (all of them when ``block``)."""
    while True:
        with _defer_lock:
            pend = _pending[0] if _pending else None
        if pend is None or not _verify(pend, block):
            return


def synchronize_counts() -> None:
    """Deferred mode only (a no-op otherwise: verified frames are checked before the operator returns).  Blocks until
    every frame issued so far has had its instance count checked; raises GsrError if one overflowed."""
    _drain_pending(block=True)


def last_counts(dev, P: int, W: int, H: int) -> tuple:
    """(num_rendered, num_visible) of the most recent checked frame of that shape on ``dev`` ((0, 0) if none)."""
    key = (torch.device(dev).index or 0, int(P), int(W), int(H), _binning_mode_value)
    with _defer_lock:
        st = _states.get(key)
        return st.last_counts if st is not None else (0, 0)


def reissued_frames(dev, P: int, W: int, H: int) -> int:
    """Frames of that shape that were issued a second time (verified mode): they did not fit their capacity, or spanned
    2^24 depth-key steps after being issued without the depth sort's fourth pass."""
    key = (torch.device(dev).index or 0, int(P), int(W), int(H), _binning_mode_value)
    with _defer_lock:
        st = _states.get(key)
        return st.reissued if st is not None else 0


_thread_local = threading.local()


def _counts_pinned_thread():
    """Per-thread pinned host words the scan kernel mirrors (num_rendered, num_visible, depth range) into, and a ctypes
    view of them.  A frame of the two-call or the verified path has read them before the operator returns, so one buffer
    per thread serves every frame (forward and backward arrive on different threads)."""
    t = getattr(_thread_local, "pinned", None)
    if t is None:
        t = torch.zeros(16, dtype=torch.int32).pin_memory()
        _thread_local.pinned = t
        _thread_local.words = (C.c_uint32 * 16).from_address(t.data_ptr())
    return t, _thread_local.words


def _thread_event(dev_index: int) -> int:
    """The counts event of the verified path: one per (thread, device), reused by every frame (it is waited for before
    the next frame can record it again)."""
    evs = getattr(_thread_local, "events", None)
    if evs is None:
        evs = _thread_local.events = {}
    ev = evs.get(dev_index)
    if ev is None:
        ev = evs[dev_index] = _new_event(dev_index)
    return ev


class _Frame(NamedTuple):
    """What a forward leaves behind for its backward."""
    geom: torch.Tensor
    binning: torch.Tensor
    img: torch.Tensor
    radii: torch.Tensor
    layout_R: int          # (num_rendered, num_visible) the workspaces are laid out for: the real counts after the
    layout_V: int          #  two-call forward, (capacity, P) after gsr_forward
    pending: Optional[_Pending]     # deferred mode: the check that has not happened yet
    counts: Optional[tuple]         # the real (num_rendered, num_visible) when known


_DEPTH_SORT_BITS = 24                                   # csrc/gsr_common.h: the depth sort's three regular 8-bit passes
_DEPTH_SPAN_TRUSTED = int(0.9 * (1 << _DEPTH_SORT_BITS))


def _depth_span(words, V: int) -> int:
    """max - min depth key of a frame from its pinned counts (0 for a frame without visible Gaussians)."""
    mn, mx = int(words[2]), int(words[3])
    return mx - mn if V > 0 and mx >= mn else 0


def _run_forward(lib, dev, params, P: int, W: int, H: int):
    """Native forward on torch's current stream.  Returns (color, _Frame)."""
    stream = _stream(dev)
    dev_index = dev.index or 0
    radii = torch.empty(P, dtype=torch.int32, device=dev)      # written for every Gaussian by the kernel
    color = torch.empty(3, H, W, dtype=torch.float32, device=dev)
    mode = params.binning_mode
    st = _state_for((dev_index, P, W, H, mode))
    if _pending:
        _drain_pending()
    sync_mode = _sync_free_value
    sync_free = sync_mode != SYNC_OFF and st.capacity > 0 and mode != _lib.BINNING_KEYS64 and P > 0
    cached = st.fo_ws.get(stream) if (sync_free and params.forward_only) else None
    if cached is not None and cached[0] == st.capacity:
        # forward-only frames of one stream run one after the other and nothing outlives them: same workspaces every frame
        _, geom, img, binning = cached
    else:
        geom = torch.empty(lib.gsr_geom_bytes(P), dtype=torch.uint8, device=dev)
        img = torch.empty(lib.gsr_image_bytes(W, H), dtype=torch.uint8, device=dev)
        binning = None
    if sync_free:
        cap = st.capacity
        nbytes = lib.gsr_binning_bytes(cap, P, W, H, mode)
        if binning is None:
            binning = torch.empty(_round_ws(nbytes), dtype=torch.uint8, device=dev)
            if params.forward_only:
                _keep_forward_only_ws(st, stream, (cap, geom, img, binning))
        if sync_mode == SYNC_DEFERRED:
            slot, event = _pinned_slot(), _new_event(dev_index)
            params.counts_pinned = slot.data_ptr()
            pend = _Pending(event, slot, cap, st, dev_index)
            slot[0] = 0             # a frame whose enqueue fails half-way must not be read as an overflow later
            try:
                _lib.check(lib.gsr_forward(C.byref(params), geom.data_ptr(), binning.data_ptr(), nbytes, cap, img.data_ptr(),
                                           radii.data_ptr(), color.data_ptr(), event, stream), "gsr_forward")
            except _lib.GsrError:
                with _defer_lock:   # the scan kernel may already be queued and will write the slot: park it until the
                    _parked_slots.append((slot, dev_index))     # device has drained; the event was never recorded
                _release_event(event, dev_index)
                raise
            with _defer_lock:
                _pending.append(pend)
            return color, _Frame(geom, binning, img, radii, cap, P, pend, None)
        # verified mode: the whole frame is queued, the host waits for its scan kernel only.  Two things are taken on trust
        # from the frames before and checked against the counts: the instance capacity, and -- while every frame seen
        # stayed below 0.9 x 2^24 depth-key steps -- that the depth sort needs no fourth pass (GsrParams.depth_span_lt24:
        # three launches that find nothing to do, 14 us of a 6 M-Gaussian frame and 9 us of a 100 k one).
        pinned, words = _counts_pinned_thread()
        params.counts_pinned = pinned.data_ptr()
        narrow = st.depth_span < _DEPTH_SPAN_TRUSTED
        params.depth_span_lt24 = 1 if narrow else 0
        event = _thread_event(dev_index)
        _lib.check(lib.gsr_forward(C.byref(params), geom.data_ptr(), binning.data_ptr(), nbytes, cap, img.data_ptr(),
                                   radii.data_ptr(), color.data_ptr(), event, stream), "gsr_forward")
        _lib.check(lib.gsr_event_wait(event), "gsr_event_wait")
        params.depth_span_lt24 = 0
        R, V = int(words[0]), int(words[1])
        span = _depth_span(words, V)
        with _defer_lock:        # forward calls of several threads may share this (device, P, W, H) state
            st.observe(R, V, span)
        if R <= cap and not (narrow and span >> _DEPTH_SORT_BITS):
            return color, _Frame(geom, binning, img, radii, cap, P, None, (R, V))
        # The frame did not fit (its kernels dropped the instances past the capacity: no out-of-bounds access), or spans
        # more depth than it was sorted for (lists in the wrong order), and is still running.  Issue it again behind
        # itself, on the two-call path, into the same outputs -- nothing of the wrong frame has left the operator.
        with _defer_lock:
            st.reissued += 1
            if params.forward_only:
                st.fo_ws.pop(stream, None)
    pinned, _words = _counts_pinned_thread()
    params.counts_pinned = pinned.data_ptr()
    num_rendered, num_visible = C.c_uint32(0), C.c_uint32(0)
    _lib.check(lib.gsr_forward_preprocess(C.byref(params), geom.data_ptr(), _ptr(radii), stream,
                                          C.byref(num_rendered), C.byref(num_visible)), "gsr_forward_preprocess")
    R, V = int(num_rendered.value), int(num_visible.value)
    with _defer_lock:
        st.observe(R, V, _depth_span(_words, V))
    nbytes = lib.gsr_binning_bytes(R, V, W, H, mode)
    binning = torch.empty(_round_ws(nbytes), dtype=torch.uint8, device=dev)
    _lib.check(lib.gsr_forward_render(C.byref(params), geom.data_ptr(), binning.data_ptr(), nbytes, img.data_ptr(),
                                      R, V, color.data_ptr(), stream), "gsr_forward_render")
    return color, _Frame(geom, binning, img, radii, R, V, None, (R, V))


def _run_backward(lib, dev, params, frame: _Frame, grad_out_color: torch.Tensor, grads: "_lib.GsrGrads") -> None:
    if frame.pending is not None:
        _verify(frame.pending, block=True)      # deferred mode: the scan kernel of this frame's forward finished long ago
    P = int(params.P)
    nbytes = lib.gsr_backward_bytes(P, frame.layout_R)
    bwd_ws = torch.empty(_round_ws(nbytes), dtype=torch.uint8, device=dev)
    _lib.check(lib.gsr_backward(C.byref(params), _ptr(frame.radii), frame.geom.data_ptr(), frame.binning.data_ptr(),
                                frame.img.data_ptr(), frame.layout_R, frame.layout_V, grad_out_color.data_ptr(),
                                bwd_ws.data_ptr(), nbytes, C.byref(grads), _stream(dev)), "gsr_backward")


def frame_counts(color: torch.Tensor) -> tuple:
    """(num_rendered, num_visible) of the frame that produced ``color`` (an output of the operator that still carries
    its autograd node).  In deferred mode: waits for the frame's count; raises if it overflowed."""
    ctx = color.grad_fn
    if ctx is None or not hasattr(ctx, "frame_pending"):
        raise ValueError("not an output of the rasterizer with an autograd node (rendered under no_grad?)")
    pend = ctx.frame_pending
    if pend is not None:
        _verify(pend, block=True)
        return pend.counts
    return ctx.counts


def _stats_ptrs(stats, P: int, dev):
    """(accum, denom, max_radii2D) pointers of the fused densification statistics, or three Nones."""
    if stats is None:
        return None, None, None
    out = []
    for t in stats:
        if not (t.is_cuda and t.device == dev and t.is_contiguous() and t.dtype == torch.float32 and t.numel() == P):
            raise TypeError("densification accumulators must be contiguous float32 GPU tensors with one element per Gaussian")
        out.append(t.data_ptr())
    return tuple(out)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings: GaussianRasterizationSettings, forward_only: bool = False, stats=None):
        lib = _lib.load()
        dev = _require_gpu(means3D)
        P = int(means3D.shape[0])
        means3D = _f32c(means3D, "means3D", dev)
        sh = _f32c(sh, "shs", dev, align16=True)
        colors_precomp = _f32c(colors_precomp, "colors_precomp", dev)
        opacities = _f32c(opacities, "opacities", dev)
        scales = _f32c(scales, "scales", dev)
        rotations = _f32c(rotations, "rotations", dev, align16=True)
        cov3Ds_precomp = _f32c(cov3Ds_precomp, "cov3D_precomp", dev)
        if opacities.numel() != P:
            raise ValueError("opacities must have one value per Gaussian")
        _check_rows(means3D, "means3D", P, 3)
        _check_rows(sh, "shs", P, None, 3)
        _check_rows(colors_precomp, "colors_precomp", P, 3)
        _check_rows(scales, "scales", P, 3)
        _check_rows(rotations, "rotations", P, 4)
        _check_rows(cov3Ds_precomp, "cov3D_precomp", P, 6)
        if means2D is not None and means2D.numel() and means2D.shape[0] != P:
            raise ValueError(f"means2D must have one row per Gaussian, got {list(means2D.shape)}")
        H, W = int(raster_settings.image_height), int(raster_settings.image_width)

        with torch.cuda.device(dev):
            params, keep = _make_params(dev, raster_settings, means3D, sh, colors_precomp, opacities, scales,
                                        rotations, cov3Ds_precomp, forward_only=forward_only)
            try:
                color, frame = _run_forward(lib, dev, params, P, W, H)
            except _lib.GsrError:
                if raster_settings.debug:
                    torch.save((means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                tuple(raster_settings)), "snapshot_fw.dump")
                    print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise

        ctx.raster_settings = raster_settings
        ctx.profile = _lib.active_profile()     # backward runs on an autograd thread: carry the (live) object explicitly
        ctx.layout = (frame.layout_R, frame.layout_V)
        ctx.frame_pending = frame.pending
        ctx.counts = frame.counts
        ctx.binning_mode = params.binning_mode
        ctx.stats = stats
        ctx.keep = keep
        ctx.save_for_backward(means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, frame.radii,
                              frame.geom, frame.binning, frame.img)
        ctx.mark_non_differentiable(frame.radii)
        ctx.set_materialize_grads(False)        # no zeros_like(radii) per backward for the integer output
        return color, frame.radii

    @staticmethod
    def backward(ctx, grad_out_color, _grad_radii):
        if grad_out_color is None:
            return (None,) * 11
        lib = _lib.load()
        (means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, radii, geom, binning,
         img) = ctx.saved_tensors
        settings = ctx.raster_settings
        dev = means3D.device
        P = int(means3D.shape[0])
        grad_out_color = _f32c(grad_out_color, "grad_out_color", dev, align16=True)

        with torch.cuda.device(dev):
            params, keep = _make_params(dev, settings, means3D, sh, colors_precomp, opacities, scales, rotations,
                                        cov3Ds_precomp)
            params.profile = ctx.profile.handle() if ctx.profile is not None else None
            params.binning_mode = ctx.binning_mode
            new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)  # noqa: E731
            g_means3D, g_means2D, g_opac = new(P, 3), new(P, 3), new(*opacities.shape)
            g_sh = new(*sh.shape) if sh.numel() else None
            g_col = new(P, 3) if colors_precomp.numel() else None
            g_scales = new(P, 3) if scales.numel() else None
            g_rot = new(P, 4) if rotations.numel() else None
            g_cov = new(P, 6) if cov3Ds_precomp.numel() else None
            grads = _lib.GsrGrads(_ptr(g_means3D), _ptr(g_means2D), _ptr(g_sh), _ptr(g_col), _ptr(g_opac),
                                  _ptr(g_scales), _ptr(g_rot), _ptr(g_cov), None, *_stats_ptrs(ctx.stats, P, dev))
            frame = _Frame(geom, binning, img, radii, ctx.layout[0], ctx.layout[1], ctx.frame_pending, ctx.counts)
            try:
                _run_backward(lib, dev, params, frame, grad_out_color, grads)
            except _lib.GsrError:
                if settings.debug:
                    torch.save((means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, radii,
                                grad_out_color, tuple(settings)), "snapshot_bw.dump")
                    print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                raise
        del keep
        return g_means3D, g_means2D, g_sh, g_col, g_opac, g_scales, g_rot, g_cov, None, None, None


class _RasterizeGaussiansFused(torch.autograd.Function):
    """Same operator fed with the RAW parameters of ``scene/gaussian_model.py`` (``_features_dc``,
    ``_features_rest``, ``_opacity``, ``_scaling``, ``_rotation``): the ``cat`` / ``sigmoid`` / ``exp`` /
    ``normalize`` of the getters at ``scene/gaussian_model.py:151-183`` and their autograd run inside the
    preprocess kernels.  Gradients are w.r.t. the raw parameters."""

    @staticmethod
    def forward(ctx, means3D, means2D, f_dc, f_rest, raw_opacity, raw_scales, raw_rotations,
                raster_settings: GaussianRasterizationSettings, forward_only: bool = False, stats=None, visible=None):
        """``visible``: None, or a bool [P] tensor the forward fills with ``radii > 0`` (render()'s visibility_filter)."""
        lib = _lib.load()
        dev = _require_gpu(means3D)
        P = int(means3D.shape[0])
        means3D = _f32c(means3D, "means3D", dev)
        f_dc = _f32c(f_dc, "f_dc", dev)
        f_rest = _f32c(f_rest, "f_rest", dev, align16=True)
        raw_opacity = _f32c(raw_opacity, "opacity", dev)
        raw_scales = _f32c(raw_scales, "scaling", dev)
        raw_rotations = _f32c(raw_rotations, "rotation", dev, align16=True)
        n_rest = int(f_rest.shape[1]) if f_rest.dim() == 3 else -1
        if f_dc.shape[0] != P or f_dc.numel() != 3 * P or f_rest.shape[0] != P or n_rest not in (0, 15):
            raise ValueError("fused inputs need f_dc [P,1,3] and f_rest [P,15,3] (degree-3 storage) or [P,0,3] (degree 0)")
        _check_rows(means3D, "means3D", P, 3)
        if n_rest:
            _check_rows(f_rest, "f_rest", P, 15, 3)
        _check_rows(raw_scales, "scaling", P, 3)
        _check_rows(raw_rotations, "rotation", P, 4)
        if raw_opacity.numel() != P or raw_scales.numel() != 3 * P or raw_rotations.numel() != 4 * P:
            raise ValueError("fused inputs need opacity [P,1], scaling [P,3] and rotation [P,4]")
        H, W = int(raster_settings.image_height), int(raster_settings.image_width)
        empty = torch.empty(0, dtype=torch.float32, device=dev)
        flags = _lib.ACT_SCALE_EXP | _lib.ACT_ROT_NORMALIZE | _lib.ACT_OPACITY_SIGMOID
        with torch.cuda.device(dev):
            # degree-0 storage: f_dc [P,1,3] is the whole SH tensor (M = 1), there is no rest to split off
            params, keep = _make_params(dev, raster_settings, means3D, f_dc, empty, raw_opacity, raw_scales,
                                        raw_rotations, empty, sh_rest=f_rest if n_rest else None, act_flags=flags,
                                        forward_only=forward_only)
            if visible is not None:
                if visible.dtype != torch.bool or visible.numel() != P or not visible.is_contiguous() or visible.device != dev:
                    raise TypeError("visible must be a contiguous bool [P] tensor on the Gaussians' device")
                params.visible_out = visible.data_ptr()
            try:
                color, frame = _run_forward(lib, dev, params, P, W, H)
            except _lib.GsrError:
                if raster_settings.debug:      # same snapshot convention as the getter-fed operator above
                    torch.save((means3D, f_dc, f_rest, raw_opacity, raw_scales, raw_rotations, tuple(raster_settings)),
                               "snapshot_fw.dump")
                    print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise
        ctx.raster_settings = raster_settings
        ctx.profile = _lib.active_profile()
        ctx.layout = (frame.layout_R, frame.layout_V)
        ctx.frame_pending = frame.pending
        ctx.counts = frame.counts
        ctx.binning_mode = params.binning_mode
        ctx.act_flags = flags
        ctx.stats = stats
        ctx.keep = keep
        ctx.save_for_backward(means3D, f_dc, f_rest, raw_opacity, raw_scales, raw_rotations, frame.radii, frame.geom,
                              frame.binning, frame.img)
        ctx.mark_non_differentiable(frame.radii)
        ctx.set_materialize_grads(False)        # no 24-MB zeros_like(radii) per backward for the integer output
        return color, frame.radii

    @staticmethod
    def backward(ctx, grad_out_color, _grad_radii):
        if grad_out_color is None:
            return (None,) * 11
        lib = _lib.load()
        means3D, f_dc, f_rest, raw_opacity, raw_scales, raw_rotations, radii, geom, binning, img = ctx.saved_tensors
        settings = ctx.raster_settings
        dev = means3D.device
        P = int(means3D.shape[0])
        grad_out_color = _f32c(grad_out_color, "grad_out_color", dev, align16=True)
        empty = torch.empty(0, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            has_rest = f_rest.numel() > 0
            params, keep = _make_params(dev, settings, means3D, f_dc, empty, raw_opacity, raw_scales, raw_rotations,
                                        empty, sh_rest=f_rest if has_rest else None, act_flags=ctx.act_flags)
            params.profile = ctx.profile.handle() if ctx.profile is not None else None
            params.binning_mode = ctx.binning_mode
            new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)  # noqa: E731
            g_means3D, g_means2D = new(P, 3), new(P, 3)
            g_dc, g_rest = new(*f_dc.shape), new(*f_rest.shape)
            g_opac, g_scales, g_rot = new(*raw_opacity.shape), new(P, 3), new(P, 4)
            grads = _lib.GsrGrads(_ptr(g_means3D), _ptr(g_means2D), _ptr(g_dc), None, _ptr(g_opac), _ptr(g_scales),
                                  _ptr(g_rot), None, _ptr(g_rest) if has_rest else None, *_stats_ptrs(ctx.stats, P, dev))
            frame = _Frame(geom, binning, img, radii, ctx.layout[0], ctx.layout[1], ctx.frame_pending, ctx.counts)
            try:
                _run_backward(lib, dev, params, frame, grad_out_color, grads)
            except _lib.GsrError:
                if settings.debug:
                    torch.save((means3D, f_dc, f_rest, raw_opacity, raw_scales, raw_rotations, radii, grad_out_color,
                                tuple(settings)), "snapshot_bw.dump")
                    print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                raise
        del keep
        return g_means3D, g_means2D, g_dc, g_rest, g_opac, g_scales, g_rot, None, None, None, None


def _forward_only(*tensors) -> bool:
    """True when no backward can follow (inference): the library then skips everything only a backward reads."""
    return not (torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors))


class _NoGraph:
    """Stands in for the autograd context when no backward can follow: the operator's forward runs as a plain function
    (``Function.apply`` and its bookkeeping are ~20 us of host time per frame, a tenth of a 100 k-Gaussian frame)."""

    def save_for_backward(self, *tensors):
        pass

    def mark_non_differentiable(self, *tensors):
        pass

    def set_materialize_grads(self, value):
        pass


def rasterize_gaussians_fused(means3D, means2D, f_dc, f_rest, raw_opacity, raw_scales, raw_rotations, raster_settings,
                              densify_stats=None, visible=None):
    """``densify_stats``: None, or (xyz_gradient_accum, denom, max_radii2D) -- the backward then also accumulates the
    densification statistics of ``scene/gaussian_model.py:775-777`` / ``train.py:130``.
    ``visible``: None, or a bool [P] tensor that receives ``radii > 0`` from the preprocess kernel."""
    if _forward_only(means3D, means2D, f_dc, f_rest, raw_opacity, raw_scales, raw_rotations):
        with torch.no_grad():
            return _RasterizeGaussiansFused.forward(_NoGraph(), means3D, means2D, f_dc, f_rest, raw_opacity, raw_scales,
                                                    raw_rotations, raster_settings, True, None, visible)
    return _RasterizeGaussiansFused.apply(means3D, means2D, f_dc, f_rest, raw_opacity, raw_scales, raw_rotations,
                                          raster_settings, False, densify_stats, visible)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings, densify_stats=None):
    if _forward_only(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp):
        with torch.no_grad():
            return _RasterizeGaussians.forward(_NoGraph(), means3D, means2D, sh, colors_precomp, opacities, scales,
                                               rotations, cov3Ds_precomp, raster_settings, True, None)
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings, False, densify_stats)


class GaussianRasterizer(nn.Module):
    """Same call contract as the module the reference constructs per frame
    (``gaussian_renderer/__init__.py:57``) and calls at ``:257-265``."""

    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        """Frustum (near-plane) visibility of the upstream module's ``markVisible``; bool ``[P]``."""
        lib = _lib.load()
        dev = _require_gpu(positions)
        with torch.no_grad(), torch.cuda.device(dev):
            pos = _f32c(positions, "positions", dev)
            view = _f32c(self.raster_settings.viewmatrix, "viewmatrix", dev)
            vis = torch.empty(pos.shape[0], dtype=torch.uint8, device=dev)
            _lib.check(lib.gsr_mark_visible(int(pos.shape[0]), _ptr(pos), view.data_ptr(), _ptr(vis), _stream(dev)),
                       "gsr_mark_visible")
        return vis.bool()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, densify_stats=None):
        raster_settings = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
           ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        empty = torch.empty(0, dtype=torch.float32, device=means3D.device)
        shs = empty if shs is None else shs
        colors_precomp = empty if colors_precomp is None else colors_precomp
        scales = empty if scales is None else scales
        rotations = empty if rotations is None else rotations
        cov3D_precomp = empty if cov3D_precomp is None else cov3D_precomp
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, raster_settings, densify_stats)

    def forward_fused(self, means3D, means2D, f_dc, f_rest, raw_opacity, raw_scales, raw_rotations, densify_stats=None):
        """Raw-parameter entry: see :class:`_RasterizeGaussiansFused`."""
        return rasterize_gaussians_fused(means3D, means2D, f_dc, f_rest, raw_opacity, raw_scales, raw_rotations,
                                         self.raster_settings, densify_stats)
