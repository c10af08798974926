"""ctypes binding of ``libgsr_hip.so`` (the C ABI declared in ``include/gsr.h``).

There is no CPU fallback: if the shared library is missing or does not load, importing the
operator raises, and every call on a non-GPU tensor raises.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# GSR_LIB_PATH (read once, at import): lets the A/B tuning tools (tools/ab_*.sh) point the binding at a candidate build
# instead of copying it over the in-tree library
LIB_PATH = os.environ.get("GSR_LIB_PATH") or os.path.join(_HERE, "libgsr_hip.so")

ABI_VERSION = 13


class GsrParams(C.Structure):
    _fields_ = [
        ("P", C.c_int32), ("M", C.c_int32), ("D", C.c_int32),
        ("width", C.c_int32), ("height", C.c_int32),
        ("tan_fovx", C.c_float), ("tan_fovy", C.c_float), ("scale_modifier", C.c_float),
        ("prefiltered", C.c_int32), ("debug", C.c_int32),
        ("means3D", C.c_void_p), ("shs", C.c_void_p), ("colors_precomp", C.c_void_p),
        ("opacities", C.c_void_p), ("scales", C.c_void_p), ("rotations", C.c_void_p),
        ("cov3D_precomp", C.c_void_p), ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p),
        ("campos", C.c_void_p), ("bg", C.c_void_p), ("profile", C.c_void_p),
        ("shs_rest", C.c_void_p), ("act_flags", C.c_int32), ("binning_mode", C.c_int32),
        ("counts_pinned", C.c_void_p), ("forward_only", C.c_int32), ("debug_flags", C.c_int32),
        ("visible_out", C.c_void_p), ("depth_span_lt24", C.c_int32),
    ]


class GsrGrads(C.Structure):
    _fields_ = [
        ("dL_dmeans3D", C.c_void_p), ("dL_dmeans2D", C.c_void_p), ("dL_dshs", C.c_void_p),
        ("dL_dcolors", C.c_void_p), ("dL_dopacities", C.c_void_p), ("dL_dscales", C.c_void_p),
        ("dL_drotations", C.c_void_p), ("dL_dcov3D", C.c_void_p), ("dL_dshs_rest", C.c_void_p),
        ("stats_xyz_gradient_accum", C.c_void_p), ("stats_denom", C.c_void_p), ("stats_max_radii2D", C.c_void_p),
    ]

ACT_SCALE_EXP, ACT_ROT_NORMALIZE, ACT_OPACITY_SIGMOID = 1, 2, 4
BINNING_TWO_LEVEL, BINNING_KEYS64, BINNING_TWO_LEVEL_CULLED = 0, 1, 2
DSSIM_ONE_MINUS_MEAN, DSSIM_CLAMPED_HALF = 0, 1
DEBUG_NO_MINIBLOCK_CULL = 1


# name -> (restype, argtypes); every symbol include/gsr.h declares
SYMBOLS = {
    "gsr_abi_version": (C.c_int, []),
    "gsr_last_error": (C.c_char_p, []),
    "gsr_build_info": (C.c_char_p, []),
    "gsr_geom_bytes": (C.c_size_t, [C.c_int32]),
    "gsr_image_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "gsr_binning_bytes": (C.c_size_t, [C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_int32]),
    "gsr_backward_bytes": (C.c_size_t, [C.c_int32, C.c_uint32]),
    "gsr_forward_preprocess": (C.c_int, [C.POINTER(GsrParams), C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "gsr_forward_render": (C.c_int, [C.POINTER(GsrParams), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                     C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "gsr_forward": (C.c_int, [C.POINTER(GsrParams), C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p,
                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gsr_event_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "gsr_event_destroy": (C.c_int, [C.c_void_p]),
    "gsr_event_wait": (C.c_int, [C.c_void_p]),
    "gsr_event_query": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "gsr_enable_markers": (C.c_int, [C.c_int32]),
    "gsr_backward": (C.c_int, [C.POINTER(GsrParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                               C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(GsrGrads), C.c_void_p]),
    "gsr_mark_visible": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gsr_sort_scratch_bytes": (C.c_size_t, [C.c_uint32]),
    "gsr_sort_pairs_u64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32,
                                     C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]),
    "gsr_debug_read_geom": (C.c_int, [C.c_void_p, C.c_int32] + [C.c_void_p] * 9),
    "gsr_debug_read_binning": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32,
                                         C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gsr_debug_read_counts": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_uint32), C.c_void_p]),
    "gsr_debug_read_image": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p]),
    "gsr_l1_loss_workspace_bytes": (C.c_size_t, []),
    "gsr_l1_loss_fwd_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p]),
    "gsr_debug_render_stats": (C.c_int, [C.POINTER(GsrParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                         C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gsr_l1_dssim_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "gsr_l1_dssim_loss_fwd_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                            C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gsr_splat2d_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "gsr_splat2d_forward": (C.c_int, [C.c_int32] * 4 + [C.c_void_p] * 7 + [C.c_size_t, C.c_void_p,
                                      C.POINTER(C.c_int32), C.c_void_p]),
    "gsr_splat2d_backward": (C.c_int, [C.c_int32] * 4 + [C.c_void_p] * 5 + [C.c_size_t] + [C.c_void_p] * 7),
    "gsr_knn3_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "gsr_dist2_knn3": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gsr_densify_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "gsr_densify_plan": (C.c_int, [C.c_int32] + [C.c_void_p] * 4 + [C.c_float] * 4 + [C.c_void_p, C.c_size_t,
                                   C.POINTER(C.c_uint32), C.c_void_p]),
    "gsr_densify_gather_rows": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.c_int32,
                                          C.c_void_p, C.c_void_p]),
    "gsr_densify_split_children": (C.c_int, [C.c_int32] + [C.c_void_p] * 5 + [C.POINTER(C.c_uint32), C.c_void_p,
                                             C.c_void_p, C.c_void_p]),
    "gsr_profile_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "gsr_profile_destroy": (C.c_int, [C.c_void_p]),
    "gsr_profile_collect": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]),
    "gsr_stage_name": (C.c_char_p, [C.c_int32]),
    "gsr_densify_stats": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p]),
}

_lib = None
_lock = threading.Lock()


class GsrError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load (once) and type the library.  Raises ``GsrError`` if it is missing: build it with
    ``python -c 'import __graft_entry__ as g; g.build()'`` or ``make -C mvs_gaussian_splatting_amd/csrc``."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise GsrError(f"HIP extension not built: {LIB_PATH} is missing (no CPU fallback exists); "
                           f"run `make -C {os.path.join(_HERE, 'csrc')}`")
        try:
            lib = C.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover - depends on the machine
            raise GsrError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.gsr_abi_version() != ABI_VERSION:
            raise GsrError(f"ABI mismatch: library {lib.gsr_abi_version()} != binding {ABI_VERSION}")
        _lib = lib
    return _lib


def enable_markers(on: bool = True) -> None:
    """roctx ranges ("gsr:<stage>") around every stage, for ``rocprofv3 --marker-trace``; off by default."""
    check(load().gsr_enable_markers(1 if on else 0), "gsr_enable_markers")


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().gsr_last_error().decode("utf-8", "replace")
        raise GsrError(f"{what} failed (code {rc}): {msg}")


STAGE_COUNT = 8
_active_profile = threading.local()


class StageProfile:
    """HIP-event stage timers of the C library (``gsr_profile_*``).  While active (``with prof:``) every
    rasterizer call made from this thread records an event pair around each stage on its stream."""

    def __init__(self):
        self._h = C.c_void_p()
        check(load().gsr_profile_create(C.byref(self._h)), "gsr_profile_create")

    def __enter__(self):
        _active_profile.obj = self
        return self

    def __exit__(self, *exc):
        _active_profile.obj = None

    def handle(self):
        """The native handle, or None once closed (a backward that outlives close() then runs untimed)."""
        return self._h if self._h else None

    def collect(self):
        """-> {stage name: (total ms, intervals)}; blocks on the recorded events and clears them."""
        lib = load()
        ms = (C.c_double * STAGE_COUNT)()
        cnt = (C.c_uint32 * STAGE_COUNT)()
        check(lib.gsr_profile_collect(self._h, ms, cnt), "gsr_profile_collect")
        return {lib.gsr_stage_name(i).decode(): (ms[i], cnt[i]) for i in range(STAGE_COUNT)}

    def close(self):
        if self._h:
            load().gsr_profile_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def active_profile():
    """The StageProfile active on this thread (autograd contexts hold on to the object, not the raw handle, so
    that the handle cannot be freed under a retained graph's backward)."""
    return getattr(_active_profile, "obj", None)


def active_profile_handle():
    obj = active_profile()
    return obj.handle() if obj is not None else None
