"""ctypes wrapper of the plain-C oracle (``oracle/c/gsr_oracle.c``) -- TEST INFRASTRUCTURE ONLY.
Independent scalar restatement of the forward pass used to cross-check the PyTorch oracle and the HIP path."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle_c.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            subprocess.run(["make", "-C", os.path.join(_HERE, "c")], check=True)
        _lib = C.CDLL(_LIB)
        _lib.gsr_oracle_forward.restype = C.c_long
    return _lib


def _f(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def forward_c(means3D, opacities, settings, shs=None, colors_precomp=None, scales=None, rotations=None,
              cov3D_precomp=None):
    """-> dict(color[3,H,W], radii[P], final_T, n_contrib, keys, point_list, ranges).  Inputs: torch / numpy."""
    lib = load()
    t = lambda x: None if x is None else (x.detach().cpu().numpy() if hasattr(x, "detach") else x)  # noqa: E731
    means3D, opacities = _f(t(means3D)), _f(t(opacities)).reshape(-1)
    shs, colors_precomp, scales, rotations, cov3D_precomp = (_f(t(v)) for v in (shs, colors_precomp, scales, rotations, cov3D_precomp))
    P = means3D.shape[0]
    M = 0 if shs is None else shs.shape[1]
    H, W = int(settings.image_height), int(settings.image_width)
    V, Mx = _f(t(settings.viewmatrix)), _f(t(settings.projmatrix))
    campos, bg = _f(t(settings.campos)), _f(t(settings.bg))
    color = np.zeros((3, H, W), np.float32)
    radii = np.zeros(P, np.int32)
    final_T = np.zeros((H, W), np.float32)
    n_contrib = np.zeros((H, W), np.uint32)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    ranges = np.zeros((gx * gy, 2), np.int64)
    p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)  # noqa: E731
    args = [C.c_int(P), C.c_int(M), C.c_int(int(settings.sh_degree)), C.c_int(W), C.c_int(H),
            C.c_float(settings.tanfovx), C.c_float(settings.tanfovy), C.c_float(settings.scale_modifier),
            p(means3D), p(shs), p(colors_precomp), p(opacities), p(scales), p(rotations), p(cov3D_precomp),
            p(V), p(Mx), p(campos), p(bg), p(color), p(radii), p(final_T), p(n_contrib)]
    R = lib.gsr_oracle_forward(*args, None, None, None)
    if R < 0:
        raise MemoryError("gsr_oracle_forward")
    keys = np.zeros(max(R, 1), np.uint64)
    plist = np.zeros(max(R, 1), np.uint32)
    lib.gsr_oracle_forward(*args, p(keys), p(plist), p(ranges))
    return {"color": color, "radii": radii, "final_T": final_T, "n_contrib": n_contrib, "keys": keys[:R],
            "point_list": plist[:R], "ranges": ranges, "R": int(R)}
