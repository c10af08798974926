"""Pure-PyTorch CPU restatement of the tile-based differentiable Gaussian rasterizer.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``): the parity checker and the
"pure-PyTorch CPU rasterizer" baseline ``BASELINE.json:north_star`` asks for.

What it restates
----------------
The operator the reference calls at ``gaussian_renderer/__init__.py:57,257-265``
(``GaussianRasterizer(raster_settings)(means3D, means2D, shs, colors_precomp,
opacities, scales, rotations, cov3D_precomp)``).  Its native source is absent
from ``/root/reference`` (un-vendored submodule, ``.gitmodules:4-6``), so the
algorithm follows ``SURVEY.md`` Appendix A.1-A.6, constrained by the in-repo
Python twins of its sub-steps:

* SH -> RGB ....... ``utils/sh_utils.py:57-112`` + ``gaussian_renderer/__init__.py:80-84``
* cov3D ........... ``scene/gaussian_model.py:28-32`` + ``utils/general_utils.py:64-110``
* matrices ........ ``scene/cameras.py:48-57`` + ``utils/graphics_utils.py:22-29,51-71``
* L1 loss ......... ``utils/loss_utils.py:17-18``

Everything is dtype-generic: run it in float32 for forward parity / CPU timing and
in float64 (autograd) as the gradient truth.  Every arithmetic expression is spelled
out as individual elementwise ops in a fixed order (no matmul) so that the HIP
kernels, compiled with ``-ffp-contract=off`` for the per-Gaussian stages, can mirror
the float32 op sequence exactly.

Two places where the upstream backward is *not* the exact derivative are kept
(``upstream_grad=True``, default), so that autograd of this file reproduces it:

* alpha = min(0.99, o*G) is back-propagated as if unclamped (A.5);
* conic = inverse(cov2D) uses ``1/(det^2 + 1e-7)`` in its backward (A.6 (i)).
"""
from __future__ import annotations

import math
from typing import Dict, NamedTuple, Optional, Sequence, Tuple

import numpy as np
import torch

TILE = 16
NEAR_Z = 0.2
DILATION = 0.3
FOV_GUARD = 1.3
ALPHA_MAX = 0.99
ALPHA_MIN = 1.0 / 255.0
T_STOP = 1e-4

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
         -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658,
         0.3731763325901154, -0.4570457994644658, 1.445305721320277,
         -0.5900435899266435)


class RasterSettings(NamedTuple):
    """The 12 fields built at ``gaussian_renderer/__init__.py:42-55``."""
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
    prefiltered: bool = False
    debug: bool = False


# --------------------------------------------------------------------------------------
# sub-steps that have a Python twin in the reference
# --------------------------------------------------------------------------------------
def eval_sh_ref(deg: int, sh: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    """SH -> colour *before* the +0.5 / clamp.  ``sh`` is ``[P, M, 3]`` (the layout of
    ``GaussianModel.get_features``, ``scene/gaussian_model.py:176-179``), ``dirs`` is
    ``[P, 3]`` unit vectors.  Polynomial and evaluation order of
    ``utils/sh_utils.py:74-100`` (degrees 0..3)."""
    assert 0 <= deg <= 3
    assert sh.shape[1] >= (deg + 1) ** 2
    result = SH_C0 * sh[:, 0]
    if deg > 0:
        x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
        result = result - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
        if deg > 1:
            xx, yy, zz = x * x, y * y, z * z
            xy, yz, xz = x * y, y * z, x * z
            result = (result
                      + SH_C2[0] * xy * sh[:, 4]
                      + SH_C2[1] * yz * sh[:, 5]
                      + SH_C2[2] * (2.0 * zz - xx - yy) * sh[:, 6]
                      + SH_C2[3] * xz * sh[:, 7]
                      + SH_C2[4] * (xx - yy) * sh[:, 8])
            if deg > 2:
                result = (result
                          + SH_C3[0] * y * (3.0 * xx - yy) * sh[:, 9]
                          + SH_C3[1] * xy * z * sh[:, 10]
                          + SH_C3[2] * y * (4.0 * zz - xx - yy) * sh[:, 11]
                          + SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy) * sh[:, 12]
                          + SH_C3[4] * x * (4.0 * zz - xx - yy) * sh[:, 13]
                          + SH_C3[5] * z * (xx - yy) * sh[:, 14]
                          + SH_C3[6] * x * (xx - 3.0 * yy) * sh[:, 15])
    return result


def build_cov3d_ref(scales: torch.Tensor, scale_modifier: float, rot: torch.Tensor) -> torch.Tensor:
    """``[P,6]`` (xx,xy,xz,yy,yz,zz) of ``R diag(s^2) R^T``; quaternion (w,x,y,z) used as
    passed, not re-normalised (A.2).  Twin: ``scene/gaussian_model.py:28-32`` with
    ``utils/general_utils.py:64-73,90-110`` (which normalises first; the host hands the
    rasterizer already-normalised rotations, ``scene/gaussian_model.py:168-169``)."""
    s = scale_modifier * scales
    r, x, y, z = rot[:, 0], rot[:, 1], rot[:, 2], rot[:, 3]
    R = [[1.0 - 2.0 * (y * y + z * z), 2.0 * (x * y - r * z), 2.0 * (x * z + r * y)],
         [2.0 * (x * y + r * z), 1.0 - 2.0 * (x * x + z * z), 2.0 * (y * z - r * x)],
         [2.0 * (x * z - r * y), 2.0 * (y * z + r * x), 1.0 - 2.0 * (x * x + y * y)]]
    L = [[R[i][k] * s[:, k] for k in range(3)] for i in range(3)]

    def dot(i, j):
        return L[i][0] * L[j][0] + L[i][1] * L[j][1] + L[i][2] * L[j][2]

    return torch.stack([dot(0, 0), dot(0, 1), dot(0, 2), dot(1, 1), dot(1, 2), dot(2, 2)], dim=1)


def l1_loss_ref(x: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    """``utils/loss_utils.py:17-18``."""
    return torch.abs(x - gt).mean()


# --------------------------------------------------------------------------------------
# pieces whose upstream backward is not the exact derivative
# --------------------------------------------------------------------------------------
class _ConicUpstream(torch.autograd.Function):
    """conic = inverse of [[a,b],[b,c]]; backward per Appendix A.6 (i) with the
    ``1/(det^2 + 1e-7)`` denominator.  Gradients are in the *true-derivative* convention
    (the incoming d/dconic_xy is the full derivative, not upstream's internal half)."""

    @staticmethod
    def forward(ctx, a, b, c):
        det = a * c - b * b
        det_inv = 1.0 / det
        ctx.save_for_backward(a, b, c)
        return c * det_inv, -b * det_inv, a * det_inv

    @staticmethod
    def backward(ctx, gx, gy, gz):
        a, b, c = ctx.saved_tensors
        den = a * c - b * b
        k = 1.0 / (den * den + 1e-7)
        da = k * (-c * c * gx + b * c * gy + (den - a * c) * gz)
        dc = k * (-a * a * gz + a * b * gy + (den - a * c) * gx)
        db = k * (2.0 * b * c * gx - (den + 2.0 * b * b) * gy + 2.0 * a * b * gz)
        return da, db, dc


def _conic(a, b, c, upstream_grad: bool):
    if upstream_grad:
        return _ConicUpstream.apply(a, b, c)
    det_inv = 1.0 / (a * c - b * b)
    return c * det_inv, -b * det_inv, a * det_inv


# --------------------------------------------------------------------------------------
# A.2 preprocess
# --------------------------------------------------------------------------------------
def _xform3(p, m):
    """[x,y,z,1] @ m, first three components (row-vector convention, A.1)."""
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    return [m[0, c] * x + m[1, c] * y + m[2, c] * z + m[3, c] for c in range(3)]


def _xform4(p, m):
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    return [m[0, c] * x + m[1, c] * y + m[2, c] * z + m[3, c] for c in range(4)]


def preprocess_ref(means3D: torch.Tensor, opacities: torch.Tensor, settings: RasterSettings, *,
                   shs: Optional[torch.Tensor] = None, colors_precomp: Optional[torch.Tensor] = None,
                   scales: Optional[torch.Tensor] = None, rotations: Optional[torch.Tensor] = None,
                   cov3D_precomp: Optional[torch.Tensor] = None,
                   means2D: Optional[torch.Tensor] = None,
                   upstream_grad: bool = True) -> Dict[str, torch.Tensor]:
    """Appendix A.2.  Works on the compacted set of Gaussians that pass the near-plane cull
    so that no inf/nan is ever produced on a differentiable path.

    Returns a dict; tensors prefixed ``v_`` are indexed by *visible-candidate* slot
    (``idx`` maps slot -> Gaussian index); ``radii``/``tiles_touched``/``depth`` are full ``[P]``.
    """
    if (shs is None) == (colors_precomp is None):
        raise ValueError("Please provide excatly one of either SHs or precomputed colors!")
    if ((scales is None or rotations is None) and cov3D_precomp is None) or \
       ((scales is not None or rotations is not None) and cov3D_precomp is not None):
        raise ValueError("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")

    dt = means3D.dtype
    P = means3D.shape[0]
    H, W = int(settings.image_height), int(settings.image_width)
    V = settings.viewmatrix.to(dt)
    M = settings.projmatrix.to(dt)
    campos = settings.campos.to(dt)
    tanx, tany = float(settings.tanfovx), float(settings.tanfovy)
    grid_x, grid_y = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE

    radii = torch.zeros(P, dtype=torch.int32)
    tiles_touched = torch.zeros(P, dtype=torch.int64)
    depth_full = torch.zeros(P, dtype=dt)

    with torch.no_grad():
        z_all = _xform3(means3D, V)[2]
        idx = torch.nonzero(z_all > NEAR_Z, as_tuple=False).squeeze(1)

    p = means3D[idx]
    pv = _xform3(p, V)
    ph = _xform4(p, M)
    p_w = 1.0 / (ph[3] + 1e-7)
    ndc_x, ndc_y = ph[0] * p_w, ph[1] * p_w
    if means2D is not None:
        # grad carrier: value unchanged, d/d(means2D[:, :2]) == d/d(ndc)  (A.5 scaling 0.5*W, 0.5*H)
        m2 = means2D[idx]
        ndc_x = ndc_x + (m2[:, 0] - m2[:, 0].detach())
        ndc_y = ndc_y + (m2[:, 1] - m2[:, 1].detach())

    if cov3D_precomp is not None:
        cov3D = cov3D_precomp[idx]
    else:
        cov3D = build_cov3d_ref(scales[idx], float(settings.scale_modifier), rotations[idx])

    # ---- EWA projection -------------------------------------------------------------
    # scalars are formed in the working dtype (float32 run == the kernel's float32 scalars)
    tanx_t, tany_t = torch.tensor(tanx, dtype=dt), torch.tensor(tany, dtype=dt)
    fx = W / (2.0 * tanx_t)
    fy = H / (2.0 * tany_t)
    limx, limy = FOV_GUARD * tanx_t, FOV_GUARD * tany_t
    tz = pv[2]
    txtz = pv[0] / tz
    tytz = pv[1] / tz
    tx = torch.minimum(limx, torch.maximum(-limx, txtz)) * tz
    ty = torch.minimum(limy, torch.maximum(-limy, tytz)) * tz
    if upstream_grad:
        # A.6 (ii): inside the guard band tx == view x (d/dx = 1, d/dz = 0); outside, the whole
        # tx/ty path is zeroed (including its lim*z dependence on z).  Values stay bit-identical.
        in_x = ((txtz >= -limx) & (txtz <= limx)).to(dt)
        in_y = ((tytz >= -limy) & (tytz <= limy)).to(dt)
        tx = tx.detach() + in_x * (pv[0] - pv[0].detach())
        ty = ty.detach() + in_y * (pv[1] - pv[1].detach())
    j00 = fx / tz
    j02 = -(fx * tx) / (tz * tz)
    j11 = fy / tz
    j12 = -(fy * ty) / (tz * tz)
    # Wv[i][j] = V[j][i]: view = Wv . world
    A0 = [j00 * V[j, 0] + j02 * V[j, 2] for j in range(3)]
    A1 = [j11 * V[j, 1] + j12 * V[j, 2] for j in range(3)]
    S = [[cov3D[:, 0], cov3D[:, 1], cov3D[:, 2]],
         [cov3D[:, 1], cov3D[:, 3], cov3D[:, 4]],
         [cov3D[:, 2], cov3D[:, 4], cov3D[:, 5]]]
    B0 = [A0[0] * S[0][j] + A0[1] * S[1][j] + A0[2] * S[2][j] for j in range(3)]
    B1 = [A1[0] * S[0][j] + A1[1] * S[1][j] + A1[2] * S[2][j] for j in range(3)]
    a = (B0[0] * A0[0] + B0[1] * A0[1] + B0[2] * A0[2]) + DILATION
    b = B0[0] * A1[0] + B0[1] * A1[1] + B0[2] * A1[2]
    c = (B1[0] * A1[0] + B1[1] * A1[1] + B1[2] * A1[2]) + DILATION

    with torch.no_grad():
        det = a * c - b * b
        ok = det != 0
        mid = 0.5 * (a + c)
        sq = torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
        lam = torch.maximum(mid + sq, mid - sq)
        rad = torch.ceil(3.0 * torch.sqrt(lam))
        px = ((ndc_x + 1.0) * W - 1.0) * 0.5
        py = ((ndc_y + 1.0) * H - 1.0) * 0.5

        def lo(v, g):
            return torch.clamp(torch.trunc((v - rad) / TILE), 0, g)

        def hi(v, g):
            return torch.clamp(torch.trunc((v + rad + (TILE - 1)) / TILE), 0, g)

        # non-finite guard (cannot occur for z > 0.2 and finite inputs, kept for safety)
        fin = torch.isfinite(rad) & torch.isfinite(px) & torch.isfinite(py)
        ok = ok & fin
        rad = torch.where(fin, rad, torch.zeros_like(rad))
        pxs = torch.where(fin, px, torch.zeros_like(px))
        pys = torch.where(fin, py, torch.zeros_like(py))
        x0, x1 = lo(pxs, grid_x), hi(pxs, grid_x)
        y0, y1 = lo(pys, grid_y), hi(pys, grid_y)
        area = ((x1 - x0) * (y1 - y0)).to(torch.int64)
        ok = ok & (area > 0)

    # safe inversion (det == 0 rows are dropped by `ok` anyway)
    one = torch.ones_like(a)
    a_s = torch.where(ok, a, one)
    b_s = torch.where(ok, b, torch.zeros_like(b))
    c_s = torch.where(ok, c, one)
    cxx, cxy, cyy = _conic(a_s, b_s, c_s, upstream_grad)
    pix_x = ((ndc_x + 1.0) * W - 1.0) * 0.5
    pix_y = ((ndc_y + 1.0) * H - 1.0) * 0.5

    # ---- colour ---------------------------------------------------------------------
    if colors_precomp is not None:
        rgb = colors_precomp[idx]
        clamped = torch.zeros(idx.shape[0], 3, dtype=torch.bool)
    else:
        d = p - campos[None, :]
        ln = torch.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2])
        dirs = d / ln[:, None]
        raw = eval_sh_ref(int(settings.sh_degree), shs[idx], dirs) + 0.5
        clamped = (raw < 0).detach()
        rgb = torch.clamp_min(raw, 0.0)

    keep = ok
    vis_idx = idx[keep]
    radii[vis_idx] = rad[keep].to(torch.int32)
    tiles_touched[vis_idx] = area[keep]
    depth_full[vis_idx] = pv[2][keep].detach()

    return {
        "idx": idx, "keep": keep,
        "v_depth": pv[2], "v_xy": torch.stack([pix_x, pix_y], dim=1),
        "v_conic": torch.stack([cxx, cxy, cyy], dim=1), "v_opacity": opacities[idx].reshape(-1),
        "v_rgb": rgb, "v_clamped": clamped, "v_cov3D": cov3D, "v_cov2D": torch.stack([a, b, c], dim=1),
        "v_rect": torch.stack([x0, y0, x1, y1], dim=1).to(torch.int64),
        "radii": radii, "tiles_touched": tiles_touched, "depth": depth_full,
        "grid": (grid_x, grid_y),
    }


# --------------------------------------------------------------------------------------
# A.3 binning
# --------------------------------------------------------------------------------------
def duplicate_with_keys_ref(pre: Dict[str, torch.Tensor]) -> Tuple[np.ndarray, np.ndarray]:
    """Emit one (key, value) per (Gaussian, overlapped tile), y outer / x inner, Gaussians in
    index order.  key = tile_id << 32 | bits(float32 depth); value = Gaussian index."""
    grid_x, _ = pre["grid"]
    keep = pre["keep"].numpy()
    gid = pre["idx"].numpy()[keep]
    rect = pre["v_rect"].numpy()[keep]
    depth = pre["v_depth"].detach().to(torch.float32).numpy()[keep]
    w = rect[:, 2] - rect[:, 0]
    cnt = w * (rect[:, 3] - rect[:, 1])
    R = int(cnt.sum())
    start = np.cumsum(cnt) - cnt
    local = np.arange(R, dtype=np.int64) - np.repeat(start, cnt)
    wr = np.repeat(w, cnt)
    ty = np.repeat(rect[:, 1], cnt) + local // np.maximum(wr, 1)
    tx = np.repeat(rect[:, 0], cnt) + local % np.maximum(wr, 1)
    tile = (ty * grid_x + tx).astype(np.uint64)
    dbits = np.repeat(depth.view(np.uint32).astype(np.uint64), cnt)
    keys = (tile << np.uint64(32)) | dbits
    vals = np.repeat(gid, cnt).astype(np.uint32)
    return keys, vals


def bin_ref(pre: Dict[str, torch.Tensor]) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """A.3: duplicate, stable sort by key, tile ranges.  Equal keys keep emission (= Gaussian
    index) order, which is what a stable LSD radix sort yields.
    Returns (sorted_keys u64[R], point_list u32[R], ranges i64[T,2])."""
    grid_x, grid_y = pre["grid"]
    keys, vals = duplicate_with_keys_ref(pre)
    order = np.argsort(keys, kind="stable")
    keys, vals = keys[order], vals[order]
    T = grid_x * grid_y
    tile = (keys >> np.uint64(32)).astype(np.int64)
    ranges = np.zeros((T, 2), dtype=np.int64)
    if keys.size:
        first = np.searchsorted(tile, np.arange(T), side="left")
        last = np.searchsorted(tile, np.arange(T), side="right")
        nonempty = last > first
        ranges[nonempty, 0] = first[nonempty]
        ranges[nonempty, 1] = last[nonempty]
    return keys, vals, ranges


# --------------------------------------------------------------------------------------
# A.4 render (autograd gives A.5)
# --------------------------------------------------------------------------------------
def _alpha_st(o, G, upstream_grad: bool):
    raw = o * G
    if upstream_grad:
        return raw + (torch.clamp_max(raw, ALPHA_MAX) - raw).detach()
    return torch.clamp_max(raw, ALPHA_MAX)


COND_EPS = 2.5e-7   # ~4 ulp(float32): rounding of the three products, their sums and the pre-scaled coefficients


def render_tiles_ref(pre: Dict[str, torch.Tensor], point_list: np.ndarray, ranges: np.ndarray,
                     settings: RasterSettings, *, tiles: Optional[Sequence[int]] = None,
                     chunk: int = 256, upstream_grad: bool = True, want_margin: bool = False,
                     guard_stats: Optional[dict] = None):
    """Front-to-back alpha compositing per 16x16 tile (A.4).  ``tiles`` restricts the work
    to a subset (used for the bounded CPU-baseline sample); untouched pixels stay 0.

    Returns (color[3,H,W], final_T[H,W], n_contrib[H,W] int32[, margin[H,W]]).  ``margin`` is
    the smallest relative distance of any evaluated decision of that pixel to its threshold
    (alpha vs 1/255, T vs 1e-4): pixels with a tiny margin may legitimately flip between
    float implementations and are excluded from tight comparisons by the tests.  A pixel whose
    float32 evaluation is ill-conditioned is flagged the same way (margin 0): for strongly
    correlated ("needle") Gaussians the three terms of ``power`` cancel, so two float32 evaluation
    orders (fma contraction, pre-scaled coefficients) differ by ~1e-7 x (sum of |terms|) in
    ``power`` and by as much, relatively, in alpha; the pixel is robust only while the blended
    bound sum_i w_i * COND_EPS * (|terms|_i) stays below 3e-6 (COND_EPS = 4 float32 ulps).

    ``guard_stats`` (a dict, filled in place): how often upstream's ``power > 0 -> skip`` guard (A.4) fired on a pair the
    pixel was still considering -- ``pairs`` (count) and ``pix`` (bool [H,W]: pixels with at least one such pair).  The
    true exponent is never positive; the guard only fires on the rounding noise of this expanded three-term form.
    """
    dt = pre["v_xy"].dtype
    H, W = int(settings.image_height), int(settings.image_width)
    grid_x, grid_y = pre["grid"]
    bg = settings.bg.to(dt)
    P_slots = pre["idx"].shape[0]
    # Gaussian index -> visible slot
    slot_of = torch.full((int(pre["radii"].shape[0]),), -1, dtype=torch.int64)
    slot_of[pre["idx"]] = torch.arange(P_slots)
    plist = torch.from_numpy(point_list.astype(np.int64))

    xy, conic, opac, rgb = pre["v_xy"], pre["v_conic"], pre["v_opacity"], pre["v_rgb"]
    Hp, Wp = grid_y * TILE, grid_x * TILE
    tile_imgs = {}
    final_T = torch.ones(Hp, Wp, dtype=dt)
    n_contrib = torch.zeros(Hp, Wp, dtype=torch.int32)
    margin = torch.full((Hp, Wp), float("inf"), dtype=dt) if want_margin else None
    guard_pix = torch.zeros(Hp, Wp, dtype=torch.bool) if guard_stats is not None else None
    guard_pairs = 0

    lx = torch.arange(TILE).repeat(TILE)
    ly = torch.arange(TILE).repeat_interleave(TILE)
    a_min = torch.tensor(ALPHA_MIN, dtype=dt)
    t_stop = torch.tensor(T_STOP, dtype=dt)

    tile_iter = range(grid_x * grid_y) if tiles is None else tiles
    for t in tile_iter:
        ty, tx = divmod(int(t), grid_x)
        pxi = tx * TILE + lx
        pyi = ty * TILE + ly
        inside = (pxi < W) & (pyi < H)
        pxf, pyf = pxi.to(dt), pyi.to(dt)
        s, e = int(ranges[t, 0]), int(ranges[t, 1])
        T = torch.ones(TILE * TILE, dtype=dt)
        C = torch.zeros(3, TILE * TILE, dtype=dt)
        done = ~inside
        last = torch.zeros(TILE * TILE, dtype=torch.int64)
        mg = torch.full((TILE * TILE,), float("inf"), dtype=dt) if want_margin else None
        cond = torch.zeros(TILE * TILE, dtype=dt) if want_margin else None
        pos = s
        while pos < e and not bool(done.all()):
            n = min(chunk, e - pos)
            sl = slot_of[plist[pos:pos + n]]
            g_xy, g_con, g_o, g_rgb = xy[sl], conic[sl], opac[sl], rgb[sl]
            dx = g_xy[:, 0:1] - pxf[None, :]
            dy = g_xy[:, 1:2] - pyf[None, :]
            power = -0.5 * (g_con[:, 0:1] * dx * dx + g_con[:, 2:3] * dy * dy) - g_con[:, 1:2] * dx * dy
            G = torch.exp(power)
            alpha = _alpha_st(g_o[:, None], G, upstream_grad)
            valid = (power <= 0) & (alpha >= a_min)
            one_minus = torch.where(valid, 1.0 - alpha, torch.ones_like(alpha))
            cp = torch.cumprod(one_minus, dim=0) * T[None, :]          # test_T after each entry
            live = (cp >= t_stop) & (~done)[None, :]                   # monotone in the entry index
            T_excl = torch.cat([T[None, :], cp[:-1]], dim=0)
            use = valid & live
            w = torch.where(use, alpha * T_excl, torch.zeros_like(alpha))
            C = C + torch.einsum("np,nc->cp", w, g_rgb)
            n_live = live.sum(dim=0)
            has = n_live > 0
            if guard_stats is not None:
                with torch.no_grad():
                    before = torch.cat([(~done)[None, :], live[:-1]], 0)      # the pixel was still composing at this entry
                    fired = (power > 0) & before
                    guard_pairs += int(fired.sum())
                    gp = fired.any(dim=0).reshape(TILE, TILE)
                    guard_pix[ty * TILE:(ty + 1) * TILE, tx * TILE:(tx + 1) * TILE] |= gp
            T_new = cp.gather(0, (n_live - 1).clamp(min=0)[None, :])[0]
            T = torch.where(has, T_new, T)
            ar = torch.arange(1, n + 1)[:, None] + (pos - s)
            last = torch.maximum(last, torch.where(use, ar, torch.zeros_like(ar)).max(dim=0).values)
            if want_margin:
                with torch.no_grad():
                    considered = live | (torch.cat([torch.ones(1, live.shape[1], dtype=torch.bool), live[:-1]], 0)
                                         & (~done)[None, :])
                    m_a = torch.where(considered & (power <= 0), (alpha - a_min).abs() / a_min,
                                      torch.full_like(alpha, float("inf")))
                    m_t = torch.where(considered & valid, (cp - t_stop).abs() / t_stop,
                                      torch.full_like(alpha, float("inf")))
                    mg = torch.minimum(mg, torch.minimum(m_a, m_t).min(dim=0).values)
                    terms = (0.5 * (g_con[:, 0:1] * dx * dx).abs() + 0.5 * (g_con[:, 2:3] * dy * dy).abs()
                             + (g_con[:, 1:2] * dx * dy).abs())
                    cond = cond + (w.detach() * terms * COND_EPS).sum(dim=0)
            done = done | (n_live < n)
            pos += n
        out = C + T[None, :] * bg[:, None]
        tile_imgs[int(t)] = out
        ys, xs = ty * TILE, tx * TILE
        final_T[ys:ys + TILE, xs:xs + TILE] = T.detach().reshape(TILE, TILE)
        n_contrib[ys:ys + TILE, xs:xs + TILE] = last.to(torch.int32).reshape(TILE, TILE)
        if want_margin:
            mg = torch.where(cond > 3e-6, torch.zeros_like(mg), mg)
            margin[ys:ys + TILE, xs:xs + TILE] = mg.reshape(TILE, TILE)

    # assemble the padded image functionally (keeps autograd)
    zero_tile = torch.zeros(3, TILE * TILE, dtype=dt)
    rows = []
    for ty in range(grid_y):
        row = [tile_imgs.get(ty * grid_x + tx, zero_tile).reshape(3, TILE, TILE) for tx in range(grid_x)]
        rows.append(torch.cat(row, dim=2))
    color = torch.cat(rows, dim=1)[:, :H, :W]
    res = (color, final_T[:H, :W], n_contrib[:H, :W])
    if want_margin:
        res = res + (margin[:H, :W],)
    if guard_stats is not None:
        guard_stats["pairs"] = guard_pairs
        guard_stats["pix"] = guard_pix[:H, :W]
    return res


def rasterize_ref(means3D: torch.Tensor, means2D: Optional[torch.Tensor], opacities: torch.Tensor,
                  settings: RasterSettings, *, shs=None, colors_precomp=None, scales=None, rotations=None,
                  cov3D_precomp=None, upstream_grad: bool = True, want_margin: bool = False,
                  tiles: Optional[Sequence[int]] = None, want_aux: bool = False, guard_stats: Optional[dict] = None):
    """Whole operator: returns ``(color[3,H,W], radii[P] int32)`` like the reference call at
    ``gaussian_renderer/__init__.py:257-265`` (plus an aux dict when ``want_aux``)."""
    pre = preprocess_ref(means3D, opacities, settings, shs=shs, colors_precomp=colors_precomp,
                         scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp,
                         means2D=means2D, upstream_grad=upstream_grad)
    keys, plist, ranges = bin_ref(pre)
    out = render_tiles_ref(pre, plist, ranges, settings, tiles=tiles, upstream_grad=upstream_grad,
                           want_margin=want_margin, guard_stats=guard_stats)
    if want_aux:
        aux = {"pre": pre, "keys": keys, "point_list": plist, "ranges": ranges,
               "final_T": out[1], "n_contrib": out[2]}
        if want_margin:
            aux["margin"] = out[3]
        return out[0], pre["radii"], aux
    return out[0], pre["radii"]
