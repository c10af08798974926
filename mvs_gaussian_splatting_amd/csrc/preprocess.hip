// Per-Gaussian stages: preprocess forward and preprocess backward (§8 a11).
// One lane per Gaussian, 256-lane blocks.  Compiled with -ffp-contract=off so that the fp32
// op sequence equals the CPU oracle's elementwise sequence (oracle/rasterizer_ref.py).
//
// Behaviour restated from SURVEY.md Appendix A.2 / A.6 (the upstream CUDA source is absent
// from /root/reference); Python twins of the sub-steps in the reference:
//   SH -> RGB   utils/sh_utils.py:57-112, gaussian_renderer/__init__.py:80-84
//   cov3D       scene/gaussian_model.py:28-32, utils/general_utils.py:64-110
//   matrices    scene/cameras.py:48-57, utils/graphics_utils.py:22-29
#include "gsr_common.h"
#include "gsr_launch.h"

namespace gsr {

constexpr float SH_C0 = 0.28209479177387814f;
constexpr float SH_C1 = 0.4886025119029199f;
constexpr float SH_C2_0 = 1.0925484305920792f;
constexpr float SH_C2_1 = -1.0925484305920792f;
constexpr float SH_C2_2 = 0.31539156525252005f;
constexpr float SH_C2_3 = -1.0925484305920792f;
constexpr float SH_C2_4 = 0.5462742152960396f;
constexpr float SH_C3_0 = -0.5900435899266435f;
constexpr float SH_C3_1 = 2.890611442640554f;
constexpr float SH_C3_2 = -0.4570457994644658f;
constexpr float SH_C3_3 = 0.3731763325901154f;
constexpr float SH_C3_4 = -0.4570457994644658f;
constexpr float SH_C3_5 = 1.445305721320277f;
constexpr float SH_C3_6 = -0.5900435899266435f;

struct Mat4 { float m[16]; };   // m[4*row + col] of the row-vector-convention tensor

__device__ inline void load_mat(const float* __restrict__ src, Mat4& d) {
#pragma unroll
  for (int i = 0; i < 16; ++i) d.m[i] = src[i];
}

// cov3D (xx,xy,xz,yy,yz,zz) = R diag(s^2) R^T, quaternion used as passed (A.2)
__device__ inline void cov3d_from_scale_rot(float sx, float sy, float sz, float mod, float r, float x, float y,
                                            float z, float* cov) {
  const float s0 = mod * sx, s1 = mod * sy, s2 = mod * sz;
  const float R00 = 1.0f - 2.0f * (y * y + z * z), R01 = 2.0f * (x * y - r * z), R02 = 2.0f * (x * z + r * y);
  const float R10 = 2.0f * (x * y + r * z), R11 = 1.0f - 2.0f * (x * x + z * z), R12 = 2.0f * (y * z - r * x);
  const float R20 = 2.0f * (x * z - r * y), R21 = 2.0f * (y * z + r * x), R22 = 1.0f - 2.0f * (x * x + y * y);
  const float L00 = R00 * s0, L01 = R01 * s1, L02 = R02 * s2;
  const float L10 = R10 * s0, L11 = R11 * s1, L12 = R12 * s2;
  const float L20 = R20 * s0, L21 = R21 * s1, L22 = R22 * s2;
  cov[0] = L00 * L00 + L01 * L01 + L02 * L02;
  cov[1] = L00 * L10 + L01 * L11 + L02 * L12;
  cov[2] = L00 * L20 + L01 * L21 + L02 * L22;
  cov[3] = L10 * L10 + L11 * L11 + L12 * L12;
  cov[4] = L10 * L20 + L11 * L21 + L12 * L22;
  cov[5] = L20 * L20 + L21 * L21 + L22 * L22;
}

// EWA projection intermediates shared by forward and backward.
struct Proj {
  float tx, ty, tz;        // clamped view-space point
  float txtz, tytz;        // unclamped ratios
  float j00, j02, j11, j12;
  float A0[3], A1[3];      // rows of J * Wv
  float a, b, c;           // dilated cov2D
};

__device__ inline void project_cov(const Mat4& V, float vx, float vy, float vz, const float* cov, float fx,
                                   float fy, float limx, float limy, Proj& o) {
  o.tz = vz;
  o.txtz = vx / vz;
  o.tytz = vy / vz;
  o.tx = fminf(limx, fmaxf(-limx, o.txtz)) * vz;
  o.ty = fminf(limy, fmaxf(-limy, o.tytz)) * vz;
  o.j00 = fx / vz;
  o.j02 = -(fx * o.tx) / (vz * vz);
  o.j11 = fy / vz;
  o.j12 = -(fy * o.ty) / (vz * vz);
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    o.A0[j] = o.j00 * V.m[4 * j + 0] + o.j02 * V.m[4 * j + 2];
    o.A1[j] = o.j11 * V.m[4 * j + 1] + o.j12 * V.m[4 * j + 2];
  }
  const float S[3][3] = {{cov[0], cov[1], cov[2]}, {cov[1], cov[3], cov[4]}, {cov[2], cov[4], cov[5]}};
  float B0[3], B1[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    B0[j] = o.A0[0] * S[0][j] + o.A0[1] * S[1][j] + o.A0[2] * S[2][j];
    B1[j] = o.A1[0] * S[0][j] + o.A1[1] * S[1][j] + o.A1[2] * S[2][j];
  }
  o.a = (B0[0] * o.A0[0] + B0[1] * o.A0[1] + B0[2] * o.A0[2]) + DILATION;
  o.b = B0[0] * o.A1[0] + B0[1] * o.A1[1] + B0[2] * o.A1[2];
  o.c = (B1[0] * o.A1[0] + B1[1] * o.A1[1] + B1[2] * o.A1[2]) + DILATION;
}

// SH basis in the evaluation order of utils/sh_utils.py:74-100; sh = [M][3] row of one Gaussian.
template <typename LoadSH>
__device__ inline void eval_sh(int deg, LoadSH sh, float x, float y, float z, float* out) {
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    float res = SH_C0 * sh(0, ch);
    if (deg > 0) {
      res = res - SH_C1 * y * sh(1, ch) + SH_C1 * z * sh(2, ch) - SH_C1 * x * sh(3, ch);
      if (deg > 1) {
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        res = res + SH_C2_0 * xy * sh(4, ch) + SH_C2_1 * yz * sh(5, ch) +
              SH_C2_2 * (2.0f * zz - xx - yy) * sh(6, ch) + SH_C2_3 * xz * sh(7, ch) +
              SH_C2_4 * (xx - yy) * sh(8, ch);
        if (deg > 2) {
          res = res + SH_C3_0 * y * (3.0f * xx - yy) * sh(9, ch) + SH_C3_1 * xy * z * sh(10, ch) +
                SH_C3_2 * y * (4.0f * zz - xx - yy) * sh(11, ch) +
                SH_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh(12, ch) +
                SH_C3_4 * x * (4.0f * zz - xx - yy) * sh(13, ch) + SH_C3_5 * z * (xx - yy) * sh(14, ch) +
                SH_C3_6 * x * (xx - 3.0f * yy) * sh(15, ch);
        }
      }
    }
    out[ch] = res;
  }
}

// ---- fused input activations: the raw parameters of scene/gaussian_model.py:151-183 -------
struct Activated {
  float sc[3];      // activated scales
  float4 q;         // activated (normalised) quaternion
  float qn;         // norm used for the normalisation (1 when not normalising)
  float op;         // activated opacity
};
// the raw values are read first (all of a Gaussian's small loads are issued together, ahead of the arithmetic that
// decides whether they are needed), activated later
__device__ inline void load_scale_rot_raw(const GsrParams& p, int idx, Activated& a) {
  a.sc[0] = p.scales[3 * (size_t)idx];
  a.sc[1] = p.scales[3 * (size_t)idx + 1];
  a.sc[2] = p.scales[3 * (size_t)idx + 2];
  a.q = reinterpret_cast<const float4*>(p.rotations)[idx];
}
__device__ inline void activate_scale_rot(const GsrParams& p, Activated& a) {
  if (p.act_flags & GSR_ACT_SCALE_EXP) {
#pragma unroll
    for (int k = 0; k < 3; ++k) a.sc[k] = expf(a.sc[k]);
  }
  a.qn = 1.0f;
  if (p.act_flags & GSR_ACT_ROT_NORMALIZE) {
    a.qn = fmaxf(sqrtf(a.q.x * a.q.x + a.q.y * a.q.y + a.q.z * a.q.z + a.q.w * a.q.w), 1e-12f);
    a.q.x = a.q.x / a.qn; a.q.y = a.q.y / a.qn; a.q.z = a.q.z / a.qn; a.q.w = a.q.w / a.qn;
  }
}
__device__ inline void load_scale_rot(const GsrParams& p, int idx, Activated& a) {
  load_scale_rot_raw(p, idx, a);
  activate_scale_rot(p, a);
}
__device__ inline float activate_opacity(const GsrParams& p, float o) {
  return (p.act_flags & GSR_ACT_OPACITY_SIGMOID) ? 1.0f / (1.0f + expf(-o)) : o;
}
__device__ inline float load_opacity(const GsrParams& p, int idx) {
  const float o = p.opacities[idx];
  return (p.act_flags & GSR_ACT_OPACITY_SIGMOID) ? 1.0f / (1.0f + expf(-o)) : o;
}

// One Gaussian's 16 x 3 SH coefficients into registers, coefficient-major (k, channel): either the unsplit [P,16,3] row
// (twelve 16-byte loads) or f_dc [P,1,3] + f_rest [P,15,3] (180-byte rows, 4-byte aligned: the compiler still emits
// 16-byte loads in unaligned access mode).  With D = 0 only the DC term is read.
__device__ inline void load_sh_row48(const GsrParams& p, int idx, float* __restrict__ f) {
  if (p.shs_rest) {
    const float* __restrict__ dc = p.shs + 3 * (size_t)idx;
    const float* __restrict__ rr = p.shs_rest + (size_t)idx * 45;
    f[0] = dc[0]; f[1] = dc[1]; f[2] = dc[2];
    if (p.D > 0) {
#pragma unroll
      for (int i = 0; i < 45; ++i) f[3 + i] = rr[i];
    }
  } else {
    const float4* s4 = reinterpret_cast<const float4*>(p.shs + (size_t)idx * 48);
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const float4 v = s4[k];
      f[4 * k] = v.x; f[4 * k + 1] = v.y; f[4 * k + 2] = v.z; f[4 * k + 3] = v.w;
    }
  }
}

// Split SH inputs: f_rest is [P,15,3] (180-byte rows, not 16-byte aligned per row).  A wave's 64 rows are
// 11520 contiguous bytes: they are moved with 16-byte accesses into / out of an LDS image with the natural
// row stride 45 (odd: conflict-free), and each lane then works on its own row.
constexpr int REST_ROW = 45;
__device__ inline void wave_load_rows45(const float* __restrict__ base, int first, int rows_valid, float* st,
                                        int lane) {
  const float* __restrict__ src = base + (size_t)first * REST_ROW;
  const int nfl = rows_valid * REST_ROW;
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const int f = (i * WAVE + lane) * 4;
    if (f + 3 < nfl) {
      const float4 v = *reinterpret_cast<const float4*>(src + f);
      st[f] = v.x; st[f + 1] = v.y; st[f + 2] = v.z; st[f + 3] = v.w;
    } else {
      for (int e = 0; e < 4; ++e)
        if (f + e < nfl) st[f + e] = src[f + e];
    }
  }
}
__device__ inline void wave_store_rows45(float* __restrict__ base, int first, int rows_valid, const float* st,
                                         int lane) {
  float* __restrict__ dst = base + (size_t)first * REST_ROW;
  const int nfl = rows_valid * REST_ROW;
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const int f = (i * WAVE + lane) * 4;
    if (f + 3 < nfl) {
      *reinterpret_cast<float4*>(dst + f) = make_float4(st[f], st[f + 1], st[f + 2], st[f + 3]);
    } else {
      for (int e = 0; e < 4; ++e)
        if (f + e < nfl) dst[f + e] = st[f + e];
    }
  }
}

__global__ __launch_bounds__(PRE_BLOCK) void preprocess_fwd_kernel(GsrParams p, GeomRec* __restrict__ rec,
                                                                   BinInfo* __restrict__ bin,
                                                                   uint32_t* __restrict__ block_sums,
                                                                   uint32_t* __restrict__ block_vis,
                                                                   int32_t* __restrict__ radii,
                                                                   uint32_t* __restrict__ block_big,
                                                                   uint32_t* __restrict__ big_list,
                                                                   uint2* __restrict__ block_range) {
  __shared__ uint32_t wave_big[PRE_BLOCK / WAVE];
  __shared__ uint32_t wave_sums[PRE_BLOCK / WAVE];
  __shared__ uint32_t wave_vis[PRE_BLOCK / WAVE];
  __shared__ uint2 wave_range[PRE_BLOCK / WAVE];
  const int idx = blockIdx.x * PRE_BLOCK + threadIdx.x;
  const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
  const int W = p.width, H = p.height;
  const int grid_x = (W + TILE - 1) / TILE, grid_y = (H + TILE - 1) / TILE;
  uint32_t tiles = 0;
  int32_t radius = 0;
  BinInfo bi{0u, 0u, 0.0f, 0u};
  GeomRec g;
  bool vis = false;
  int rx0 = 0, ry0 = 0, rw = 0, rh = 0;
  float px = 0.f, py = 0.f, pz = 0.f;
  Proj pr;
  pr.a = pr.c = 1.0f;
  Activated act;
  float op_raw = 0.0f;
  // SH rows held in registers: the M = 16 layouts (split or not); other M read their coefficients in place
  const bool sh_regs = !p.colors_precomp && (p.shs_rest != nullptr || p.M == 16);
  bool sh_early = false;
  float shrow[48];

  // ---- geometry: cull, project, cov3D -> cov2D -> conic, radius, tile rect -----------------------
  if (idx < p.P) {
    Mat4 V, Mx;
    load_mat(p.viewmatrix, V);
    load_mat(p.projmatrix, Mx);
    px = p.means3D[3 * (size_t)idx + 0]; py = p.means3D[3 * (size_t)idx + 1]; pz = p.means3D[3 * (size_t)idx + 2];
    // every small per-Gaussian load goes out with the position: one memory latency instead of three in a row
    // (scales / rotation after the near-plane test, opacity after the colour) for 32 bytes that nearly every
    // Gaussian needs anyway
    if (!p.cov3D_precomp) load_scale_rot_raw(p, idx, act);
    op_raw = p.opacities[idx];
    const float vx = V.m[0] * px + V.m[4] * py + V.m[8] * pz + V.m[12];
    const float vy = V.m[1] * px + V.m[5] * py + V.m[9] * pz + V.m[13];
    const float vz = V.m[2] * px + V.m[6] * py + V.m[10] * pz + V.m[14];
    if (vz > NEAR_Z) {
      const float hx = Mx.m[0] * px + Mx.m[4] * py + Mx.m[8] * pz + Mx.m[12];
      const float hy = Mx.m[1] * px + Mx.m[5] * py + Mx.m[9] * pz + Mx.m[13];
      const float hw = Mx.m[3] * px + Mx.m[7] * py + Mx.m[11] * pz + Mx.m[15];
      const float p_w = 1.0f / (hw + 0.0000001f);
      const float ndc_x = hx * p_w, ndc_y = hy * p_w;
      // A Gaussian whose centre is on the screen is all but certainly kept: its SH row is requested now, so that the
      // 192 bytes arrive while the covariance is projected (the rest -- centre outside, splat reaching in -- load late).
      if (sh_regs && fabsf(ndc_x) < 1.0f && fabsf(ndc_y) < 1.0f) {
        sh_early = true;
        load_sh_row48(p, idx, shrow);
      }

      float cov[6];
      if (p.cov3D_precomp) {
#pragma unroll
        for (int k = 0; k < 6; ++k) cov[k] = p.cov3D_precomp[6 * (size_t)idx + k];
      } else {
        activate_scale_rot(p, act);
        cov3d_from_scale_rot(act.sc[0], act.sc[1], act.sc[2], p.scale_modifier, act.q.x, act.q.y, act.q.z, act.q.w, cov);
      }
      const float fx = (float)W / (2.0f * p.tan_fovx), fy = (float)H / (2.0f * p.tan_fovy);
      project_cov(V, vx, vy, vz, cov, fx, fy, FOV_GUARD * p.tan_fovx, FOV_GUARD * p.tan_fovy, pr);
      const float det = pr.a * pr.c - pr.b * pr.b;
      if (det != 0.0f) {
        const float det_inv = 1.0f / det;
        const float mid = 0.5f * (pr.a + pr.c);
        const float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
        const float lam = fmaxf(mid + sq, mid - sq);
        const float rad = ceilf(3.0f * sqrtf(lam));
        const float mx = ((ndc_x + 1.0f) * (float)W - 1.0f) * 0.5f;
        const float my = ((ndc_y + 1.0f) * (float)H - 1.0f) * 0.5f;
        if (__builtin_isfinite(rad) && __builtin_isfinite(mx) && __builtin_isfinite(my)) {
          // clamp in float before the int cast (C truncation toward zero, then clamp to [0, grid])
          const float gxf = (float)grid_x, gyf = (float)grid_y;
          const int x0 = (int)fminf(gxf, fmaxf(0.0f, truncf((mx - rad) / (float)TILE)));
          const int y0 = (int)fminf(gyf, fmaxf(0.0f, truncf((my - rad) / (float)TILE)));
          const int x1 = (int)fminf(gxf, fmaxf(0.0f, truncf((mx + rad + (float)(TILE - 1)) / (float)TILE)));
          const int y1 = (int)fminf(gyf, fmaxf(0.0f, truncf((my + rad + (float)(TILE - 1)) / (float)TILE)));
          const int area = (x1 - x0) * (y1 - y0);
          if (area > 0) {
            vis = true;
            radius = (int32_t)rad;
            tiles = (uint32_t)area;
            rx0 = x0; ry0 = y0; rw = x1 - x0; rh = y1 - y0;
            g.x = mx; g.y = my;
            g.cxx = pr.c * det_inv; g.cxy = -pr.b * det_inv; g.cyy = pr.a * det_inv;
            g.tile_mask = full_mask(tiles);
            g.rect_min = (uint32_t)x0 | ((uint32_t)y0 << 16);
            g.rect_wh = (uint32_t)(x1 - x0) | ((uint32_t)(y1 - y0) << 16);
            g.kk = -pr.b / pr.c;
            g.isyy = 1.0f / pr.c;
            bi.depth = vz;
          }
        }
      }
    }
  }

  // ---- colour ------------------------------------------------------------------------------------
  if (vis) {
    float rgb[3];
    uint32_t flags = 0;
    if (p.colors_precomp) {
      rgb[0] = p.colors_precomp[3 * (size_t)idx];
      rgb[1] = p.colors_precomp[3 * (size_t)idx + 1];
      rgb[2] = p.colors_precomp[3 * (size_t)idx + 2];
    } else {
      const float dx = px - p.campos[0], dy = py - p.campos[1], dz = pz - p.campos[2];
      const float ln = sqrtf(dx * dx + dy * dy + dz * dz);
      const float ux = dx / ln, uy = dy / ln, uz = dz / ln;
      if (sh_regs) {
        if (!sh_early) load_sh_row48(p, idx, shrow);
        eval_sh(p.D, [&](int k, int ch) { return shrow[3 * k + ch]; }, ux, uy, uz, rgb);
      } else {
        const float* __restrict__ s = p.shs + (size_t)idx * p.M * 3;
        eval_sh(p.D, [&](int k, int ch) { return s[3 * k + ch]; }, ux, uy, uz, rgb);
      }
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        rgb[ch] = rgb[ch] + 0.5f;
        if (rgb[ch] < 0.0f) flags |= 1u << ch;
        rgb[ch] = fmaxf(rgb[ch], 0.0f);
      }
    }
    const float op = activate_opacity(p, op_raw);
    // conservative half-extent of the region where alpha = op*exp(power) can reach 1/255
    float ext_x = -1.0f, ext_y = -1.0f;
    if (op >= ALPHA_MIN) {
      const float t = 2.0f * logf(op * 255.0f) * 1.0001f + 1e-4f;
      ext_x = sqrtf(t * pr.a) * 1.0001f + 0.01f;
      ext_y = sqrtf(t * pr.c) * 1.0001f + 0.01f;
    }
    g.opacity = op; g.r = rgb[0]; g.g = rgb[1]; g.b = rgb[2];
    g.ext_x = ext_x; g.ext_y = ext_y;
    if (p.binning_mode == GSR_BINNING_TWO_LEVEL_CULLED) {
      // Drop the tiles the alpha >= 1/255 ellipse cannot reach (every pixel of such a tile fails the alpha test anyway):
      //  1. rects of more than 32 tiles shrink to the tiles the ellipse's bounding box overlaps -- upstream's rect is a
      //     square of 3 sqrt(lambda_max), far too tall for an elongated splat;
      //  2. rects of at most 32 tiles (also after step 1) get a per-tile mask from the same two conservative tests the
      //     compositing kernels apply per sub-block (bounding box, then the exact ellipse-vs-rectangle minimum with a
      //     safety margin) on the tile's 16x16 pixel centres.
      const float inv = 1.0f / (float)TILE;
      int tx_lo = 0, tx_hi = -1, ty_lo = 0, ty_hi = -1;     // empty when no pixel can reach alpha >= 1/255
      if (ext_x >= 0.0f) {
        // tile k holds the pixel centres 16k .. 16k+15
        tx_lo = max(0, (int)ceilf((g.x - ext_x - 15.0f) * inv) - rx0);
        tx_hi = min(rw - 1, (int)floorf((g.x + ext_x) * inv) - rx0);
        ty_lo = max(0, (int)ceilf((g.y - ext_y - 15.0f) * inv) - ry0);
        ty_hi = min(rh - 1, (int)floorf((g.y + ext_y) * inv) - ry0);
      }
      if (tiles > MASK_TILES) {
        if (tx_hi < tx_lo || ty_hi < ty_lo) {
          rw = rh = 1;                       // a 1x1 rect with an empty mask: no instances
          tiles = 0;
          g.tile_mask = 0;
        } else {
          rx0 += tx_lo; ry0 += ty_lo;
          rw = tx_hi - tx_lo + 1; rh = ty_hi - ty_lo + 1;
          tx_hi -= tx_lo; ty_hi -= ty_lo; tx_lo = ty_lo = 0;
          tiles = (uint32_t)(rw * rh);
          g.tile_mask = full_mask(tiles);
        }
        g.rect_min = (uint32_t)rx0 | ((uint32_t)ry0 << 16);
        g.rect_wh = (uint32_t)rw | ((uint32_t)rh << 16);
      }
      if (tiles != 0u && tiles <= MASK_TILES) {
        uint32_t m = 0;
        if (tiles == 1u) {
          m = (tx_hi >= tx_lo && ty_hi >= ty_lo) ? 1u : 0u;
        } else if (tx_hi >= tx_lo && ty_hi >= ty_lo) {
          const float t = 2.0f * __logf(255.0f * op) * 1.001f + 2e-3f;
          const float icxx = __builtin_amdgcn_rcpf(g.cxx), icyy = __builtin_amdgcn_rcpf(g.cyy);
          for (int ty = ty_lo; ty <= ty_hi; ++ty) {
            const float dy0 = (float)((ry0 + ty) * TILE) - g.y;
            for (int tx = tx_lo; tx <= tx_hi; ++tx) {
              const float dx0 = (float)((rx0 + tx) * TILE) - g.x;
              if (!(qmin_rect(g.cxx, g.cxy, g.cyy, icxx, icyy, dx0, dx0 + 15.0f, dy0, dy0 + 15.0f) > t))
                m |= 1u << (ty * rw + tx);
            }
          }
        }
        g.tile_mask = m;
        tiles = (uint32_t)__popc(m);
      }
    }
    bi.rect_min = g.rect_min; bi.rect_wh = g.rect_wh;
    bi.mask = g.tile_mask;
    g.rect_min |= flags << 29;         // the binning copy (bi) stays clean
    rec[idx] = g;
  }
  if (idx < p.P) {
    radii[idx] = radius;
    bin[idx] = bi;
    if (p.visible_out) p.visible_out[idx] = radius > 0 ? 1 : 0;
  }

  // range of the depth keys of the Gaussians that enter the binning: the depth sort works on (bits - min), which
  // needs 24 bits (three 8-bit passes) instead of 32 unless the frame spans more than 2^24 float32 steps of depth.
  // Per block: max(~bits) (i.e. ~min: 0 when the block holds none) and max(bits); the scan kernel folds the blocks.
  uint32_t dinv = tiles != 0u ? ~__float_as_uint(bi.depth) : 0u, dmax = tiles != 0u ? __float_as_uint(bi.depth) : 0u;
#pragma unroll
  for (int d = WAVE / 2; d > 0; d >>= 1) {
    dinv = max(dinv, (uint32_t)__shfl_xor((int)dinv, d, WAVE));
    dmax = max(dmax, (uint32_t)__shfl_xor((int)dmax, d, WAVE));
  }

  // block total of tiles_touched -> first level of the hierarchical scan (§8 a5)
  const uint32_t ws = wave_reduce_add_u32(tiles);
  const uint32_t wv = (uint32_t)__popcll(__ballot(tiles != 0u));   // Gaussians that enter the binning
  // Large splats are rare: their gradient rows are pre-summed cooperatively by the backward (sum_big_rows_kernel), which
  // finds them in this block's stretch of big_list.  (One frame-wide list behind an atomic counter needed a memset node in
  // front of every frame, and on a scene with 40 k such splats the same-address atomics cost this kernel 85 us.)
  const bool big = tiles > ROWS_COOP;
  const unsigned long long big_lanes = __ballot(big);
  if (lane == 0) {
    wave_sums[wid] = ws; wave_vis[wid] = wv; wave_range[wid] = make_uint2(dinv, dmax);
    wave_big[wid] = (uint32_t)__popcll(big_lanes);
  }
  __syncthreads();
  if (big) {
    uint32_t k = (uint32_t)__popcll(big_lanes & ((1ull << lane) - 1ull));
    for (int w = 0; w < wid; ++w) k += wave_big[w];
    big_list[(size_t)blockIdx.x * PRE_BLOCK + k] = (uint32_t)idx;
  }
  if (threadIdx.x == 0) {
    uint32_t nbig = 0;
#pragma unroll
    for (int w = 0; w < PRE_BLOCK / WAVE; ++w) nbig += wave_big[w];
    block_big[blockIdx.x] = nbig;
    uint32_t s = 0, v = 0;
    uint2 r = make_uint2(0u, 0u);
#pragma unroll
    for (int w = 0; w < PRE_BLOCK / WAVE; ++w) {
      s += wave_sums[w]; v += wave_vis[w];
      r.x = max(r.x, wave_range[w].x); r.y = max(r.y, wave_range[w].y);
    }
    block_sums[blockIdx.x] = s;
    block_vis[blockIdx.x] = v;
    block_range[blockIdx.x] = r;
  }
}

// Exclusive scan of per-block totals; one block per array (blockIdx 0: a, 1: b); writes offs[nb] and the total.
// Rounds of 32768 entries: every lane owns 32 contiguous entries (one 128-byte line: all 32 loads in flight at once),
// scans them in registers, the block scans the 1024 lane sums (two barriers), and the lane writes its 32 prefixes.
__global__ __launch_bounds__(1024) void scan_block_sums_kernel(const uint32_t* __restrict__ sums_a,
                                                                uint32_t* __restrict__ offs_a,
                                                                uint32_t* __restrict__ total_a,
                                                                const uint32_t* __restrict__ sums_b,
                                                                uint32_t* __restrict__ offs_b,
                                                                uint32_t* __restrict__ total_b, int nb,
                                                                uint32_t* __restrict__ host_mirror,
                                                                const uint2* __restrict__ block_range,
                                                                const uint32_t* __restrict__ sums_c,
                                                                uint32_t* __restrict__ offs_c,
                                                                uint32_t* __restrict__ total_c) {
  constexpr int PER = 32;
  __shared__ uint32_t wave_tot[1024 / WAVE];
  __shared__ uint2 wave_range[1024 / WAVE];
  if (blockIdx.x == 2) {      // third block (only launched with block_range): fold the per-block depth-key ranges
    uint2 r = make_uint2(0u, 0u);
    for (int b0 = threadIdx.x; b0 < nb; b0 += 8 * 1024) {      // eight independent loads in flight per lane
      uint2 q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) q[u] = b0 + u * 1024 < nb ? block_range[b0 + u * 1024] : make_uint2(0u, 0u);
#pragma unroll
      for (int u = 0; u < 8; ++u) { r.x = max(r.x, q[u].x); r.y = max(r.y, q[u].y); }
    }
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) {
      r.x = max(r.x, (uint32_t)__shfl_xor((int)r.x, d, WAVE));
      r.y = max(r.y, (uint32_t)__shfl_xor((int)r.y, d, WAVE));
    }
    if ((threadIdx.x & (WAVE - 1)) == 0) wave_range[threadIdx.x / WAVE] = r;
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < 1024 / WAVE; ++w) { r.x = max(r.x, wave_range[w].x); r.y = max(r.y, wave_range[w].y); }
      total_a[4] = r.x;                 // total_a = [R, V, big, -, ~min, max]
      total_a[5] = r.y;
      if (host_mirror) { host_mirror[2] = ~r.x; host_mirror[3] = r.y; }
    }
    return;
  }
  const uint32_t* __restrict__ block_sums = blockIdx.x == 0 ? sums_a : blockIdx.x == 1 ? sums_b : sums_c;
  uint32_t* __restrict__ block_offs = blockIdx.x == 0 ? offs_a : blockIdx.x == 1 ? offs_b : offs_c;
  uint32_t* __restrict__ total = blockIdx.x == 0 ? total_a : blockIdx.x == 1 ? total_b : total_c;
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
  uint32_t carry = 0;
  for (int base = 0; base < nb; base += 1024 * PER) {
    const int lo = base + tid * PER;
    uint32_t v[PER];
    if (lo + PER <= nb) {                      // whole 128-byte line: eight 16-byte loads (the arrays are 256-byte aligned)
      const uint4* src = reinterpret_cast<const uint4*>(block_sums + lo);
#pragma unroll
      for (int k = 0; k < PER / 4; ++k) {
        const uint4 q = src[k];
        v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w;
      }
    } else {
#pragma unroll
      for (int k = 0; k < PER; ++k) v[k] = lo + k < nb ? block_sums[lo + k] : 0u;
    }
    uint32_t mine = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) mine += v[k];
    const uint32_t inc = wave_incl_scan_u32(mine);
    if (base) __syncthreads();                 // the previous round's readers of wave_tot are done
    if (lane == WAVE - 1) wave_tot[wid] = inc;
    __syncthreads();
    uint32_t woff = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 1024 / WAVE; ++w) {
      const uint32_t t = wave_tot[w];
      if (w < wid) woff += t;
      all += t;
    }
    uint32_t run = carry + woff + inc - mine;
    if (lo + PER <= nb) {
      uint4* dst = reinterpret_cast<uint4*>(block_offs + lo);
#pragma unroll
      for (int k = 0; k < PER / 4; ++k) {
        uint4 q;
        q.x = run; run += v[4 * k];
        q.y = run; run += v[4 * k + 1];
        q.z = run; run += v[4 * k + 2];
        q.w = run; run += v[4 * k + 3];
        dst[k] = q;
      }
    } else {
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        if (lo + k < nb) block_offs[lo + k] = run;
        run += v[k];
      }
    }
    carry += all;
  }
  if (tid == 0) {
    block_offs[nb] = carry;
    *total = carry;
    if (host_mirror && blockIdx.x < 2) host_mirror[blockIdx.x] = carry;   // pinned host memory: visible once the kernel has completed
  }
}

// ------------------------------------------------------------------------------------------
// preprocess backward (§8 a11, Appendix A.6): per Gaussian, sums its per-instance gradient rows
// (written by the compositing backward) in slot order, then chains conic -> cov2D -> cov3D /
// mean, mean2D -> mean3D, colour -> SH (+ direction), cov3D -> scale / rotation.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PRE_BLOCK) void preprocess_bwd_kernel(GsrParams p, const int32_t* __restrict__ radii,
                                                                   const GeomRec* __restrict__ rec,
                                                                   const uint32_t* __restrict__ slot_base,
                                                                   const GradRow* __restrict__ rows,
                                                                   const uint8_t* __restrict__ row_flags,
                                                                   GsrGrads g) {
  // dL_dsh of 64 Gaussians is 12 KB of contiguous memory: each wave stages its rows in LDS (row stride 49
  // floats: conflict-free) and streams them out with 16-byte stores, instead of 48 lane-strided dword stores.
  constexpr int SH_ROW = 49;
  __shared__ float sh_stage[PRE_BLOCK / WAVE][WAVE * SH_ROW];
  const int idx = blockIdx.x * PRE_BLOCK + threadIdx.x;
  const bool valid = idx < p.P;
  const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
  const int M = p.M;
  const bool split = p.shs_rest != nullptr;
  const bool sh_lds = p.shs != nullptr && M == 16 && g.dL_dshs != nullptr;
  float* my_row = &sh_stage[wid][lane * SH_ROW];
  float dmean[3] = {0.f, 0.f, 0.f};
  float dm2x = 0.f, dm2y = 0.f, dop = 0.f;
  float dcol[3] = {0.f, 0.f, 0.f};
  float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float dscale[3] = {0.f, 0.f, 0.f};
  float drot[4] = {0.f, 0.f, 0.f, 0.f};
  const bool vis = valid && radii[idx] > 0;
  const int wave_first = blockIdx.x * PRE_BLOCK + wid * WAVE;
  const int rows_valid = min(WAVE, p.P - wave_first);
  // ---- (0) deterministic sum of this Gaussian's instance rows ---------------------------------------------------
  // A splat that covers more than ROWS_COOP tiles (thousands, for a large one) was folded into its first row by
  // sum_big_rows_kernel: a single lane walking thousands of flag bytes stalls its whole wave.
  GeomRec r;
  float dcxx = 0.f, dcxy = 0.f, dcyy = 0.f;
  bool any_row = false;
  uint32_t n_rows = 0, slot0 = 0;
  if (vis) {
    r = rec[idx];
    const uint32_t n_all = bin_count(r.rect_wh, r.tile_mask);
    n_rows = n_all > ROWS_COOP ? 1u : n_all;
    slot0 = slot_base[idx];
  }
#ifndef GSR_BWD_ROWS_PER_LANE
  // The wave reads the rows of its 64 Gaussians TOGETHER: the rows are numbered 0 .. T-1 across the wave (prefix sum of the
  // row counts), 256 of them per trip -- every lane requests the flag bytes of four rows, then the flagged rows, whoever
  // owns them -- and are laid out in LDS in that order; each lane then adds ITS rows from LDS in slot order.  The order of
  // the additions is the same as before (bitwise reproducible, independent of what else shares the wave), but a Gaussian
  // with 60 rows no longer holds its wave for thirty dependent trips to memory while the other 63 lanes wait: on a
  // heavy-tailed scene (5.7 rows per Gaussian, up to ROWS_COOP) that was most of this kernel.  The staging area is the
  // wave's SH output stage, which is not in use yet.
  {
    constexpr int RSTRIDE = 10;                                   // nine sums + the flag
    constexpr uint32_t TRIP = 256;
    float* stage = sh_stage[wid];                                 // 64 * 49 floats: 256 rows * 10 + two 64-entry tables fit
    uint32_t* t_excl = reinterpret_cast<uint32_t*>(stage + TRIP * RSTRIDE);
    uint32_t* t_slot = t_excl + WAVE;
    static_assert(TRIP * RSTRIDE + 2 * WAVE <= WAVE * SH_ROW, "row staging must fit the SH stage");
    const uint32_t inc = wave_incl_scan_u32(n_rows), excl = inc - n_rows;
    const uint32_t T = (uint32_t)__shfl((int)inc, WAVE - 1, WAVE);
    t_excl[lane] = excl;
    t_slot[lane] = slot0;
    __builtin_amdgcn_wave_barrier();
    for (uint32_t base = 0; base < T; base += TRIP) {
      uint32_t slot[4];
      uint8_t f[4];
      GradRow q[4];
#pragma unroll
      for (uint32_t u = 0; u < 4; ++u) {
        const uint32_t j = base + u * WAVE + (uint32_t)lane;
        f[u] = 0;
        slot[u] = 0;
        if (j < T) {
          // owner: the last lane whose exclusive prefix is <= j (lanes without rows share their successor's prefix and
          // come before it)
          int lo = 0, hi = WAVE - 1;
#pragma unroll
          for (int it = 0; it < 6; ++it) {
            const int mid = (lo + hi + 1) >> 1;
            if (t_excl[mid] <= j) lo = mid; else hi = mid - 1;
          }
          slot[u] = t_slot[lo] + (j - t_excl[lo]);
          f[u] = row_flags[slot[u]];
        }
      }
#pragma unroll
      for (uint32_t u = 0; u < 4; ++u)
        if (f[u]) q[u] = rows[slot[u]];
#pragma unroll
      for (uint32_t u = 0; u < 4; ++u) {
        float* d = stage + (u * WAVE + (uint32_t)lane) * RSTRIDE;
        if (f[u]) {
          d[0] = q[u].dmx; d[1] = q[u].dmy; d[2] = q[u].dcxx; d[3] = q[u].dcxy; d[4] = q[u].dcyy;
          d[5] = q[u].dop; d[6] = q[u].dr; d[7] = q[u].dg; d[8] = q[u].db;
        }
        d[9] = f[u] ? 1.0f : 0.0f;
      }
      __builtin_amdgcn_wave_barrier();
      const uint32_t lo_j = max(excl, base), hi_j = min(excl + n_rows, base + TRIP);
      for (uint32_t j = lo_j; j < hi_j; ++j) {
        const float* d = stage + (j - base) * RSTRIDE;
        if (d[9] != 0.0f) {
          any_row = true;
          dm2x += d[0]; dm2y += d[1]; dcxx += d[2]; dcxy += d[3]; dcyy += d[4];
          dop += d[5]; dcol[0] += d[6]; dcol[1] += d[7]; dcol[2] += d[8];
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
#else
  if (vis) {
    // ROW_BATCH rows per trip to memory: the flag bytes of a batch are requested together, then the flagged rows -- the
    // additions stay in slot order (bitwise reproducible).
    constexpr uint32_t ROW_BATCH = 4;
    for (uint32_t k = 0; k < n_rows; k += ROW_BATCH) {
      uint8_t f[ROW_BATCH];
      GradRow q[ROW_BATCH];
#pragma unroll
      for (uint32_t u = 0; u < ROW_BATCH; ++u) f[u] = k + u < n_rows ? row_flags[slot0 + k + u] : (uint8_t)0;
#pragma unroll
      for (uint32_t u = 0; u < ROW_BATCH; ++u)
        if (f[u]) q[u] = rows[slot0 + k + u];
#pragma unroll
      for (uint32_t u = 0; u < ROW_BATCH; ++u)
        if (f[u]) {
          any_row = true;
          dm2x += q[u].dmx; dm2y += q[u].dmy; dcxx += q[u].dcxx; dcxy += q[u].dcxy; dcyy += q[u].dcyy;
          dop += q[u].dop; dcol[0] += q[u].dr; dcol[1] += q[u].dg; dcol[2] += q[u].db;
        }
    }
  }
#endif

  // A visible Gaussian that received no row (behind saturated pixels, or below 1/255 everywhere) has every sum zero and
  // therefore every gradient zero: it takes the path of an invisible one -- zeros are written, and its position /
  // scale / rotation / SH row (296 bytes) are not read.
  const bool active = vis && any_row;
  if (split) {
    // dL/df_rest rows are staged in LDS (row stride 45) and streamed out with 16-byte stores
    float* st = sh_stage[wid];
    my_row = st + lane * REST_ROW;
    if (!active) {
#pragma unroll
      for (int k = 0; k < REST_ROW; ++k) my_row[k] = 0.0f;
      if (valid) { g.dL_dshs[3 * (size_t)idx] = 0.f; g.dL_dshs[3 * (size_t)idx + 1] = 0.f; g.dL_dshs[3 * (size_t)idx + 2] = 0.f; }
    }
  } else if (sh_lds && !active) {
#pragma unroll
    for (int k = 0; k < 48; ++k) my_row[k] = 0.0f;
  }

  if (active) {
    {
      // the rows carry the first moments M = sum h * (mean - pixel); the conic is the same for every tile of the
      // Gaussian, so dL/dmean2D = -0.5 * (W, H) .* (conic M) is applied once here instead of per pixel pair
      const float mx = dm2x, my = dm2y;
      dm2x = -0.5f * (float)p.width * (r.cxx * mx + r.cxy * my);
      dm2y = -0.5f * (float)p.height * (r.cxy * mx + r.cyy * my);
    }

    Mat4 V, Mx;
    load_mat(p.viewmatrix, V);
    load_mat(p.projmatrix, Mx);
    const float px = p.means3D[3 * (size_t)idx + 0], py = p.means3D[3 * (size_t)idx + 1],
                pz = p.means3D[3 * (size_t)idx + 2];
    const float vx = V.m[0] * px + V.m[4] * py + V.m[8] * pz + V.m[12];
    const float vy = V.m[1] * px + V.m[5] * py + V.m[9] * pz + V.m[13];
    const float vz = V.m[2] * px + V.m[6] * py + V.m[10] * pz + V.m[14];

    float cov[6];
    float4 q4 = make_float4(1.f, 0.f, 0.f, 0.f);
    float sc[3] = {0.f, 0.f, 0.f};
    Activated act;
    act.qn = 1.0f;
    if (p.cov3D_precomp) {
#pragma unroll
      for (int k = 0; k < 6; ++k) cov[k] = p.cov3D_precomp[6 * (size_t)idx + k];
    } else {
      load_scale_rot(p, idx, act);
      q4 = act.q;
      sc[0] = act.sc[0]; sc[1] = act.sc[1]; sc[2] = act.sc[2];
      cov3d_from_scale_rot(sc[0], sc[1], sc[2], p.scale_modifier, q4.x, q4.y, q4.z, q4.w, cov);
    }
    const float fx = (float)p.width / (2.0f * p.tan_fovx), fy = (float)p.height / (2.0f * p.tan_fovy);
    const float limx = FOV_GUARD * p.tan_fovx, limy = FOV_GUARD * p.tan_fovy;
    Proj pr;
    project_cov(V, vx, vy, vz, cov, fx, fy, limx, limy, pr);

    // ---- (i) conic -> cov2D (true-derivative convention for dcxy; 1e-7 as upstream) -------
    const float a = pr.a, b = pr.b, c = pr.c;
    const float den = a * c - b * b;
    const float k2 = 1.0f / (den * den + 0.0000001f);
    const float dL_da = k2 * (-c * c * dcxx + b * c * dcxy + (den - a * c) * dcyy);
    const float dL_dc = k2 * (-a * a * dcyy + a * b * dcxy + (den - a * c) * dcxx);
    const float dL_db = k2 * (2.0f * b * c * dcxx - (den + 2.0f * b * b) * dcxy + 2.0f * a * b * dcyy);

    // ---- (ii) cov2D = A S A^T  ->  dS (6 unique) and dA ------------------------------------
    const float* A0 = pr.A0;
    const float* A1 = pr.A1;
    // dL/dS_jk (full symmetric matrix entry) = dL_da A0j A0k + dL_db A0j A1k + dL_dc A1j A1k ;
    // unique off-diagonals collect both (j,k) and (k,j).
    auto dS = [&](int j, int k) { return dL_da * A0[j] * A0[k] + dL_db * A0[j] * A1[k] + dL_dc * A1[j] * A1[k]; };
    dcov[0] = dS(0, 0);
    dcov[3] = dS(1, 1);
    dcov[5] = dS(2, 2);
    dcov[1] = dS(0, 1) + dS(1, 0);
    dcov[2] = dS(0, 2) + dS(2, 0);
    dcov[4] = dS(1, 2) + dS(2, 1);
    // dL/dA0_j = 2 dL_da (S A0)_j + dL_db (S A1)_j ; dL/dA1_j = 2 dL_dc (S A1)_j + dL_db (S A0)_j
    const float S[3][3] = {{cov[0], cov[1], cov[2]}, {cov[1], cov[3], cov[4]}, {cov[2], cov[4], cov[5]}};
    float SA0[3], SA1[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      SA0[j] = S[j][0] * A0[0] + S[j][1] * A0[1] + S[j][2] * A0[2];
      SA1[j] = S[j][0] * A1[0] + S[j][1] * A1[1] + S[j][2] * A1[2];
    }
    float dA0[3], dA1[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      dA0[j] = 2.0f * dL_da * SA0[j] + dL_db * SA1[j];
      dA1[j] = 2.0f * dL_dc * SA1[j] + dL_db * SA0[j];
    }
    // A0_j = j00 V[j][0] + j02 V[j][2] ; A1_j = j11 V[j][1] + j12 V[j][2]
    float dj00 = 0.f, dj02 = 0.f, dj11 = 0.f, dj12 = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      dj00 += dA0[j] * V.m[4 * j + 0];
      dj02 += dA0[j] * V.m[4 * j + 2];
      dj11 += dA1[j] * V.m[4 * j + 1];
      dj12 += dA1[j] * V.m[4 * j + 2];
    }
    // j00 = fx/tz ; j02 = -fx tx / tz^2 ; j11 = fy/tz ; j12 = -fy ty / tz^2 ; tx = clamp(vx/vz)*vz
    const float tz = pr.tz, tz2 = tz * tz, tz3 = tz2 * tz;
    const float xm = (pr.txtz < -limx || pr.txtz > limx) ? 0.0f : 1.0f;
    const float ym = (pr.tytz < -limy || pr.tytz > limy) ? 0.0f : 1.0f;
    const float dtx = -fx / tz2 * dj02;                 // d/d(clamped tx)
    const float dty = -fy / tz2 * dj12;
    // clamped tx = clamp(vx/vz) * vz: inside -> tx = vx (d/dvx = 1, d/dvz = 0); outside the guard band
    // upstream zeroes the tx/ty path entirely (A.6 (ii)), including its lim*vz dependence on vz
    float dvx = xm * dtx;
    float dvy = ym * dty;
    float dvz = -fx / tz2 * dj00 - fy / tz2 * dj11 + (2.0f * fx * pr.tx) / tz3 * dj02 + (2.0f * fy * pr.ty) / tz3 * dj12;
    // view = [p,1] @ V  ->  dL/dp_i = sum_c V[i][c] dv_c   (assigned, A.6 (ii))
#pragma unroll
    for (int i = 0; i < 3; ++i) dmean[i] = V.m[4 * i + 0] * dvx + V.m[4 * i + 1] * dvy + V.m[4 * i + 2] * dvz;

    // ---- (iii) mean2D (NDC units) -> mean3D through p_hom / (w + 1e-7) ----------------------
    {
      const float hx = Mx.m[0] * px + Mx.m[4] * py + Mx.m[8] * pz + Mx.m[12];
      const float hy = Mx.m[1] * px + Mx.m[5] * py + Mx.m[9] * pz + Mx.m[13];
      const float hw = Mx.m[3] * px + Mx.m[7] * py + Mx.m[11] * pz + Mx.m[15];
      const float m_w = 1.0f / (hw + 0.0000001f);
      const float mul1 = hx * m_w * m_w, mul2 = hy * m_w * m_w;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        dmean[i] += (Mx.m[4 * i + 0] * m_w - Mx.m[4 * i + 3] * mul1) * dm2x +
                    (Mx.m[4 * i + 1] * m_w - Mx.m[4 * i + 3] * mul2) * dm2y;
      }
    }

    // ---- (iv) colour -> SH coefficients and view direction ----------------------------------
    if (p.shs) {
      float dc[3];
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) dc[ch] = ((rect_clamp_flags(r.rect_min) >> ch) & 1u) ? 0.0f : dcol[ch];
      const float dxr = px - p.campos[0], dyr = py - p.campos[1], dzr = pz - p.campos[2];
      const float ln = sqrtf(dxr * dxr + dyr * dyr + dzr * dzr);
      const float x = dxr / ln, y = dyr / ln, z = dzr / ln;
      const float* __restrict__ s = p.shs + (size_t)idx * M * 3;
      float* __restrict__ ds = g.dL_dshs + (size_t)idx * M * 3;
      const int deg = p.D;
      float dRx[3] = {0.f, 0.f, 0.f}, dRy[3] = {0.f, 0.f, 0.f}, dRz[3] = {0.f, 0.f, 0.f};
      float4 sv4[12];
      if (M == 16 && !split) {
        const float4* s4 = reinterpret_cast<const float4*>(s);
#pragma unroll
        for (int k = 0; k < 12; ++k) sv4[k] = s4[k];
      } else if (split && deg > 0) {
        const float* __restrict__ rr = p.shs_rest + (size_t)idx * REST_ROW;
        float* f = reinterpret_cast<float*>(sv4);
#pragma unroll
        for (int i = 0; i < REST_ROW; ++i) f[3 + i] = rr[i];     // f[3k+ch], k >= 1
      }
      const float* sreg = reinterpret_cast<const float*>(sv4);
      auto emit = [&](int k, float basis) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
          const float v = basis * dc[ch];
          if (split) {
            if (k == 0) g.dL_dshs[3 * (size_t)idx + ch] = v;
            else my_row[3 * (k - 1) + ch] = v;
          } else if (sh_lds) {
            my_row[3 * k + ch] = v;
          } else {
            ds[3 * k + ch] = v;
          }
        }
      };
      // d rgb / d dir: components that are identically zero are skipped (0 * c cannot be folded under IEEE rules and
      // this file is built with -ffp-contract=off: the sums below may contract, they feed no integer decision)
      auto dirg = [&](int k, float bx, float by, float bz) {
#pragma clang fp contract(fast)
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
          const float c = (M == 16) ? sreg[3 * k + ch] : s[3 * k + ch];
          if (!(__builtin_constant_p(bx) && bx == 0.0f)) dRx[ch] += bx * c;
          if (!(__builtin_constant_p(by) && by == 0.0f)) dRy[ch] += by * c;
          if (!(__builtin_constant_p(bz) && bz == 0.0f)) dRz[ch] += bz * c;
        }
      };
      emit(0, SH_C0);
      if (deg > 0) {
        dirg(1, 0.f, -SH_C1, 0.f); dirg(2, 0.f, 0.f, SH_C1); dirg(3, -SH_C1, 0.f, 0.f);
        emit(1, -SH_C1 * y); emit(2, SH_C1 * z); emit(3, -SH_C1 * x);
        if (deg > 1) {
          const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
          dirg(4, SH_C2_0 * y, SH_C2_0 * x, 0.f);
          dirg(5, 0.f, SH_C2_1 * z, SH_C2_1 * y);
          dirg(6, SH_C2_2 * -2.0f * x, SH_C2_2 * -2.0f * y, SH_C2_2 * 4.0f * z);
          dirg(7, SH_C2_3 * z, 0.f, SH_C2_3 * x);
          dirg(8, SH_C2_4 * 2.0f * x, SH_C2_4 * -2.0f * y, 0.f);
          emit(4, SH_C2_0 * xy); emit(5, SH_C2_1 * yz); emit(6, SH_C2_2 * (2.0f * zz - xx - yy));
          emit(7, SH_C2_3 * xz); emit(8, SH_C2_4 * (xx - yy));
          if (deg > 2) {
            dirg(9, SH_C3_0 * 6.0f * xy, SH_C3_0 * (3.0f * xx - 3.0f * yy), 0.f);
            dirg(10, SH_C3_1 * yz, SH_C3_1 * xz, SH_C3_1 * xy);
            dirg(11, SH_C3_2 * -2.0f * xy, SH_C3_2 * (4.0f * zz - xx - 3.0f * yy), SH_C3_2 * 8.0f * yz);
            dirg(12, SH_C3_3 * -6.0f * xz, SH_C3_3 * -6.0f * yz, SH_C3_3 * (6.0f * zz - 3.0f * xx - 3.0f * yy));
            dirg(13, SH_C3_4 * (4.0f * zz - 3.0f * xx - yy), SH_C3_4 * -2.0f * xy, SH_C3_4 * 8.0f * xz);
            dirg(14, SH_C3_5 * 2.0f * xz, SH_C3_5 * -2.0f * yz, SH_C3_5 * (xx - yy));
            dirg(15, SH_C3_6 * (3.0f * xx - 3.0f * yy), SH_C3_6 * -6.0f * xy, 0.f);
            emit(9, SH_C3_0 * y * (3.0f * xx - yy)); emit(10, SH_C3_1 * xy * z);
            emit(11, SH_C3_2 * y * (4.0f * zz - xx - yy));
            emit(12, SH_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy));
            emit(13, SH_C3_4 * x * (4.0f * zz - xx - yy)); emit(14, SH_C3_5 * z * (xx - yy));
            emit(15, SH_C3_6 * x * (xx - 3.0f * yy));
          }
        }
      }
      const int used = (deg + 1) * (deg + 1);
      for (int k = used; k < M; ++k) emit(k, 0.0f);
      const float ddx = dRx[0] * dc[0] + dRx[1] * dc[1] + dRx[2] * dc[2];
      const float ddy = dRy[0] * dc[0] + dRy[1] * dc[1] + dRy[2] * dc[2];
      const float ddz = dRz[0] * dc[0] + dRz[1] * dc[1] + dRz[2] * dc[2];
      // through dir = d / |d|
      const float sum2 = dxr * dxr + dyr * dyr + dzr * dzr;
      const float inv3 = 1.0f / (ln * sum2);
      dmean[0] += ((sum2 - dxr * dxr) * ddx - dyr * dxr * ddy - dzr * dxr * ddz) * inv3;
      dmean[1] += (-dxr * dyr * ddx + (sum2 - dyr * dyr) * ddy - dzr * dyr * ddz) * inv3;
      dmean[2] += (-dxr * dzr * ddx - dyr * dzr * ddy + (sum2 - dzr * dzr) * ddz) * inv3;
    }

    // ---- (v) cov3D -> scale, rotation -------------------------------------------------------
    if (!p.cov3D_precomp) {
      const float mod = p.scale_modifier;
      const float qr = q4.x, qx = q4.y, qy = q4.z, qz = q4.w;
      const float s0 = mod * sc[0], s1 = mod * sc[1], s2 = mod * sc[2];
      const float R[3][3] = {{1.0f - 2.0f * (qy * qy + qz * qz), 2.0f * (qx * qy - qr * qz), 2.0f * (qx * qz + qr * qy)},
                             {2.0f * (qx * qy + qr * qz), 1.0f - 2.0f * (qx * qx + qz * qz), 2.0f * (qy * qz - qr * qx)},
                             {2.0f * (qx * qz - qr * qy), 2.0f * (qy * qz + qr * qx), 1.0f - 2.0f * (qx * qx + qy * qy)}};
      const float sv[3] = {s0, s1, s2};
      // Sigma = L L^T, L = R diag(s).  dL/dSigma as a full symmetric matrix G (off-diag halves).
      const float Gm[3][3] = {{dcov[0], 0.5f * dcov[1], 0.5f * dcov[2]},
                              {0.5f * dcov[1], dcov[3], 0.5f * dcov[4]},
                              {0.5f * dcov[2], 0.5f * dcov[4], dcov[5]}};
      // dL/dL = 2 G L
      float dLm[3][3];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k)
          dLm[i][k] = 2.0f * (Gm[i][0] * R[0][k] * sv[k] + Gm[i][1] * R[1][k] * sv[k] + Gm[i][2] * R[2][k] * sv[k]);
      float dR[3][3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          acc += dLm[i][k] * R[i][k];
          dR[i][k] = dLm[i][k] * sv[k];
        }
        dscale[k] = acc * mod;
      }
      // R(q) derivative, quaternion as passed (no normalisation Jacobian, A.6 (v))
      drot[0] = 2.0f * (qz * (dR[1][0] - dR[0][1]) + qy * (dR[0][2] - dR[2][0]) + qx * (dR[2][1] - dR[1][2]));
      drot[1] = 2.0f * (qy * (dR[0][1] + dR[1][0]) + qz * (dR[0][2] + dR[2][0]) + qr * (dR[2][1] - dR[1][2])) -
                4.0f * qx * (dR[1][1] + dR[2][2]);
      drot[2] = 2.0f * (qx * (dR[0][1] + dR[1][0]) + qr * (dR[0][2] - dR[2][0]) + qz * (dR[1][2] + dR[2][1])) -
                4.0f * qy * (dR[0][0] + dR[2][2]);
      drot[3] = 2.0f * (qr * (dR[1][0] - dR[0][1]) + qx * (dR[0][2] + dR[2][0]) + qy * (dR[1][2] + dR[2][1])) -
                4.0f * qz * (dR[0][0] + dR[1][1]);
      // chain through the fused activations back to the raw parameters
      if (p.act_flags & GSR_ACT_SCALE_EXP) {
#pragma unroll
        for (int k = 0; k < 3; ++k) dscale[k] *= sc[k];
      }
      if (p.act_flags & GSR_ACT_ROT_NORMALIZE) {
        const float dot = qr * drot[0] + qx * drot[1] + qy * drot[2] + qz * drot[3];
        drot[0] = (drot[0] - qr * dot) / act.qn;
        drot[1] = (drot[1] - qx * dot) / act.qn;
        drot[2] = (drot[2] - qy * dot) / act.qn;
        drot[3] = (drot[3] - qz * dot) / act.qn;
      }
    }
    if (p.act_flags & GSR_ACT_OPACITY_SIGMOID) dop *= r.opacity * (1.0f - r.opacity);
  } else if (valid && p.shs && g.dL_dshs && !sh_lds) {
    float* __restrict__ ds = g.dL_dshs + (size_t)idx * M * 3;
    for (int k = 0; k < M * 3; ++k) ds[k] = 0.0f;
  }

  if (split) {
    __builtin_amdgcn_wave_barrier();
    wave_store_rows45(g.dL_dshs_rest, wave_first, rows_valid, sh_stage[wid], lane);
  } else if (sh_lds) {
    // the wave's 64 rows = 3072 contiguous floats of dL_dsh; LDS ops of one wave execute in order
    __builtin_amdgcn_wave_barrier();
    const int first = wave_first;
    float* __restrict__ out = g.dL_dshs + (size_t)first * 48;
    const float* __restrict__ st = sh_stage[wid];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int flat = i * (WAVE * 4) + lane * 4;
      const int row = flat / 48, c = flat - row * 48;
      if (row < rows_valid) {
        const float* q = st + row * SH_ROW + c;
        *reinterpret_cast<float4*>(out + flat) = make_float4(q[0], q[1], q[2], q[3]);
      }
    }
  }
  if (!valid) return;

  // ---- write every output row in full (no caller zero-fill needed) ---------------------------
#pragma unroll
  for (int i = 0; i < 3; ++i) g.dL_dmeans3D[3 * (size_t)idx + i] = dmean[i];
  g.dL_dmeans2D[3 * (size_t)idx + 0] = dm2x;
  g.dL_dmeans2D[3 * (size_t)idx + 1] = dm2y;
  g.dL_dmeans2D[3 * (size_t)idx + 2] = 0.0f;
  g.dL_dopacities[idx] = dop;
  if (g.stats_xyz_gradient_accum && vis) {
    // densification statistics of the reference's training loop (scene/gaussian_model.py:775-777, train.py:130),
    // fused here because dL_dmeans2D and the radius are in registers: saves the stand-alone kernel's re-read
    g.stats_xyz_gradient_accum[idx] += densify_grad_norm(dm2x, dm2y);
    g.stats_denom[idx] += 1.0f;
    g.stats_max_radii2D[idx] = fmaxf(g.stats_max_radii2D[idx], (float)radii[idx]);
  }
  if (g.dL_dcolors) {
#pragma unroll
    for (int i = 0; i < 3; ++i) g.dL_dcolors[3 * (size_t)idx + i] = dcol[i];
  }
  if (p.cov3D_precomp) {
    if (g.dL_dcov3D)
#pragma unroll
      for (int i = 0; i < 6; ++i) g.dL_dcov3D[6 * (size_t)idx + i] = dcov[i];
  } else {
    if (g.dL_dscales)
#pragma unroll
      for (int i = 0; i < 3; ++i) g.dL_dscales[3 * (size_t)idx + i] = dscale[i];
    if (g.dL_drotations)
#pragma unroll
      for (int i = 0; i < 4; ++i) g.dL_drotations[4 * (size_t)idx + i] = drot[i];
  }
}


void launch_preprocess_fwd(const GsrParams& p, GeomRec* rec, BinInfo* bin, uint32_t* block_sums, uint32_t* block_vis,
                           int32_t* radii, uint32_t* block_big, uint32_t* big_list, uint2* block_range, hipStream_t s) {
  const int nb = (p.P + PRE_BLOCK - 1) / PRE_BLOCK;
  if (nb > 0)
    hipLaunchKernelGGL(preprocess_fwd_kernel, dim3(nb), dim3(PRE_BLOCK), 0, s, p, rec, bin, block_sums, block_vis, radii,
                       block_big, big_list, block_range);
}
void launch_scan_block_sums(const uint32_t* sums_a, uint32_t* offs_a, uint32_t* total_a, const uint32_t* sums_b,
                            uint32_t* offs_b, uint32_t* total_b, int nb, hipStream_t s, uint32_t* host_mirror,
                            const uint2* block_range, const uint32_t* sums_c, uint32_t* offs_c, uint32_t* total_c) {
  // blocks: 0 = a, 1 = b, 2 = fold of block_range, 3 = c
  hipLaunchKernelGGL(scan_block_sums_kernel, dim3(block_range ? (sums_c ? 4 : 3) : (sums_b ? 2 : 1)), dim3(1024), 0, s, sums_a,
                     offs_a, total_a, sums_b, offs_b, total_b, nb, host_mirror, block_range, sums_c, offs_c, total_c);
}
// One wave per listed Gaussian: flags are read 64 at a time -- most are clear, the tiles behind an opaque surface never
// reach the instance -- flagged rows are summed in a fixed lane / iteration order and the total replaces the first row.
// The list is segmented: block b of preprocess_fwd wrote its entries at big_list[b * PRE_BLOCK ..], and entry e of the
// frame lies in the block with big_offs[b] <= e < big_offs[b + 1] (big_offs: the scanned block counts, nb + 1 entries).
// Every wave takes a contiguous stretch of entries (balanced whatever the clustering of the big splats in memory),
// finds the block of its first entry with a 64-way search (three rounds for 23 k blocks) and then walks on: the 64
// offsets it holds in its lanes answer the following entries until the walk leaves them.
__global__ __launch_bounds__(256) void sum_big_rows_kernel(const uint32_t* __restrict__ big_count,
                                                           const uint32_t* __restrict__ big_offs, int nb,
                                                           const uint32_t* __restrict__ big_list,
                                                           const GeomRec* __restrict__ rec,
                                                           const uint32_t* __restrict__ slot_base,
                                                           GradRow* __restrict__ rows, uint8_t* __restrict__ row_flags) {
  const uint32_t count = *big_count;
  const int lane = threadIdx.x & (WAVE - 1);
  const uint32_t wave = blockIdx.x * (256 / WAVE) + threadIdx.x / WAVE, nwaves = gridDim.x * (256 / WAVE);
  const uint32_t per = (count + nwaves - 1) / nwaves;
  const uint32_t e0 = wave * per, e1 = min(count, e0 + per);
  if (e0 >= e1) return;
  // largest b in [0, nb) with big_offs[b] <= e0: narrow [lo, hi] 64 probes at a time
  int lo = 0, hi = nb - 1;
  while (lo < hi) {
    const int span = hi - lo, step = (span + WAVE - 1) / WAVE;             // probes at lo + step, lo + 2 step, ...
    const int pos = lo + (lane + 1) * step;
    const bool le = pos <= hi && big_offs[pos] <= e0;
    const int hits = __popcll(__ballot(le));                              // monotone: the first `hits` probes are <= e0
    const int nlo = lo + hits * step;
    hi = min(hi, nlo + step - 1);
    lo = nlo;
  }
  int win = lo;                                                           // the lanes hold big_offs[win + 1 + lane],
  uint32_t win_off = big_offs[win];                                       // win_off = big_offs[win]
  uint32_t next = win + 1 + lane <= nb ? big_offs[win + 1 + lane] : 0xffffffffu;
  for (uint32_t e = e0; e < e1; ++e) {
    // block of e: the first b >= win with big_offs[b + 1] > e
    unsigned long long m = __ballot(next > e);
    while (m == 0ull) {
      win_off = (uint32_t)__shfl((int)next, WAVE - 1, WAVE);              // big_offs[win + 64]: <= e, so inside the array
      win += WAVE;
      next = win + 1 + lane <= nb ? big_offs[win + 1 + lane] : 0xffffffffu;
      m = __ballot(next > e);
    }
    const int first = __ffsll((long long)m) - 1, b = win + first;
    const uint32_t off_b = first == 0 ? win_off : (uint32_t)__shfl((int)next, first - 1, WAVE);   // big_offs[b], from the lanes
    const uint32_t idx = big_list[(size_t)b * PRE_BLOCK + (e - off_b)];
    const uint2 rr = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(rec + idx) + 48);   // rect_min, rect_wh
    const uint32_t mask = rec[idx].tile_mask;
    const uint32_t n = bin_count(rr.y, mask), slot0 = slot_base[idx];
    float acc[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bool any = false;
    for (uint32_t k = (uint32_t)lane; k < n; k += WAVE) {
      const uint32_t s = slot0 + k;
      if (row_flags[s]) {
        const GradRow q = rows[s];
        any = true;
        acc[0] += q.dmx; acc[1] += q.dmy; acc[2] += q.dcxx; acc[3] += q.dcxy; acc[4] += q.dcyy;
        acc[5] += q.dop; acc[6] += q.dr; acc[7] += q.dg; acc[8] += q.db;
      }
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = wave_reduce_add_f32(acc[i]);
    const bool hit = __ballot(any) != 0ull;
    if (lane == 0) {
      GradRow t;
      t.dmx = acc[0]; t.dmy = acc[1]; t.dcxx = acc[2]; t.dcxy = acc[3]; t.dcyy = acc[4];
      t.dop = acc[5]; t.dr = acc[6]; t.dg = acc[7]; t.db = acc[8];
      rows[slot0] = t;
      row_flags[slot0] = hit ? 1 : 0;
    }
  }
}

#ifndef SUM_BIG_BLOCKS
#define SUM_BIG_BLOCKS 1024   // an entry is a chain of five dependent trips to memory: 4096 waves (heavy-tailed C4 -20 us against 512 blocks; 2048 blocks: -30 us there but +10 us on a frame without big splats)
#endif
void launch_sum_big_rows(const uint32_t* big_count, const uint32_t* big_offs, int nb, const uint32_t* big_list,
                         const GeomRec* rec, const uint32_t* slot_base, GradRow* rows, uint8_t* row_flags, hipStream_t s) {
  if (nb > 0)
    hipLaunchKernelGGL(sum_big_rows_kernel, dim3(SUM_BIG_BLOCKS), dim3(256), 0, s, big_count, big_offs, nb, big_list, rec, slot_base, rows,
                       row_flags);
}
void launch_preprocess_bwd(const GsrParams& p, const int32_t* radii, const GeomRec* rec, const uint32_t* slot_base,
                           const GradRow* rows, const uint8_t* row_flags, const GsrGrads& g, hipStream_t s) {
  const int nb = (p.P + PRE_BLOCK - 1) / PRE_BLOCK;
  if (nb > 0) hipLaunchKernelGGL(preprocess_bwd_kernel, dim3(nb), dim3(PRE_BLOCK), 0, s, p, radii, rec, slot_base, rows, row_flags, g);
}

}  // namespace gsr
