/*
 * Plain-C restatement of the rasterizer forward -- TEST INFRASTRUCTURE ONLY.
 * A second, independent CPU oracle next to oracle/rasterizer_ref.py: scalar loops in upstream's own
 * per-Gaussian / per-pixel order, float32 throughout, compiled with -ffp-contract=off so that the per-Gaussian
 * stage performs the same float32 operation sequence as the PyTorch restatement and the HIP kernels.
 *
 * Parity status: unpinned at the rasterizer boundary (the reference's CUDA source is absent, see
 * oracle/__init__.py); sub-steps with a Python twin in the reference are cited inline.
 *
 *   gsr_oracle_forward(...)  ->  color[3,H,W], radii[P], final_T[H,W], n_contrib[H,W], plus the binning:
 *                                num_rendered and (optionally) sorted keys / point list / tile ranges
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TILE 16

static const float SH_C0 = 0.28209479177387814f, SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                               -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

typedef struct { uint64_t key; uint32_t val; uint32_t seq; } Pair;

static int cmp_pair(const void* a, const void* b) {
  const Pair* x = (const Pair*)a; const Pair* y = (const Pair*)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->seq < y->seq ? -1 : (x->seq > y->seq);      /* emission order: what a stable sort keeps */
}

/* SH -> colour before +0.5 / clamp: utils/sh_utils.py:74-100, sh = [M][3] of one Gaussian */
static void eval_sh(int deg, const float* sh, float x, float y, float z, float* out) {
  for (int ch = 0; ch < 3; ++ch) {
#define S(k) sh[3 * (k) + ch]
    float res = SH_C0 * S(0);
    if (deg > 0) {
      res = res - SH_C1 * y * S(1) + SH_C1 * z * S(2) - SH_C1 * x * S(3);
      if (deg > 1) {
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        res = res + SH_C2[0] * xy * S(4) + SH_C2[1] * yz * S(5) + SH_C2[2] * (2.0f * zz - xx - yy) * S(6) +
              SH_C2[3] * xz * S(7) + SH_C2[4] * (xx - yy) * S(8);
        if (deg > 2)
          res = res + SH_C3[0] * y * (3.0f * xx - yy) * S(9) + SH_C3[1] * xy * z * S(10) +
                SH_C3[2] * y * (4.0f * zz - xx - yy) * S(11) + SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * S(12) +
                SH_C3[4] * x * (4.0f * zz - xx - yy) * S(13) + SH_C3[5] * z * (xx - yy) * S(14) +
                SH_C3[6] * x * (xx - 3.0f * yy) * S(15);
      }
    }
#undef S
    out[ch] = res;
  }
}

static float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* returns num_rendered, or -1 on allocation failure.  Any of keys_out / list_out / ranges_out may be NULL;
 * when given they must hold num_rendered (call once with NULLs to size) / T*2 entries. */
long gsr_oracle_forward(int P, int M, int D, int W, int H, float tanfovx, float tanfovy, float scale_modifier,
                        const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                        const float* scales, const float* rotations, const float* cov3D_precomp,
                        const float* V /* [4][4] row-vector convention */, const float* Mx, const float* campos,
                        const float* bg, float* out_color, int32_t* radii, float* final_T, uint32_t* n_contrib,
                        uint64_t* keys_out, uint32_t* list_out, int64_t* ranges_out) {
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE, T = gx * gy;
  float* xy = (float*)malloc(sizeof(float) * 2 * (size_t)P);
  float* con_o = (float*)malloc(sizeof(float) * 4 * (size_t)P);
  float* rgb = (float*)malloc(sizeof(float) * 3 * (size_t)P);
  float* depth = (float*)malloc(sizeof(float) * (size_t)P);
  int* rect = (int*)malloc(sizeof(int) * 4 * (size_t)P);
  if (!xy || !con_o || !rgb || !depth || !rect) return -1;
  const float fx = (float)W / (2.0f * tanfovx), fy = (float)H / (2.0f * tanfovy);
  const float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
  long R = 0;
  /* ---- A.2 preprocess ------------------------------------------------------------------------- */
  for (int i = 0; i < P; ++i) {
    radii[i] = 0;
    rect[4 * i] = rect[4 * i + 1] = rect[4 * i + 2] = rect[4 * i + 3] = 0;
    const float px = means3D[3 * i], py = means3D[3 * i + 1], pz = means3D[3 * i + 2];
    const float vx = V[0] * px + V[4] * py + V[8] * pz + V[12];
    const float vy = V[1] * px + V[5] * py + V[9] * pz + V[13];
    const float vz = V[2] * px + V[6] * py + V[10] * pz + V[14];
    if (!(vz > 0.2f)) continue;
    const float hx = Mx[0] * px + Mx[4] * py + Mx[8] * pz + Mx[12];
    const float hy = Mx[1] * px + Mx[5] * py + Mx[9] * pz + Mx[13];
    const float hw = Mx[3] * px + Mx[7] * py + Mx[11] * pz + Mx[15];
    const float p_w = 1.0f / (hw + 0.0000001f);
    const float ndc_x = hx * p_w, ndc_y = hy * p_w;
    float cov[6];
    if (cov3D_precomp) {
      memcpy(cov, cov3D_precomp + 6 * (size_t)i, sizeof(cov));
    } else {   /* scene/gaussian_model.py:28-32, utils/general_utils.py:64-110; quaternion as passed */
      const float s0 = scale_modifier * scales[3 * i], s1 = scale_modifier * scales[3 * i + 1],
                  s2 = scale_modifier * scales[3 * i + 2];
      const float r = rotations[4 * i], x = rotations[4 * i + 1], y = rotations[4 * i + 2], z = rotations[4 * i + 3];
      const float R00 = 1.0f - 2.0f * (y * y + z * z), R01 = 2.0f * (x * y - r * z), R02 = 2.0f * (x * z + r * y);
      const float R10 = 2.0f * (x * y + r * z), R11 = 1.0f - 2.0f * (x * x + z * z), R12 = 2.0f * (y * z - r * x);
      const float R20 = 2.0f * (x * z - r * y), R21 = 2.0f * (y * z + r * x), R22 = 1.0f - 2.0f * (x * x + y * y);
      const float L00 = R00 * s0, L01 = R01 * s1, L02 = R02 * s2, L10 = R10 * s0, L11 = R11 * s1, L12 = R12 * s2,
                  L20 = R20 * s0, L21 = R21 * s1, L22 = R22 * s2;
      cov[0] = L00 * L00 + L01 * L01 + L02 * L02; cov[1] = L00 * L10 + L01 * L11 + L02 * L12;
      cov[2] = L00 * L20 + L01 * L21 + L02 * L22; cov[3] = L10 * L10 + L11 * L11 + L12 * L12;
      cov[4] = L10 * L20 + L11 * L21 + L12 * L22; cov[5] = L20 * L20 + L21 * L21 + L22 * L22;
    }
    const float txtz = vx / vz, tytz = vy / vz;
    const float tx = fminf(limx, fmaxf(-limx, txtz)) * vz, ty = fminf(limy, fmaxf(-limy, tytz)) * vz;
    const float j00 = fx / vz, j02 = -(fx * tx) / (vz * vz), j11 = fy / vz, j12 = -(fy * ty) / (vz * vz);
    float A0[3], A1[3], B0[3], B1[3];
    for (int j = 0; j < 3; ++j) {
      A0[j] = j00 * V[4 * j + 0] + j02 * V[4 * j + 2];
      A1[j] = j11 * V[4 * j + 1] + j12 * V[4 * j + 2];
    }
    const float S[3][3] = {{cov[0], cov[1], cov[2]}, {cov[1], cov[3], cov[4]}, {cov[2], cov[4], cov[5]}};
    for (int j = 0; j < 3; ++j) {
      B0[j] = A0[0] * S[0][j] + A0[1] * S[1][j] + A0[2] * S[2][j];
      B1[j] = A1[0] * S[0][j] + A1[1] * S[1][j] + A1[2] * S[2][j];
    }
    const float a = (B0[0] * A0[0] + B0[1] * A0[1] + B0[2] * A0[2]) + 0.3f;
    const float b = B0[0] * A1[0] + B0[1] * A1[1] + B0[2] * A1[2];
    const float c = (B1[0] * A1[0] + B1[1] * A1[1] + B1[2] * A1[2]) + 0.3f;
    const float det = a * c - b * b;
    if (det == 0.0f) continue;
    const float det_inv = 1.0f / det;
    const float mid = 0.5f * (a + c);
    const float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
    const float rad = ceilf(3.0f * sqrtf(fmaxf(mid + sq, mid - sq)));
    const float mx = ((ndc_x + 1.0f) * (float)W - 1.0f) * 0.5f, my = ((ndc_y + 1.0f) * (float)H - 1.0f) * 0.5f;
    if (!isfinite(rad) || !isfinite(mx) || !isfinite(my)) continue;
    const int x0 = (int)clampf(truncf((mx - rad) / (float)TILE), 0.0f, (float)gx);
    const int y0 = (int)clampf(truncf((my - rad) / (float)TILE), 0.0f, (float)gy);
    const int x1 = (int)clampf(truncf((mx + rad + (float)(TILE - 1)) / (float)TILE), 0.0f, (float)gx);
    const int y1 = (int)clampf(truncf((my + rad + (float)(TILE - 1)) / (float)TILE), 0.0f, (float)gy);
    if ((x1 - x0) * (y1 - y0) == 0) continue;
    float col[3];
    if (colors_precomp) {
      col[0] = colors_precomp[3 * i]; col[1] = colors_precomp[3 * i + 1]; col[2] = colors_precomp[3 * i + 2];
    } else {   /* gaussian_renderer/__init__.py:80-84 */
      const float dx = px - campos[0], dy = py - campos[1], dz = pz - campos[2];
      const float ln = sqrtf(dx * dx + dy * dy + dz * dz);
      eval_sh(D, shs + (size_t)i * M * 3, dx / ln, dy / ln, dz / ln, col);
      for (int ch = 0; ch < 3; ++ch) col[ch] = fmaxf(col[ch] + 0.5f, 0.0f);
    }
    radii[i] = (int32_t)rad;
    depth[i] = vz;
    xy[2 * i] = mx; xy[2 * i + 1] = my;
    con_o[4 * i] = c * det_inv; con_o[4 * i + 1] = -b * det_inv; con_o[4 * i + 2] = a * det_inv; con_o[4 * i + 3] = opacities[i];
    rgb[3 * i] = col[0]; rgb[3 * i + 1] = col[1]; rgb[3 * i + 2] = col[2];
    rect[4 * i] = x0; rect[4 * i + 1] = y0; rect[4 * i + 2] = x1; rect[4 * i + 3] = y1;
    R += (long)(x1 - x0) * (y1 - y0);
  }
  /* ---- A.3 binning -------------------------------------------------------------------------------- */
  Pair* pairs = (Pair*)malloc(sizeof(Pair) * (size_t)(R > 0 ? R : 1));
  int64_t* ranges = (int64_t*)calloc((size_t)T * 2, sizeof(int64_t));
  if (!pairs || !ranges) return -1;
  long o = 0;
  for (int i = 0; i < P; ++i) {
    if (radii[i] <= 0) continue;
    uint32_t dbits; memcpy(&dbits, &depth[i], 4);
    for (int y = rect[4 * i + 1]; y < rect[4 * i + 3]; ++y)
      for (int x = rect[4 * i]; x < rect[4 * i + 2]; ++x) {
        pairs[o].key = ((uint64_t)(y * gx + x) << 32) | dbits;
        pairs[o].val = (uint32_t)i;
        pairs[o].seq = (uint32_t)o;
        ++o;
      }
  }
  qsort(pairs, (size_t)R, sizeof(Pair), cmp_pair);
  for (long k = 0; k < R; ++k) {
    const int64_t t = (int64_t)(pairs[k].key >> 32);
    if (k == 0 || (int64_t)(pairs[k - 1].key >> 32) != t) ranges[2 * t] = k;
    ranges[2 * t + 1] = k + 1;
  }
  if (keys_out) for (long k = 0; k < R; ++k) keys_out[k] = pairs[k].key;
  if (list_out) for (long k = 0; k < R; ++k) list_out[k] = pairs[k].val;
  if (ranges_out) memcpy(ranges_out, ranges, sizeof(int64_t) * 2 * (size_t)T);
  /* ---- A.4 compositing, one pixel at a time ------------------------------------------------------------ */
  const size_t HW = (size_t)W * H;
  for (int py = 0; py < H; ++py)
    for (int px = 0; px < W; ++px) {
      const int t = (py / TILE) * gx + (px / TILE);
      const float pxf = (float)px, pyf = (float)py;
      float Tt = 1.0f, C[3] = {0.f, 0.f, 0.f};
      uint32_t contributor = 0, last = 0;
      for (int64_t k = ranges[2 * t]; k < ranges[2 * t + 1]; ++k) {
        const uint32_t g = pairs[k].val;
        ++contributor;
        const float dx = xy[2 * g] - pxf, dy = xy[2 * g + 1] - pyf;
        const float power = -0.5f * (con_o[4 * g] * dx * dx + con_o[4 * g + 2] * dy * dy) - con_o[4 * g + 1] * dx * dy;
        if (power > 0.0f) continue;
        const float alpha = fminf(0.99f, con_o[4 * g + 3] * expf(power));
        if (alpha < 1.0f / 255.0f) continue;
        const float test_T = Tt * (1.0f - alpha);
        if (test_T < 0.0001f) break;
        for (int ch = 0; ch < 3; ++ch) C[ch] += rgb[3 * g + ch] * alpha * Tt;
        Tt = test_T;
        last = contributor;
      }
      const size_t pix = (size_t)py * W + px;
      for (int ch = 0; ch < 3; ++ch) out_color[ch * HW + pix] = C[ch] + Tt * bg[ch];
      if (final_T) final_T[pix] = Tt;
      if (n_contrib) n_contrib[pix] = last;
    }
  free(xy); free(con_o); free(rgb); free(depth); free(rect); free(pairs); free(ranges);
  return R;
}
