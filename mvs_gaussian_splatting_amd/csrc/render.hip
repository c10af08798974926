// Per-tile alpha compositing, forward and backward (§8 a10, A.5).
//
// CDNA4 mapping (not the upstream 16x16-thread block): ONE 64-lane wavefront owns one 16x16
// tile and every lane carries four pixels, one in each 8x8 sub-block, so that
//   * the per-Gaussian record is fetched from LDS once per wave (broadcast read) and amortised
//     over 256 pixel evaluations instead of 64;
//   * no workgroup barrier exists at all (a workgroup is a single wave);
//   * a sub-block whose 64 pixels cannot reach alpha >= 1/255 for a Gaussian is skipped with a
//     scalar branch: the lane that stages the Gaussian tests the conservative extent of the
//     alpha >= 1/255 ellipse (GeomRec.ext_x/ext_y) against the four sub-blocks, the wave
//     ballots the result, and only set bits are visited.  Skipped evaluations would have been
//     rejected by the alpha test anyway, so results are unchanged.
// Per-instance gradients are reduced across the wave in registers (DPP) and written as one
// row per (Gaussian, tile) instance -- no atomics; preprocess_bwd sums the rows per Gaussian.
#include "gsr_common.h"
#include "gsr_launch.h"

namespace gsr {

constexpr int BATCH = WAVE;   // instances staged per round

struct Staged {               // what one lane fetches for the instance it stages
  float4 q0, q1, q2;          // GeomRec words 0..11
  uint32_t rect_min, rect_wh;
};

__device__ inline void load_staged(const GeomRec* __restrict__ rec, uint32_t id, Staged& s) {
  const float4* r = reinterpret_cast<const float4*>(rec + id);
  s.q0 = r[0];
  s.q1 = r[1];
  s.q2 = r[2];
}

// which of the tile's four 8x8 sub-blocks can the Gaussian reach (bit k = sub-block k)
__device__ inline uint32_t subblock_mask(float gx, float gy, float ex, float ey, float tx0, float ty0) {
  if (ex < 0.0f) return 0u;
  const bool xl = (gx + ex >= tx0) && (gx - ex <= tx0 + 7.0f);
  const bool xr = (gx + ex >= tx0 + 8.0f) && (gx - ex <= tx0 + 15.0f);
  const bool yt = (gy + ey >= ty0) && (gy - ey <= ty0 + 7.0f);
  const bool yb = (gy + ey >= ty0 + 8.0f) && (gy - ey <= ty0 + 15.0f);
  return (uint32_t)(xl && yt) | ((uint32_t)(xr && yt) << 1) | ((uint32_t)(xl && yb) << 2) |
         ((uint32_t)(xr && yb) << 3);
}

__global__ __launch_bounds__(WAVE) void render_fwd_kernel(int W, int H, int grid_x,
                                                          const uint2* __restrict__ ranges,
                                                          const uint32_t* __restrict__ point_list,
                                                          const GeomRec* __restrict__ rec,
                                                          const float* __restrict__ bg,
                                                          float* __restrict__ out_color,
                                                          float* __restrict__ final_T,
                                                          uint32_t* __restrict__ n_contrib,
                                                          uint32_t* __restrict__ tile_max) {
  __shared__ float4 sA[2][BATCH];
  __shared__ float4 sB[2][BATCH];
  __shared__ float sC[2][BATCH];

  const int tile = blockIdx.x;
  const int lane = threadIdx.x;
  const int tile_x = tile % grid_x, tile_y = tile / grid_x;
  const int px0 = tile_x * TILE + (lane & 7), py0 = tile_y * TILE + (lane >> 3);
  const float tx0 = (float)(tile_x * TILE), ty0 = (float)(tile_y * TILE);

  float pxf[4], pyf[4], T[4], Cr[4], Cg[4], Cb[4];
  uint32_t last[4];
  bool done[4], inside[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int px = px0 + 8 * (k & 1), py = py0 + 8 * (k >> 1);
    pxf[k] = (float)px;
    pyf[k] = (float)py;
    inside[k] = px < W && py < H;
    done[k] = !inside[k];
    T[k] = 1.0f;
    Cr[k] = Cg[k] = Cb[k] = 0.0f;
    last[k] = 0;
  }

  const uint2 range = ranges[tile];
  const uint32_t start = range.x, end = range.y;
  bool all_done = __all(done[0] && done[1] && done[2] && done[3]);

  // software pipeline: ids two rounds ahead, records one round ahead
  uint32_t id_next = 0;
  Staged st;
  st.q0 = st.q1 = st.q2 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (start + lane < end) {
    const uint32_t id = point_list[start + lane];
    load_staged(rec, id, st);
  }
  if (start + BATCH + lane < end) id_next = point_list[start + BATCH + lane];

  int buf = 0;
  for (uint32_t pos = start; pos < end && !all_done; pos += BATCH) {
    const bool have = pos + lane < end;
    uint32_t m = have ? subblock_mask(st.q0.x, st.q0.y, st.q2.y, st.q2.z, tx0, ty0) : 0u;
    sA[buf][lane] = st.q0;
    sB[buf][lane] = st.q1;
    sC[buf][lane] = st.q2.x;
    __syncthreads();   // single-wave workgroup: orders the LDS writes before the broadcast reads
    // prefetch the next round while this one is composited
    if (pos + BATCH + lane < end) load_staged(rec, id_next, st);
    if (pos + 2 * BATCH + lane < end) id_next = point_list[pos + 2 * BATCH + lane];

    unsigned long long nz = __ballot(m != 0u);
    while (nz) {
      const int j = __ffsll((long long)nz) - 1;
      nz &= nz - 1;
      const uint32_t mj = (uint32_t)__builtin_amdgcn_readlane((int)m, j);
      const float4 a = sA[buf][j];
      const float4 b = sB[buf][j];
      const float cb = sC[buf][j];
      const uint32_t pos1 = pos - start + (uint32_t)j + 1u;   // 1-based contributor index
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (mj & (1u << k)) {
          const float dx = a.x - pxf[k], dy = a.y - pyf[k];
          const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
          const float alpha = fminf(ALPHA_MAX, b.y * __expf(power));
          const bool ok = !done[k] && power <= 0.0f && alpha >= ALPHA_MIN;
          const float test_T = T[k] * (1.0f - alpha);
          const bool stop = ok && test_T < T_STOP;
          done[k] = done[k] || stop;
          if (ok && !stop) {
            const float w = alpha * T[k];
            Cr[k] += b.z * w;
            Cg[k] += b.w * w;
            Cb[k] += cb * w;
            T[k] = test_T;
            last[k] = pos1;
          }
        }
      }
      if (__all(done[0] && done[1] && done[2] && done[3])) {
        all_done = true;
        break;
      }
    }
    buf ^= 1;
  }

  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
  const size_t HW = (size_t)W * H;
  uint32_t mx = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (inside[k]) {
      const size_t pix = (size_t)(py0 + 8 * (k >> 1)) * W + (px0 + 8 * (k & 1));
      out_color[pix] = Cr[k] + T[k] * bg0;
      out_color[HW + pix] = Cg[k] + T[k] * bg1;
      out_color[2 * HW + pix] = Cb[k] + T[k] * bg2;
      final_T[pix] = T[k];
      n_contrib[pix] = last[k];
      mx = max(mx, last[k]);
    }
  }
#pragma unroll
  for (int d = WAVE / 2; d > 0; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, WAVE));
  if (lane == 0) tile_max[tile] = mx;
}

// ---- wave-wide sum with DPP (result valid in lane 63) ---------------------------------------
template <int CTRL, int ROW_MASK>
__device__ inline float dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
  return v + __int_as_float(moved);
}
__device__ inline float wave_sum_lane63(float v) {
  v = dpp_add<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
  v = dpp_add<0x141, 0xf>(v);   // row_half_mirror
  v = dpp_add<0x140, 0xf>(v);   // row_mirror  -> every lane holds its row's sum
  v = dpp_add<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
  return v;
}

__global__ __launch_bounds__(WAVE) void render_bwd_kernel(int W, int H, int grid_x,
                                                          const uint2* __restrict__ ranges,
                                                          const uint32_t* __restrict__ point_list,
                                                          const GeomRec* __restrict__ rec,
                                                          const float* __restrict__ bg,
                                                          const float* __restrict__ final_T,
                                                          const uint32_t* __restrict__ n_contrib,
                                                          const uint32_t* __restrict__ tile_max,
                                                          const float* __restrict__ dL_dpix,
                                                          GradRow* __restrict__ rows,
                                                          uint8_t* __restrict__ row_flags) {
  __shared__ float4 sA[2][BATCH];
  __shared__ float4 sB[2][BATCH];
  __shared__ float sC[2][BATCH];

  const int tile = blockIdx.x;
  const int lane = threadIdx.x;
  const int tile_x = tile % grid_x, tile_y = tile / grid_x;
  const int px0 = tile_x * TILE + (lane & 7), py0 = tile_y * TILE + (lane >> 3);
  const float tx0 = (float)(tile_x * TILE), ty0 = (float)(tile_y * TILE);
  const size_t HW = (size_t)W * H;
  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
  const float half_w = 0.5f * (float)W, half_h = 0.5f * (float)H;

  float pxf[4], pyf[4], T[4], Tfin[4], dpr[4], dpg[4], dpb[4], bgdot[4];
  float acc_r[4], acc_g[4], acc_b[4], last_a[4], last_r[4], last_g[4], last_b[4];
  uint32_t last[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int px = px0 + 8 * (k & 1), py = py0 + 8 * (k >> 1);
    pxf[k] = (float)px;
    pyf[k] = (float)py;
    const bool in = px < W && py < H;
    const size_t pix = (size_t)py * W + px;
    Tfin[k] = in ? final_T[pix] : 0.0f;
    T[k] = Tfin[k];
    last[k] = in ? n_contrib[pix] : 0u;
    dpr[k] = in ? dL_dpix[pix] : 0.0f;
    dpg[k] = in ? dL_dpix[HW + pix] : 0.0f;
    dpb[k] = in ? dL_dpix[2 * HW + pix] : 0.0f;
    bgdot[k] = bg0 * dpr[k] + bg1 * dpg[k] + bg2 * dpb[k];
    acc_r[k] = acc_g[k] = acc_b[k] = 0.0f;
    last_a[k] = last_r[k] = last_g[k] = last_b[k] = 0.0f;
  }

  const uint2 range = ranges[tile];
  const uint32_t start = range.x;
  uint32_t hi = min(range.y - range.x, tile_max[tile]);   // instances past the last contributor got no gradient

  Staged st;
  st.q0 = st.q1 = st.q2 = make_float4(0.f, 0.f, 0.f, 0.f);
  st.rect_min = st.rect_wh = 0;
  auto stage = [&](uint32_t lo, uint32_t top) {
    if (lo + lane < top) {
      const uint32_t id = point_list[start + lo + lane];
      load_staged(rec, id, st);
      const uint2 rr = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(rec + id) + 48);
      st.rect_min = rr.x;
      st.rect_wh = rr.y;
    }
  };
  uint32_t lo = hi > BATCH ? hi - BATCH : 0u;
  stage(lo, hi);

  int buf = 0;
  while (hi > 0) {
    const bool have = lo + lane < hi;
    const uint32_t m = have ? subblock_mask(st.q0.x, st.q0.y, st.q2.y, st.q2.z, tx0, ty0) : 0u;
    // slot of this instance in the unsorted instance array (the order duplicateWithKeys emitted)
    const uint32_t rw = st.rect_wh & 0xffffu;
    const uint32_t slot = __float_as_uint(st.q2.w) + ((uint32_t)tile_y - (st.rect_min >> 16)) * rw +
                          ((uint32_t)tile_x - (st.rect_min & 0xffffu));
    sA[buf][lane] = st.q0;
    sB[buf][lane] = st.q1;
    sC[buf][lane] = st.q2.x;
    __syncthreads();
    const uint32_t cur_lo = lo;
    hi = lo;
    lo = hi > BATCH ? hi - BATCH : 0u;
    if (hi > 0) stage(lo, hi);   // prefetch the next (earlier) round

    unsigned long long nz = __ballot(m != 0u);
    while (nz) {
      const int j = 63 - __clzll((long long)nz);   // back to front
      nz &= ~(1ull << j);
      const uint32_t mj = (uint32_t)__builtin_amdgcn_readlane((int)m, j);
      const float4 a = sA[buf][j];
      const float4 b = sB[buf][j];
      const float cb = sC[buf][j];
      const uint32_t pos1 = cur_lo + (uint32_t)j + 1u;
      float g_mx = 0.f, g_my = 0.f, g_cxx = 0.f, g_cxy = 0.f, g_cyy = 0.f, g_op = 0.f, g_r = 0.f, g_g = 0.f,
            g_b = 0.f;
      bool any = false;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (mj & (1u << k)) {
          const float dx = a.x - pxf[k], dy = a.y - pyf[k];
          const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
          const float G = __expf(power);
          const float alpha = fminf(ALPHA_MAX, b.y * G);
          const bool ok = pos1 <= last[k] && power <= 0.0f && alpha >= ALPHA_MIN;
          if (ok) {
            any = true;
            const float one_m = 1.0f - alpha;
            T[k] = T[k] / one_m;
            const float dch = alpha * T[k];
            acc_r[k] = last_a[k] * last_r[k] + (1.0f - last_a[k]) * acc_r[k];
            acc_g[k] = last_a[k] * last_g[k] + (1.0f - last_a[k]) * acc_g[k];
            acc_b[k] = last_a[k] * last_b[k] + (1.0f - last_a[k]) * acc_b[k];
            last_r[k] = b.z; last_g[k] = b.w; last_b[k] = cb; last_a[k] = alpha;
            float dL_dalpha = (b.z - acc_r[k]) * dpr[k] + (b.w - acc_g[k]) * dpg[k] + (cb - acc_b[k]) * dpb[k];
            g_r += dch * dpr[k];
            g_g += dch * dpg[k];
            g_b += dch * dpb[k];
            dL_dalpha *= T[k];
            dL_dalpha += (-Tfin[k] / one_m) * bgdot[k];
            const float dL_dG = b.y * dL_dalpha;
            const float gdx = G * dx, gdy = G * dy;
            const float dG_ddelx = -gdx * a.z - gdy * a.w;
            const float dG_ddely = -gdy * b.x - gdx * a.w;
            g_mx += dL_dG * dG_ddelx * half_w;
            g_my += dL_dG * dG_ddely * half_h;
            g_cxx += -0.5f * gdx * dx * dL_dG;
            g_cxy += -gdx * dy * dL_dG;          // true derivative (upstream keeps half and doubles later)
            g_cyy += -0.5f * gdy * dy * dL_dG;
            g_op += G * dL_dalpha;
          }
        }
      }
      if (__any(any)) {
        g_mx = wave_sum_lane63(g_mx);
        g_my = wave_sum_lane63(g_my);
        g_cxx = wave_sum_lane63(g_cxx);
        g_cxy = wave_sum_lane63(g_cxy);
        g_cyy = wave_sum_lane63(g_cyy);
        g_op = wave_sum_lane63(g_op);
        g_r = wave_sum_lane63(g_r);
        g_g = wave_sum_lane63(g_g);
        g_b = wave_sum_lane63(g_b);
        const uint32_t sj = (uint32_t)__builtin_amdgcn_readlane((int)slot, j);
        if (lane == WAVE - 1) {
          float4* dst = reinterpret_cast<float4*>(rows + sj);
          dst[0] = make_float4(g_mx, g_my, g_cxx, g_cxy);
          dst[1] = make_float4(g_cyy, g_op, g_r, g_g);
          dst[2] = make_float4(g_b, 0.f, 0.f, 0.f);
          row_flags[sj] = 1;
        }
      }
    }
    buf ^= 1;
  }
}


void launch_render_fwd(int W, int H, const uint2* ranges, const uint32_t* point_list, const GeomRec* rec,
                       const float* bg, float* out_color, float* final_T, uint32_t* n_contrib, uint32_t* tile_max,
                       hipStream_t s) {
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  hipLaunchKernelGGL(render_fwd_kernel, dim3(gx * gy), dim3(WAVE), 0, s, W, H, gx, ranges, point_list, rec, bg,
                     out_color, final_T, n_contrib, tile_max);
}
void launch_render_bwd(int W, int H, const uint2* ranges, const uint32_t* point_list, const GeomRec* rec,
                       const float* bg, const float* final_T, const uint32_t* n_contrib, const uint32_t* tile_max,
                       const float* dL_dpix, GradRow* rows, uint8_t* row_flags, hipStream_t s) {
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  hipLaunchKernelGGL(render_bwd_kernel, dim3(gx * gy), dim3(WAVE), 0, s, W, H, gx, ranges, point_list, rec, bg,
                     final_T, n_contrib, tile_max, dL_dpix, rows, row_flags);
}

}  // namespace gsr
