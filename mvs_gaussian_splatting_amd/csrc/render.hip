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
#include <stdlib.h>

namespace gsr {

constexpr int BATCH = WAVE;   // instances staged per round
// Tiles are independent and each is owned by one wave; four of them share a 256-lane workgroup only to
// reach 8 waves per SIMD (a CU holds fewer single-wave workgroups than waves).  No workgroup barrier is
// used: each wave has its own LDS slice, and LDS operations of one wave execute in order.
constexpr int WAVES_PER_BLOCK = 4;
constexpr float LOG2E = 1.4426950408889634f;

// What one lane fetches for the instance it stages (GeomRec words 0..11 and, for the backward, 12..13).
struct Staged {
  float4 q0, q1, q2;
  uint32_t rect_min, rect_wh, slot_base;
};

// FULL = false skips word 11 (tile_mask), which the forward never reads: a dead destination register
// of an in-flight load gets recycled by the compiler and forces an early s_waitcnt vmcnt.
template <bool FULL>
__device__ inline void load_staged(const GeomRec* __restrict__ rec, uint32_t id, Staged& s) {
  const float4* r = reinterpret_cast<const float4*>(rec + id);
  s.q0 = r[0];
  s.q1 = r[1];
  if (FULL) {
    s.q2 = r[2];
  } else {
    const float* f = reinterpret_cast<const float*>(r + 2);
    s.q2.x = f[0];
    s.q2.y = f[1];
    s.q2.z = f[2];
  }
}

// Which of the tile's four 8x8 sub-blocks can the Gaussian reach with alpha >= 1/255 (bit k = sub-block k).
// Two conservative tests: the bounding box of the alpha >= 1/255 ellipse (GeomRec.ext_x / ext_y), then the exact
// ellipse-vs-rectangle test q_min <= 2 ln(255 opacity) with a relative + absolute safety margin.  A skipped
// sub-block would have been rejected pixel by pixel by the alpha test, so results do not change.
__device__ inline uint32_t subblock_mask(float gx, float gy, float ex, float ey, float cxx, float cxy, float cyy,
                                         float opacity, float tx0, float ty0) {
  if (ex < 0.0f) return 0u;
  const bool xl = (gx + ex >= tx0) && (gx - ex <= tx0 + 7.0f);
  const bool xr = (gx + ex >= tx0 + 8.0f) && (gx - ex <= tx0 + 15.0f);
  const bool yt = (gy + ey >= ty0) && (gy - ey <= ty0 + 7.0f);
  const bool yb = (gy + ey >= ty0 + 8.0f) && (gy - ey <= ty0 + 15.0f);
  uint32_t m = (uint32_t)(xl && yt) | ((uint32_t)(xr && yt) << 1) | ((uint32_t)(xl && yb) << 2) |
               ((uint32_t)(xr && yb) << 3);
  if (m == 0u) return 0u;
  const float t = 2.0f * 0.6931472f * __log2f(255.0f * opacity) * 1.001f + 2e-3f;
  const float icxx = __builtin_amdgcn_rcpf(cxx), icyy = __builtin_amdgcn_rcpf(cyy);
  const float ax0 = tx0 - gx, ay0 = ty0 - gy;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float dx0 = ax0 + 8.0f * (float)(k & 1), dy0 = ay0 + 8.0f * (float)(k >> 1);
    if (qmin_rect(cxx, cxy, cyy, icxx, icyy, dx0, dx0 + 7.0f, dy0, dy0 + 7.0f) > t) m &= ~(1u << k);
  }
  return m;
}

// LDS image of a staged instance.  The quadratic form is kept pre-scaled by log2(e) so that the
// exponential is a bare v_exp_f32:  log2e*power = dx*(aq*dx + bq*dy) + cq*dy*dy  with
// aq = -0.5*log2e*cxx, bq = -log2e*cxy, cq = -0.5*log2e*cyy.
struct LdsRec {
  float4 A;   // x, y, aq, bq
  float4 B;   // cq, opacity, r, g
};
__device__ inline void make_lds(const Staged& st, LdsRec& o) {
  o.A = make_float4(st.q0.x, st.q0.y, (-0.5f * LOG2E) * st.q0.z, -LOG2E * st.q0.w);
  o.B = make_float4((-0.5f * LOG2E) * st.q1.x, st.q1.y, st.q1.z, st.q1.w);
}

// Per-lane pixel state: T > 0 while the pixel is live.  A pixel that saturates keeps its final
// transmittance with the sign flipped (T < 0; outside the image: T = 0), so every later
// test_T = T*(1-alpha) <= 0 < 1e-4 keeps it out of the blend without a separate flag.
template <bool STATS, bool TRACK>
__device__ __forceinline__ void render_fwd_tile(const int tile, float4* sA, float4* sB, float* sC, int W, int H,
                                                int grid_x, const uint2* __restrict__ ranges,
                                                const uint32_t* __restrict__ point_list,
                                                const GeomRec* __restrict__ rec, const float* __restrict__ bg,
                                                float* __restrict__ out_color, float* __restrict__ final_T,
                                                uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ tile_max,
                                                unsigned long long* __restrict__ stats) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int tile_x = tile % grid_x, tile_y = tile / grid_x;
  const int px0 = tile_x * TILE + (lane & 7), py0 = tile_y * TILE + (lane >> 3);
  const float tx0 = (float)(tile_x * TILE), ty0 = (float)(tile_y * TILE);
  float pxf0 = (float)px0, pyf0 = (float)py0;
  asm volatile("" : "+v"(pxf0), "+v"(pyf0));   // keep them in registers (no per-visit re-conversion)

  float T[4], Cr[4], Cg[4], Cb[4];
  uint32_t last[4];
  uint32_t live = 0;   // wave-uniform: bit k set while sub-block k still has a live pixel
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool in = (px0 + 8 * (k & 1)) < W && (py0 + 8 * (k >> 1)) < H;
    T[k] = in ? 1.0f : 0.0f;
    Cr[k] = Cg[k] = Cb[k] = 0.0f;
    last[k] = 0;
    if (__builtin_amdgcn_ballot_w64(in) != 0ull) live |= 1u << k;
  }

  const uint2 range = ranges[tile];
  const uint32_t start = range.x, end = range.y;

  // software pipeline: ids two rounds ahead, records one round ahead.  Every prefetch is issued
  // unconditionally with a clamped index (lanes past the end re-read the last instance and are masked
  // by `have`): a conditional overwrite of `st` makes the compiler copy registers right behind the load
  // and wait for it on the spot.
  uint32_t id_next = 0;
  Staged st;
  st.q0 = st.q1 = st.q2 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (start >= end) live = 0;
  if (live) {
    load_staged<false>(rec, point_list[min(start + lane, end - 1)], st);
    id_next = point_list[min(start + BATCH + lane, end - 1)];
  }

  uint32_t st_staged = 0, st_visited = 0, st_evals = 0, st_hits = 0;
  for (uint32_t pos = start; pos < end && live; pos += BATCH) {
    const bool have = pos + lane < end;
    if (STATS) st_staged += min((uint32_t)BATCH, end - pos);
    const uint32_t m = have ? subblock_mask(st.q0.x, st.q0.y, st.q2.y, st.q2.z, st.q0.z, st.q0.w, st.q1.x, st.q1.y, tx0, ty0) : 0u;
    LdsRec lr;
    make_lds(st, lr);
    __builtin_amdgcn_wave_barrier();   // LDS ops of one wave execute in order: no hardware barrier needed
    sA[lane] = lr.A;
    sB[lane] = lr.B;
    sC[lane] = st.q2.x;
    __builtin_amdgcn_wave_barrier();
    // prefetch the next round while this one is composited
    load_staged<false>(rec, id_next, st);
    id_next = point_list[min(pos + 2 * BATCH + lane, end - 1)];

    // sub-blocks whose pixels are all parked need no further work; re-derived once per round
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (__builtin_amdgcn_ballot_w64(T[k] > 0.0f) == 0ull) live &= ~(1u << k);
    if (live == 0u) break;

    unsigned long long nz = __ballot(m != 0u);
    while (nz) {
      const int j = __ffsll((long long)nz) - 1;
      nz &= nz - 1;
      const uint32_t mj = (uint32_t)__builtin_amdgcn_readlane((int)m, j) & live;
      if (mj == 0u) continue;
      const float4 a = sA[j];
      const float4 b = sB[j];
      const float cb = sC[j];
      const uint32_t pos1 = pos - start + (uint32_t)j + 1u;   // 1-based contributor index
      if (STATS) ++st_visited;
      const float dx0 = a.x - pxf0, dy0 = a.y - pyf0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (mj & (1u << k)) {
          const float dx = (k & 1) ? dx0 - 8.0f : dx0;
          const float dy = (k >> 1) ? dy0 - 8.0f : dy0;
          const float t = fmaf(a.w, dy, a.z * dx);
          const float p2 = fmaf(dx, t, (b.x * dy) * dy);
          const float alpha = fminf(ALPHA_MAX, b.y * __builtin_amdgcn_exp2f(p2));
          const bool ok = (p2 <= 0.0f) && (alpha >= ALPHA_MIN);
          const float test_T = T[k] * (1.0f - alpha);
          const bool go = ok && !(test_T < T_STOP);
          if (STATS) { ++st_evals; if (__ballot(ok && T[k] > 0.0f) != 0ull) ++st_hits; }
          // ok && !go: the pixel saturates here (or is already parked): park it with the sign flipped
          const float parked = ok ? -fabsf(T[k]) : T[k];
          if (go) {
            const float w = alpha * T[k];
            Cr[k] = fmaf(b.z, w, Cr[k]);
            Cg[k] = fmaf(b.w, w, Cg[k]);
            Cb[k] = fmaf(cb, w, Cb[k]);
            if (TRACK) last[k] = pos1;
          }
          T[k] = go ? test_T : parked;
        }
      }
    }
  }

  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
  const size_t HW = (size_t)W * H;
  uint32_t mx = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int px = px0 + 8 * (k & 1), py = py0 + 8 * (k >> 1);
    if (px < W && py < H) {
      const float Tf = fabsf(T[k]);
      const size_t pix = (size_t)py * W + px;
      out_color[pix] = Cr[k] + Tf * bg0;
      out_color[HW + pix] = Cg[k] + Tf * bg1;
      out_color[2 * HW + pix] = Cb[k] + Tf * bg2;
      if (TRACK) {
        final_T[pix] = Tf;
        n_contrib[pix] = last[k];
        mx = max(mx, last[k]);
      }
    }
  }
  if (TRACK) {
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, WAVE));
    if (lane == 0) tile_max[tile] = mx;
  }
  if (STATS && lane == 0) {
    atomicAdd(&stats[0], (unsigned long long)(end - start));
    atomicAdd(&stats[1], (unsigned long long)st_staged);
    atomicAdd(&stats[2], (unsigned long long)st_visited);
    atomicAdd(&stats[3], (unsigned long long)st_evals);
    atomicAdd(&stats[4], (unsigned long long)st_hits);
    atomicAdd(&stats[5], (unsigned long long)mx);
  }
}

// One wave per tile; blockIdx walks the tiles longest-list-first (tile_order), so the hardware
// dispatcher hands the short tiles to the slots that free up last.
// TRACK = false (GsrParams.forward_only): nothing the backward needs is tracked or written
template <bool STATS, bool TRACK>
__global__ __launch_bounds__(WAVES_PER_BLOCK * WAVE) void render_fwd_kernel(int W, int H, int grid_x, int num_tiles,
                                                          const uint32_t* __restrict__ tile_order,
                                                          const uint2* __restrict__ ranges,
                                                          const uint32_t* __restrict__ point_list,
                                                          const GeomRec* __restrict__ rec,
                                                          const float* __restrict__ bg,
                                                          float* __restrict__ out_color,
                                                          float* __restrict__ final_T,
                                                          uint32_t* __restrict__ n_contrib,
                                                          uint32_t* __restrict__ tile_max,
                                                          unsigned long long* __restrict__ stats) {
  __shared__ float4 sA[WAVES_PER_BLOCK][BATCH];
  __shared__ float4 sB[WAVES_PER_BLOCK][BATCH];
  __shared__ float sC[WAVES_PER_BLOCK][BATCH];
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
  const int slot = blockIdx.x * WAVES_PER_BLOCK + wid;
  if (slot >= num_tiles) return;
  const int tile = __builtin_amdgcn_readfirstlane((int)tile_order[slot]);
  render_fwd_tile<STATS, TRACK>(tile, sA[wid], sB[wid], sC[wid], W, H, grid_x, ranges, point_list, rec, bg, out_color, final_T, n_contrib,
                         tile_max, stats);
}

// ---- wave-wide sums ---------------------------------------------------------------------------
template <int CTRL>
__device__ inline float dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false);
  return v + __int_as_float(moved);
}
// every lane ends with the sum over its 16-lane row
__device__ inline float row_sum16(float v) {
  v = dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);   // row_half_mirror
  v = dpp_add<0x140>(v);   // row_mirror
  return v;
}
// lanes 0..31 end with x[i] + x[i+32], lanes 32..63 with y[i-32] + y[i]
__device__ inline float fold32(float x, float y) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// rows (0,1,2,3) end with (x.r0 + x.r1, y.r0 + y.r1, x.r2 + x.r3, y.r2 + y.r3)
__device__ inline float fold16(float x, float y) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// Wave-wide sums of four values at once: row 0 of the result holds sum(q0), row 1 sum(q2),
// row 2 sum(q1), row 3 sum(q3) (every lane of the row).  10 VALU ops instead of 24.
__device__ inline float wave_sum4(float q0, float q1, float q2, float q3) {
  return row_sum16(fold16(fold32(q0, q1), fold32(q2, q3)));
}

__device__ __forceinline__ void render_bwd_tile(const int tile, float4* sA, float4* sB, float* sC, int W, int H,
                                                int grid_x, const uint2* __restrict__ ranges,
                                                const uint32_t* __restrict__ point_list,
                                                const GeomRec* __restrict__ rec,
                                                const uint32_t* __restrict__ slot_base, const float* __restrict__ bg,
                                                const float* __restrict__ final_T,
                                                const uint32_t* __restrict__ n_contrib,
                                                const uint32_t* __restrict__ tile_max,
                                                const float* __restrict__ dL_dpix, GradRow* __restrict__ rows,
                                                uint8_t* __restrict__ row_flags) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int tile_x = tile % grid_x, tile_y = tile / grid_x;
  const int px0 = tile_x * TILE + (lane & 7), py0 = tile_y * TILE + (lane >> 3);
  const float tx0 = (float)(tile_x * TILE), ty0 = (float)(tile_y * TILE);
  float pxf0 = (float)px0, pyf0 = (float)py0;
  asm volatile("" : "+v"(pxf0), "+v"(pyf0));
  const size_t HW = (size_t)W * H;
  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];

  // per pixel: T (running transmittance in front of the current instance), Bk = sum over the
  // instances behind of (c.dL_dpix)*alpha*T  +  T_final*(bg.dL_dpix)
  float T[4], Bk[4], dpr[4], dpg[4], dpb[4];
  uint32_t last[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int px = px0 + 8 * (k & 1), py = py0 + 8 * (k >> 1);
    const bool in = px < W && py < H;
    const size_t pix = (size_t)py * W + px;
    T[k] = in ? final_T[pix] : 0.0f;
    last[k] = in ? n_contrib[pix] : 0u;
    dpr[k] = in ? dL_dpix[pix] : 0.0f;
    dpg[k] = in ? dL_dpix[HW + pix] : 0.0f;
    dpb[k] = in ? dL_dpix[2 * HW + pix] : 0.0f;
    Bk[k] = T[k] * (bg0 * dpr[k] + bg1 * dpg[k] + bg2 * dpb[k]);
  }

  const uint2 range = ranges[tile];
  const uint32_t start = range.x;
  uint32_t hi = min(range.y - range.x, tile_max[tile]);   // instances past the last contributor get no gradient

  Staged st;
  st.q0 = st.q1 = st.q2 = make_float4(0.f, 0.f, 0.f, 0.f);
  st.rect_min = st.rect_wh = st.slot_base = 0;
  // unconditional, index-clamped staging loads (see the forward kernel): ids two rounds ahead,
  // records one round ahead, walking the list back to front
  auto load_id = [&](uint32_t lo, uint32_t top) { return point_list[start + min(lo + lane, top - 1)]; };
  auto load_rec = [&](uint32_t id) {
    load_staged<true>(rec, id, st);          // q2.w = tile_mask
    st.slot_base = slot_base[id];
    const uint2 rr = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(rec + id) + 48);
    st.rect_min = rr.x;
    st.rect_wh = rr.y;
  };
  uint32_t lo = hi > BATCH ? hi - BATCH : 0u;
  uint32_t id_next = 0;
  if (hi > 0) {
    load_rec(load_id(lo, hi));
    const uint32_t lo2 = lo > BATCH ? lo - BATCH : 0u;
    id_next = load_id(lo2, max(lo, 1u));
  }

  while (hi > 0) {
    const bool have = lo + lane < hi;
    const uint32_t m = have ? subblock_mask(st.q0.x, st.q0.y, st.q2.y, st.q2.z, st.q0.z, st.q0.w, st.q1.x, st.q1.y, tx0, ty0) : 0u;
    // gradient-row slot of this instance: the Gaussian's first slot + the rank of this tile among its instances
    const uint32_t rw = st.rect_wh & 0xffffu;
    const uint32_t bit = ((uint32_t)tile_y - (st.rect_min >> 16)) * rw + ((uint32_t)tile_x - (st.rect_min & 0xffffu));
    const uint32_t slot = st.slot_base + bin_rank(st.rect_wh, __float_as_uint(st.q2.w), have ? bit : 0u);
    LdsRec lr;
    make_lds(st, lr);
    __builtin_amdgcn_wave_barrier();
    sA[lane] = lr.A;
    sB[lane] = lr.B;
    sC[lane] = st.q2.x;
    __builtin_amdgcn_wave_barrier();
    const uint32_t cur_lo = lo;
    hi = lo;
    lo = hi > BATCH ? hi - BATCH : 0u;
    {   // prefetch: record of the next (earlier) round, ids of the one after it
      load_rec(id_next);
      const uint32_t lo2 = lo > BATCH ? lo - BATCH : 0u;
      id_next = load_id(lo2, max(lo, 1u));
    }

    unsigned long long nz = __ballot(m != 0u);
    while (nz) {
      const int j = 63 - __clzll((long long)nz);   // back to front
      nz &= ~(1ull << j);
      const uint32_t mj = (uint32_t)__builtin_amdgcn_readlane((int)m, j);
      const float4 a = sA[j];
      const float4 b = sB[j];
      const float cb = sC[j];
      const uint32_t pos1 = cur_lo + (uint32_t)j + 1u;
      // per-lane partial sums over the sub-blocks; un-scaled forms (constants applied after the reduction):
      //   g_mx = sum h dx, g_my = sum h dy (first moments), g_xx = sum h dx^2, g_xy = sum h dx dy,
      //   g_yy = sum h dy^2 with h = opacity*G*dL_dalpha;  g_op = sum G*dL_dalpha;  g_r/g/b = sum alpha*T*dL_dpix
      float g_mx = 0.f, g_my = 0.f, g_xx = 0.f, g_xy = 0.f, g_yy = 0.f, g_op = 0.f, g_r = 0.f, g_g = 0.f, g_b = 0.f;
      bool any = false;
      const float dx0 = a.x - pxf0, dy0 = a.y - pyf0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (mj & (1u << k)) {
          const float dx = (k & 1) ? dx0 - 8.0f : dx0;
          const float dy = (k >> 1) ? dy0 - 8.0f : dy0;
          const float adx = a.z * dx;
          const float t = fmaf(a.w, dy, adx);
          const float cdy = b.x * dy;
          const float p2 = fmaf(dx, t, cdy * dy);
          const float G = __builtin_amdgcn_exp2f(p2);
          const float alpha = fminf(ALPHA_MAX, b.y * G);
          const bool ok = (pos1 <= last[k]) && (p2 <= 0.0f) && (alpha >= ALPHA_MIN);
          if (ok) {
            any = true;
            const float rcp = __builtin_amdgcn_rcpf(1.0f - alpha);
            T[k] *= rcp;                                   // transmittance in front of this instance
            const float cd = fmaf(cb, dpb[k], fmaf(b.w, dpg[k], b.z * dpr[k]));   // c . dL_dpix
            const float dch = alpha * T[k];
            const float dL_dalpha = fmaf(T[k], cd, -Bk[k] * rcp);
            Bk[k] = fmaf(cd, dch, Bk[k]);
            g_r = fmaf(dch, dpr[k], g_r);
            g_g = fmaf(dch, dpg[k], g_g);
            g_b = fmaf(dch, dpb[k], g_b);
            const float gd = G * dL_dalpha;
            g_op += gd;
            const float h = b.y * gd;
            const float hx = h * dx, hy = h * dy;
            g_mx += hx;                                    // first moments; the conic is applied per Gaussian
            g_my += hy;                                    // by preprocess_bwd
            g_xx = fmaf(hx, dx, g_xx);
            g_xy = fmaf(hx, dy, g_xy);
            g_yy = fmaf(hy, dy, g_yy);
          }
        }
      }
      if (__builtin_amdgcn_ballot_w64(any) != 0ull) {
        const float s0 = wave_sum4(g_mx, g_xx, g_my, g_xy);   // rows: mx, my, xx, xy
        const float s1 = wave_sum4(g_yy, g_r, g_op, g_g);     // rows: yy, op, r, g
        const float s2 = row_sum16(fold16(fold32(g_b, g_b), 0.0f));   // row 0 (and 2): b
        const uint32_t sj = (uint32_t)__builtin_amdgcn_readlane((int)slot, j);
        float* dst = reinterpret_cast<float*>(rows + sj);
        if ((lane & 15) == 0) {
          const int r = lane >> 4;
          const float f0 = r < 2 ? 1.0f : (r == 2 ? -0.5f : -1.0f);
          dst[r] = s0 * f0;                                    // Mx, My (first moments), dcxx, dcxy
          const float f1 = r == 0 ? -0.5f : 1.0f;
          dst[4 + r] = s1 * f1;                                // dcyy, dop, dr, dg
          if (r == 0) {
            dst[8] = s2;
            row_flags[sj] = 1;
          }
        }
      }
    }
  }
}

__global__ __launch_bounds__(WAVES_PER_BLOCK * WAVE, 6) void render_bwd_kernel(int W, int H, int grid_x, int num_tiles,
                                                          const uint32_t* __restrict__ tile_order,
                                                          const uint2* __restrict__ ranges,
                                                          const uint32_t* __restrict__ point_list,
                                                          const GeomRec* __restrict__ rec,
                                                          const uint32_t* __restrict__ slot_base,
                                                          const float* __restrict__ bg,
                                                          const float* __restrict__ final_T,
                                                          const uint32_t* __restrict__ n_contrib,
                                                          const uint32_t* __restrict__ tile_max,
                                                          const float* __restrict__ dL_dpix,
                                                          GradRow* __restrict__ rows,
                                                          uint8_t* __restrict__ row_flags) {
  __shared__ float4 sA[WAVES_PER_BLOCK][BATCH];
  __shared__ float4 sB[WAVES_PER_BLOCK][BATCH];
  __shared__ float sC[WAVES_PER_BLOCK][BATCH];
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
  const int slot = blockIdx.x * WAVES_PER_BLOCK + wid;
  if (slot >= num_tiles) return;
  const int tile = __builtin_amdgcn_readfirstlane((int)tile_order[slot]);
  render_bwd_tile(tile, sA[wid], sB[wid], sC[wid], W, H, grid_x, ranges, point_list, rec, slot_base, bg, final_T, n_contrib,
                  tile_max, dL_dpix,
                  rows, row_flags);
}

void launch_render_fwd(int W, int H, const uint2* ranges, const uint32_t* point_list, const GeomRec* rec,
                       const float* bg, float* out_color, float* final_T, uint32_t* n_contrib, uint32_t* tile_max,
                       const uint32_t* tile_order, hipStream_t s, unsigned long long* stats) {
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  const int nblk = (gx * gy + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
  if (stats)
    hipLaunchKernelGGL((render_fwd_kernel<true, true>), dim3(nblk), dim3(WAVES_PER_BLOCK * WAVE), 0, s, W, H, gx, gx * gy, tile_order, ranges,
                       point_list, rec, bg, out_color, final_T, n_contrib, tile_max, stats);
  else if (final_T && n_contrib && tile_max)
    hipLaunchKernelGGL((render_fwd_kernel<false, true>), dim3(nblk), dim3(WAVES_PER_BLOCK * WAVE), 0, s, W, H, gx, gx * gy, tile_order, ranges,
                       point_list, rec, bg, out_color, final_T, n_contrib, tile_max, stats);
  else      // forward only: no per-pixel state for a backward
    hipLaunchKernelGGL((render_fwd_kernel<false, false>), dim3(nblk), dim3(WAVES_PER_BLOCK * WAVE), 0, s, W, H, gx, gx * gy, tile_order, ranges,
                       point_list, rec, bg, out_color, final_T, n_contrib, tile_max, stats);
}
void launch_render_bwd(int W, int H, const uint2* ranges, const uint32_t* point_list, const GeomRec* rec,
                       const uint32_t* slot_base, const float* bg, const float* final_T, const uint32_t* n_contrib, const uint32_t* tile_max,
                       const float* dL_dpix, GradRow* rows, uint8_t* row_flags, const uint32_t* tile_order,
                       hipStream_t s) {
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  const int nblk = (gx * gy + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
  hipLaunchKernelGGL(render_bwd_kernel, dim3(nblk), dim3(WAVES_PER_BLOCK * WAVE), 0, s, W, H, gx, gx * gy, tile_order, ranges, point_list, rec,
                     slot_base, bg, final_T, n_contrib, tile_max, dL_dpix, rows, row_flags);
}

}  // namespace gsr
