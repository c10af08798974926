// Per-tile alpha compositing, forward and backward (§8 a10, A.5).
//
// CDNA4 mappings (not the upstream 16x16-thread block).  What drives both: on gfx950 a wave64 v_fma/v_mul/v_add issues
// every ~2.4 cycles per SIMD, v_cmp / v_cndmask / v_min / v_max every ~4.7, v_exp / v_rcp every ~8.3, SALU every ~4.7,
// and a half-empty EXEC mask saves nothing (profiles/r02/valu_rate.txt) -- the kernels are bound by VALU issue, so the
// lever is the number of 64-lane evaluations per (Gaussian, tile) instance and the instructions per evaluation.
//
// FORWARD: one 256-lane workgroup per tile; wave w owns the 8x8 sub-block w and each of its four 16-lane groups (a DPP
// row) owns one 4x4 MINI-BLOCK.  Per round the workgroup stages 256 instances (one per lane) into LDS, the staging lane
// computes from row spans of the alpha >= 1/255 ellipse which of the tile's 16 mini-blocks the instance can reach, and
// the instances are compacted -- in list order -- into 16 per-mini-block visit lists (wave-level DPP prefix sums of
// packed 8-bit counters; no atomics).  Each 16-lane group then walks ITS OWN list: one wave instruction evaluates four
// different (instance, mini-block) pairs.  A 7x7-pixel footprint costs ~7 groups of 16 lanes instead of ~3.5 sub-blocks
// of 64.  Skipped pairs would have been rejected pixel by pixel by the alpha test, so results are unchanged.
//
// BACKWARD: one wave per tile, four pixels per lane (one per 8x8 sub-block), because the nine per-instance gradient sums
// must be reduced across the pixels of the tile: per-instance gradients are reduced across the wave in registers (DPP /
// permlane swaps) and written as one row per (Gaussian, tile) instance -- no atomics; preprocess_bwd sums the rows.
#include "gsr_common.h"
#include "gsr_launch.h"
#include <stdlib.h>

namespace gsr {

constexpr float LOG2E = 1.4426950408889634f;

// What one lane fetches for the instance it stages (GeomRec words 0..10, 14..15 and, for the backward, 11..13).
struct Staged {
  float4 q0, q1, q2;
  float kk, isyy;
  uint32_t rect_min, rect_wh, slot_base;
};

// FULL = false (forward) skips the words the forward never reads -- 4 (cyy) and 11 (tile_mask): a dead destination
// register of an in-flight load gets recycled by the compiler and forces an early s_waitcnt vmcnt right behind the
// prefetch, which exposes the whole gather latency.
template <bool FULL>
__device__ inline void load_staged(const GeomRec* __restrict__ rec, uint32_t id, Staged& s) {
  const float4* r = reinterpret_cast<const float4*>(rec + id);
  if (FULL) {
    s.q0 = r[0];
    s.q1 = r[1];
    s.q2 = r[2];
  } else {
    const float* f = reinterpret_cast<const float*>(r + 1);
    const float* f0 = reinterpret_cast<const float*>(r);
    s.q0.x = f0[0];      // word 3 (cxy) is not read either: kk replaces it
    s.q0.y = f0[1];
    s.q0.z = f0[2];
    s.q1.y = f[1];
    s.q1.z = f[2];
    s.q1.w = f[3];
    s.q2.x = f[4];
    s.q2.y = f[5];
    s.q2.z = f[6];
  }
  const float2 k = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(rec + id) + 56);
  s.kk = k.x;
  s.isyy = k.y;
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

// LDS image of a staged instance.  The quadratic form is kept as a completed square, pre-scaled by log2(e) so that
// the exponential is a bare v_exp_f32:
//    log2(e) * power = nka * u^2 + nkd * dy^2,   u = dx + kk * dy,   (dx, dy) = mean - pixel,
//    nka = -0.5 log2e cxx,  kk = cxy / cxx,  nkd = -0.5 log2e / cov_yy.
// Both terms are <= 0 in float32 whatever the rounding, so the reference's "power > 0 -> skip" guard (Appendix A.4)
// can never fire -- it only ever fired on rounding noise of the expanded form -- and costs no compare here.
struct LdsRec {
  float4 A;   // x, y, nka, kk
  float4 B;   // nkd, opacity, r, g
};
__device__ inline void make_lds(const Staged& st, LdsRec& o) {
  o.A = make_float4(st.q0.x, st.q0.y, (-0.5f * LOG2E) * st.q0.z, st.kk);
  o.B = make_float4((-0.5f * LOG2E) * st.isyy, st.q1.y, st.q1.z, st.q1.w);
}
// the same five operations in the forward and in the backward: both must take the same alpha >= 1/255 decisions
__device__ __forceinline__ float pair_p2(float dx, float dy, float nka, float kk, float nkd) {
#pragma clang fp contract(off)
  const float u = __builtin_fmaf(kk, dy, dx);
  const float s = nka * u;
  const float v = (nkd * dy) * dy;
  return __builtin_fmaf(s, u, v);
}

// ---- wave-level helpers of the forward --------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_shift_add(uint32_t v) {
  return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
// inclusive prefix sum over the 64 lanes of a wave; the four bytes of v are independent counters (totals <= 64).
// Called with all lanes active.
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t v) {
  v = dpp_shift_add<0x111, 0xf>(v);   // row_shr:1
  v = dpp_shift_add<0x112, 0xf>(v);   // row_shr:2
  v = dpp_shift_add<0x114, 0xf>(v);   // row_shr:4
  v = dpp_shift_add<0x118, 0xf>(v);   // row_shr:8: inclusive within each row of 16
  v = dpp_shift_add<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
  v = dpp_shift_add<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3
  return v;
}

constexpr int QROUND = 256;           // instances staged per round (one per lane of the workgroup)
constexpr int QLIST = QROUND + 8;     // list capacity: the walk reads up to 4 entries past the longest list
constexpr int QDUMMY = QROUND;        // LDS slot of the all-zero record the lists are padded with (alpha = 0)

// Mini-block reach mask of one instance: bit 4*r + c <-> the 4x4 pixel block at tile-relative (4c, 4r).
// The alpha >= 1/255 region is the ellipse q <= t.  Along a pixel row dy (relative to the mean) its x-extent is
//    xc(dy) -+ hw(dy),  xc = mx - kk dy,  hw^2 = (t / cxx) * max(0, 1 - (dy / ey)^2)       (ey = sqrt(t cov_yy)).
// The region is convex, so x_left(dy) is convex with its minimum at the ellipse's leftmost point dy = dyl: over a block
// row [ya, yb] the smallest x_left is x_left(clamp(dyl, ya, yb)) -- one span evaluation per side and block row.
// Everything is inflated (t by 1.002 twice over, half-widths by 0.03 px; dyl only needs to be approximate, x_left is
// flat there), so that a dropped pair is a pair every pixel of which fails alpha >= 1/255 in the evaluation's float32
// arithmetic.
__device__ inline uint32_t miniblock_mask(float mx, float my, float cxx, float kk, float isyy, float ex, float ey) {
  if (ex < 0.0f) return 0u;
  if (mx + ex < 0.0f || mx - ex > 15.0f || my + ey < 0.0f || my - ey > 15.0f) return 0u;
  const float t = ey * ey * isyy * 1.002f;                    // >= 2 ln(255 opacity), from the inflated ey
  const float h2 = t * __builtin_amdgcn_rcpf(cxx) * 1.002f;
  const float iey = __builtin_amdgcn_rcpf(ey);
  // row (relative to the mean) of the ellipse's leftmost point: -ex * cov_xy / cov_xx, cov_xy / cov_xx = -kk (ey/ex)^2;
  // the rightmost point is at -dyl
  const float dyl = kk * ey * (ey * __builtin_amdgcn_rcpf(ex));
  auto half_width = [&](float dy) {
    const float u = dy * iey;
    return __builtin_amdgcn_sqrtf(fmaxf(0.0f, 1.0f - u * u) * h2) + 0.03f;
  };
  uint32_t m = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float ya = (float)(4 * r) - my, yb = ya + 3.0f;
    const float dl = __builtin_amdgcn_fmed3f(dyl, ya, yb), dr = __builtin_amdgcn_fmed3f(-dyl, ya, yb);
    const float left = (mx - kk * dl) - half_width(dl);
    const float right = (mx - kk * dr) + half_width(dr);
    // block column c holds the pixel centres 4c .. 4c+3
    const float lo = fmaxf(0.0f, ceilf((left - 3.0f) * 0.25f));
    const float hi = fminf(3.0f, floorf(right * 0.25f));
    if ((yb >= -ey) && (ya <= ey) && hi >= lo) {
      const uint32_t cols = ((2u << (4 * r)) << (uint32_t)hi) - ((1u << (4 * r)) << (uint32_t)lo);
      m |= cols;
    }
  }
  return m;
}

// One (instance, pixel) pair of the forward.  T > 0 while the pixel is live; a pixel that saturates keeps its final
// transmittance with the sign flipped (outside the image: T = 0), so every later test_T = T (1 - alpha) <= 0 < 1e-4
// keeps it out of the blend without a separate flag.
template <bool TRACK, bool CLAMP>
__device__ __forceinline__ void blend_pair(const float4 a, const float4 b, const float cb, const uint32_t pos1,
                                           const float pxf, const float pyf, float& T, float& Cr, float& Cg, float& Cb,
                                           uint32_t& last) {
  const float dx = a.x - pxf, dy = a.y - pyf;
  const float p2 = pair_p2(dx, dy, a.z, a.w, b.x);
  float alpha = b.y * __builtin_amdgcn_exp2f(p2);
  if (CLAMP) alpha = fminf(ALPHA_MAX, alpha);     // opacity <= 0.99 cannot reach the clamp: exp2(p2 <= 0) <= 1
  if (alpha >= ALPHA_MIN) {
    const float test_T = T * (1.0f - alpha);
    if (!(test_T < T_STOP)) {
      const float w = alpha * T;
      Cr = fmaf(b.z, w, Cr);
      Cg = fmaf(b.w, w, Cg);
      Cb = fmaf(cb, w, Cb);
      if (TRACK) last = pos1;
      T = test_T;
    } else {
      T = -fabsf(T);      // saturates here (or is parked already)
    }
  }
}

// The visit lists hold the LDS byte offset of the staged record's A / B vectors (16 * slot; the colour-b array is
// addressed with a quarter of it).  One-deep software pipeline: the next list entry and the next record are fetched
// while the current pair is evaluated.
template <bool TRACK, bool CLAMP>
__device__ __forceinline__ void walk_lists(const uint16_t* __restrict__ mylist, const uint32_t nmax, const uint32_t base1,
                                           const char* sA, const char* sB, const char* sC, const float pxf,
                                           const float pyf, float& T, float& Cr, float& Cg, float& Cb, uint32_t& last) {
  auto ldA = [&](uint32_t e) { return *reinterpret_cast<const float4*>(sA + e); };
  auto ldB = [&](uint32_t e) { return *reinterpret_cast<const float4*>(sB + e); };
  auto ldC = [&](uint32_t e) { return *reinterpret_cast<const float*>(sC + (e >> 2)); };
  uint32_t e0 = mylist[0], e1 = mylist[1];
  float4 a0 = ldA(e0), b0 = ldB(e0);
  float c0 = ldC(e0);
#pragma unroll 2
  for (uint32_t i = 0; i < nmax; ++i) {
    const float4 a1 = ldA(e1), b1 = ldB(e1);
    const float c1 = ldC(e1);
    const uint32_t e2 = mylist[i + 2];
    blend_pair<TRACK, CLAMP>(a0, b0, c0, base1 + (e0 >> 4), pxf, pyf, T, Cr, Cg, Cb, last);
    a0 = a1; b0 = b1; c0 = c1;
    e0 = e1; e1 = e2;
  }
}

template <bool STATS, bool TRACK>
__global__ __launch_bounds__(256, 8) void render_fwd_kernel(int W, int H, int grid_x, int num_tiles,
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
  __shared__ float4 sA[QROUND + 1];
  __shared__ float4 sB[QROUND + 1];
  __shared__ float sC[QROUND + 4];
  __shared__ __attribute__((aligned(16))) uint16_t sList[16][QLIST];
  __shared__ uint4 sCnt[4];             // per wave: counts of the 16 mini-blocks, one byte each
  __shared__ uint32_t sFlag[2][4];      // [0] any opacity > ALPHA_MAX in the round, [1] tile_max partials
  (void)num_tiles;
  const int tid = threadIdx.x;
  const int lane = tid & (WAVE - 1);
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = lane >> 4, q = lane & 15;
  const int tile = __builtin_amdgcn_readfirstlane((int)tile_order[blockIdx.x]);
  const int tile_x = tile % grid_x, tile_y = tile / grid_x;
  const int bx = 8 * (wid & 1) + 4 * (grp & 1) + (q & 3), by = 8 * (wid >> 1) + 4 * (grp >> 1) + (q >> 2);
  const int px = tile_x * TILE + bx, py = tile_y * TILE + by;
  const float tx0 = (float)(tile_x * TILE), ty0 = (float)(tile_y * TILE);
  const float pxf = (float)px, pyf = (float)py;
  const bool inside = px < W && py < H;

  float T = inside ? 1.0f : 0.0f, Cr = 0.0f, Cg = 0.0f, Cb = 0.0f;
  uint32_t last = 0;

  const uint2 range = ranges[tile];
  const uint32_t start = range.x, len = range.y - range.x;
  if (tid == 0) {
    sA[QDUMMY] = make_float4(0.f, 0.f, 0.f, 0.f);
    sB[QDUMMY] = make_float4(0.f, 0.f, 0.f, 0.f);
    sC[QDUMMY] = 0.0f;
  }
  // this 16-lane group's mini-block: row / column of 4x4 blocks inside the tile (bit 4 * blk_r + blk_c of the reach mask)
  const int blk_r = 2 * (wid >> 1) + (grp >> 1), blk_c = 2 * (wid & 1) + (grp & 1);
  const uint16_t* mylist = &sList[4 * blk_r + blk_c][0];
  unsigned long long st_pairs = 0, st_evals = 0;

  // software pipeline: instance ids are fetched two rounds ahead and GeomRec lines one round ahead -- the gathers of
  // round r+1 are in flight while round r is composited (a gather of 64-byte lines at 3 TB/s would otherwise cost as
  // much time as the compositing itself)
  Staged st;
  st.q0 = st.q1 = st.q2 = make_float4(0.f, 0.f, 0.f, 0.f);
  st.kk = st.isyy = 0.0f;
  uint32_t id_next = 0;
  if (len) {
    load_staged<false>(rec, point_list[start + min((uint32_t)tid, len - 1)], st);
    id_next = point_list[start + min((uint32_t)(QROUND + tid), len - 1)];
  }
  for (uint32_t base = 0; base < len; base += QROUND) {
    // ---- (a) stage one instance per lane, find the mini-blocks it reaches ------------------------------------
    const bool have = base + tid < len;
    {   // pad every list with the dummy slot (16-byte stores)
      const uint4 d = make_uint4(16u * QDUMMY * 0x10001u, 16u * QDUMMY * 0x10001u, 16u * QDUMMY * 0x10001u, 16u * QDUMMY * 0x10001u);
      uint4* l4 = reinterpret_cast<uint4*>(&sList[0][0]);
      for (int c = tid; c < 16 * QLIST * 2 / 16; c += 256) l4[c] = d;
    }
    LdsRec lr;
    make_lds(st, lr);
    sA[tid] = lr.A;
    sB[tid] = lr.B;
    sC[tid] = st.q2.x;
    const uint32_t m16 = have ? miniblock_mask(st.q0.x - tx0, st.q0.y - ty0, st.q0.z, st.kk, st.isyy, st.q2.y, st.q2.z) : 0u;
    // ---- (b) wave-level ranks: four packed registers of four 8-bit counters --------------------------------------
    uint32_t own[4], incl[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      own[r] = (((m16 >> (4 * r)) & 0xfu) * 0x00204081u) & 0x01010101u;
      incl[r] = wave_incl_scan_dpp(own[r]);
    }
    if (lane == WAVE - 1) sCnt[wid] = make_uint4(incl[0], incl[1], incl[2], incl[3]);
    const bool opaque = have && lr.B.y > ALPHA_MAX;
    const bool wave_opaque = __builtin_amdgcn_ballot_w64(opaque) != 0ull;
    if (lane == 0) sFlag[0][wid] = wave_opaque ? 1u : 0u;
    __syncthreads();
    // ---- (c) list positions = counts of the earlier waves + rank within the wave ---------------------------------
    {
      uint32_t offs[4] = {0u, 0u, 0u, 0u};
      for (int w = 0; w < wid; ++w) {
        const uint4 c = sCnt[w];
        offs[0] += c.x; offs[1] += c.y; offs[2] += c.z; offs[3] += c.w;     // <= 192 per byte
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t posp = offs[r] + incl[r] - own[r];                   // <= 255 per byte
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if ((m16 >> (4 * r + k)) & 1u) sList[4 * r + k][(posp >> (8 * k)) & 0xffu] = (uint16_t)(16 * tid);
      }
    }
    const bool clamp = (sFlag[0][0] | sFlag[0][1] | sFlag[0][2] | sFlag[0][3]) != 0u;
    // this lane's list length: mini-block (blk_r, blk_c) = byte blk_c of packed register blk_r
    uint32_t n_lane = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const uint32_t* c = reinterpret_cast<const uint32_t*>(&sCnt[w]);
      n_lane += (c[blk_r] >> (8 * blk_c)) & 0xffu;
    }
    const uint32_t nmax = max(max((uint32_t)__builtin_amdgcn_readlane((int)n_lane, 0), (uint32_t)__builtin_amdgcn_readlane((int)n_lane, 16)),
                              max((uint32_t)__builtin_amdgcn_readlane((int)n_lane, 32), (uint32_t)__builtin_amdgcn_readlane((int)n_lane, 48)));
    // gathers of the next round (clamped indices: lanes past the end re-read the last instance and are masked by `have`)
    load_staged<false>(rec, id_next, st);
    id_next = point_list[start + min(base + 2u * QROUND + (uint32_t)tid, len - 1)];
    __syncthreads();
    // ---- (d) every 16-lane group walks its own list --------------------------------------------------------------
    if (STATS) { st_pairs += (q == 0) ? n_lane : 0u; st_evals += (lane == 0) ? nmax : 0u; }
#ifndef GSR_EXPERIMENT_NO_WALK     // tuning experiment: staging / culling / compaction only
    if (clamp) walk_lists<TRACK, true>(mylist, nmax, base + 1u, (const char*)sA, (const char*)sB, (const char*)sC, pxf, pyf, T, Cr, Cg, Cb, last);
    else       walk_lists<TRACK, false>(mylist, nmax, base + 1u, (const char*)sA, (const char*)sB, (const char*)sC, pxf, pyf, T, Cr, Cg, Cb, last);
#else
    if (clamp && nmax == 0xffffffffu) T = 0.0f;
#endif
    // ---- (e) stop when every pixel of the tile is parked; also fences the LDS reuse ------------------------------
    if (__syncthreads_and(!(T > 0.0f))) break;
  }

  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
  const size_t HW = (size_t)W * H;
  if (inside) {
    const float Tf = fabsf(T);
    const size_t pix = (size_t)py * W + px;
    out_color[pix] = Cr + Tf * bg0;
    out_color[HW + pix] = Cg + Tf * bg1;
    out_color[2 * HW + pix] = Cb + Tf * bg2;
    if (TRACK) {
      final_T[pix] = Tf;
      n_contrib[pix] = last;
    }
  }
  if (TRACK) {
    uint32_t mx = inside ? last : 0u;
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, WAVE));
    if (lane == 0) sFlag[1][wid] = mx;
    __syncthreads();
    if (tid == 0) tile_max[tile] = max(max(sFlag[1][0], sFlag[1][1]), max(sFlag[1][2], sFlag[1][3]));
  }
  if (STATS) {
    // [0] instances in the tile lists, [1] instances staged, [2] (instance, mini-block) pairs, [3] wave evaluations
    // (each covers up to four pairs), [5] sum of tile_max
    st_pairs = (unsigned long long)wave_reduce_add_u32((uint32_t)st_pairs);
    if (lane == 0) {
      atomicAdd(&stats[2], st_pairs);
      atomicAdd(&stats[3], st_evals);
    }
    if (tid == 0) {
      atomicAdd(&stats[0], (unsigned long long)len);
      atomicAdd(&stats[1], (unsigned long long)min(len, ((len + QROUND - 1) / QROUND) * QROUND));
      if (TRACK) atomicAdd(&stats[5], (unsigned long long)max(max(sFlag[1][0], sFlag[1][1]), max(sFlag[1][2], sFlag[1][3])));
    }
  }
}

// ---- backward -------------------------------------------------------------------------------------------------------
// Same mapping as the forward (workgroup per tile, wave per 8x8 sub-block, 16-lane group per 4x4 mini-block, per-group
// visit lists), walked back to front.  Each wave instruction evaluates four (instance, mini-block) pairs; the nine
// gradient sums of a pair are reduced over the 16 lanes of its DPP row by a fold tree (26 VALU operations for all nine
// values: after every level two half-empty registers are merged into one through a bank-masked DPP move), added into
// the owning wave's LDS accumulator of the instance, and when the round is over one lane per instance adds the four
// waves' accumulators in wave order and stores ONE 48-byte row per (Gaussian, tile) instance -- no global atomics,
// bitwise reproducible; preprocess_bwd sums a Gaussian's rows.
constexpr int BROUND = 128;            // instances staged per round
constexpr int BLIST = BROUND + 8;      // the walk reads up to 4 entries past the longest list
constexpr int BDUMMY = BROUND;

// v + (v moved by a DPP control every lane of which has a valid source: rotations, mirrors, quad permutes);
// bound_ctrl lets the compiler fold the move into a single v_add_f32_dpp
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true);
  return v + __int_as_float(moved);
}
// lanes of the banks in BANK_MASK (bank = 4 consecutive lanes of a 16-lane row) take b, the others keep a
template <int BANK_MASK>
__device__ __forceinline__ float dpp_merge(float a, float b) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(a), __float_as_int(b), 0xE4, 0xf, BANK_MASK, false));
}
// Sums of nine values over each 16-lane row.  Result: n0 banks (0,1,2,3) = sums of (v0, v2, v1, v3), n1 banks = sums of
// (v4, v6, v5, v7), n2 every bank = sum of v8 (each sum in all four lanes of its bank).
__device__ __forceinline__ void row_fold9(const float (&v)[9], float& n0, float& n1, float& n2) {
  float w[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) w[i] = dpp_add<0x128>(v[i]);            // row_ror:8  -> lanes l and l+8 agree
  float m0 = dpp_merge<0xc>(w[0], w[1]);                              // lanes 0-7: v0, lanes 8-15: v1
  float m1 = dpp_merge<0xc>(w[2], w[3]);
  float m2 = dpp_merge<0xc>(w[4], w[5]);
  float m3 = dpp_merge<0xc>(w[6], w[7]);
  float m4 = w[8];
  m0 = dpp_add<0x141>(m0); m1 = dpp_add<0x141>(m1); m2 = dpp_add<0x141>(m2);    // row_half_mirror: lanes i and 7-i
  m3 = dpp_add<0x141>(m3); m4 = dpp_add<0x141>(m4);
  n0 = dpp_merge<0xa>(m0, m1);                                        // banks 0,2 from m0 (v0 | v1), banks 1,3 from m1
  n1 = dpp_merge<0xa>(m2, m3);
  n2 = m4;
  n0 = dpp_add<0xB1>(n0); n1 = dpp_add<0xB1>(n1); n2 = dpp_add<0xB1>(n2);      // quad_perm [1,0,3,2]
  n0 = dpp_add<0x4E>(n0); n1 = dpp_add<0x4E>(n1); n2 = dpp_add<0x4E>(n2);      // quad_perm [2,3,0,1]
}

// One (instance, pixel) pair of the backward; `acc` is the lane's accumulator row of the instance in its wave's LDS
// slab and `vsel` the lane's value index inside fold registers n0 / n1 (see row_fold9).
template <bool CLAMP>
__device__ __forceinline__ void grad_pair(const float4 a, const float4 b, const float cb, const uint32_t pos1,
                                          const float pxf, const float pyf, float& T, float& Bk, const float dpr,
                                          const float dpg, const float dpb, const uint32_t last, float* acc,
                                          const bool clash, const int sel, const int acc_lane, const int grp) {
  const float dx = a.x - pxf, dy = a.y - pyf;
  const float p2 = pair_p2(dx, dy, a.z, a.w, b.x);
  const float G = __builtin_amdgcn_exp2f(p2);
  float alpha = b.y * G;
  if (CLAMP) alpha = fminf(ALPHA_MAX, alpha);
  const bool ok = (pos1 <= last) && (alpha >= ALPHA_MIN);
  if (__builtin_amdgcn_ballot_w64(ok) != 0ull) {             // else: nothing to accumulate for any of the four pairs
    // lanes that do not contribute run the same instructions on G = alpha = 0: every product below is then exactly 0
    const float Gm = ok ? G : 0.0f;
    const float am = ok ? alpha : 0.0f;
    const float rcp = __builtin_amdgcn_rcpf(1.0f - am);
    if (ok) T *= rcp;                                          // transmittance in front of this instance
    const float cd = fmaf(cb, dpb, fmaf(b.w, dpg, b.z * dpr));  // c . dL_dpix
    const float dch = am * T;
    const float dL_dalpha = fmaf(T, cd, -Bk * rcp);
    Bk = fmaf(cd, dch, Bk);
    const float gd = Gm * dL_dalpha;
    const float h = b.y * gd;
    const float hx = h * dx, hy = h * dy;
    // un-scaled sums (constants applied once per row): first moments h dx, h dy (the conic is applied per Gaussian by
    // preprocess_bwd), second moments h dx^2, h dx dy, h dy^2, G dL_dalpha, alpha T dL_dpix
    const float v[9] = {hx, hy, hx * dx, hx * dy, hy * dy, gd, dch * dpr, dch * dpg, dch * dpb};
    float n0, n1, n2;
    row_fold9(v, n0, n1, n2);
    // lane j of bank k takes the bank's value of n0 (j = 0), n1 (j = 1), n2 (j = 2, bank 0 only): the nine sums of the
    // row sit in nine lanes of one register and go out with a single LDS operation
    const float nn = (sel == 0) ? n0 : (sel == 1 ? n1 : n2);
    // Plain read-add-write into the wave's own slab (LDS float atomics retire about one lane per cycle per CU: they
    // cost 0.45 ms at C4).  Two groups of the wave may hold the same instance at the same step (`clash`, found once
    // per round from the lists): then the groups go one after the other, in group order, so the sums stay bitwise
    // reproducible.
    const bool adds = acc_lane >= 0;
    float* dst = acc + (acc_lane < 0 ? 0 : acc_lane);
    if (!clash) {
      if (adds) *dst += nn;
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (adds && grp == g) *dst += nn;
        __builtin_amdgcn_wave_barrier();     // LDS operations of one wave execute in order
      }
    }
  }
}

template <bool CLAMP>
__device__ __forceinline__ void walk_lists_bwd(const uint16_t* __restrict__ mylist, const uint32_t nmax, const uint32_t base1,
                                               const char* sA, const char* sB, const char* sC, float* slab,
                                               const float pxf, const float pyf, float& T, float& Bk, const float dpr,
                                               const float dpg, const float dpb, const uint32_t last, const int sel,
                                               const int acc_lane, const int grp, const unsigned long long clash_lo,
                                               const unsigned long long clash_hi) {
  auto ldA = [&](uint32_t e) { return *reinterpret_cast<const float4*>(sA + e); };
  auto ldB = [&](uint32_t e) { return *reinterpret_cast<const float4*>(sB + e); };
  auto ldC = [&](uint32_t e) { return *reinterpret_cast<const float*>(sC + (e >> 2)); };
  auto clash_at = [&](uint32_t i) { return (((i < 64u ? clash_lo : clash_hi) >> (i & 63u)) & 1ull) != 0ull; };
  // unrolled by hand (the compiler does not unroll a loop with convergent operations): two pairs per trip, records
  // one pair ahead; an odd tail lands on the dummy entry the lists are padded with
  uint32_t e0 = mylist[0], e1 = mylist[1];
  float4 a0 = ldA(e0), b0 = ldB(e0);
  float c0 = ldC(e0);
  for (uint32_t i = 0; i < nmax; i += 2) {
    const float4 a1 = ldA(e1), b1 = ldB(e1);
    const float c1 = ldC(e1);
    const uint32_t e2 = mylist[i + 2], e3 = mylist[i + 3];
    grad_pair<CLAMP>(a0, b0, c0, base1 + (e0 >> 4), pxf, pyf, T, Bk, dpr, dpg, dpb, last, slab + 9 * (e0 >> 4), clash_at(i), sel, acc_lane, grp);
    a0 = ldA(e2); b0 = ldB(e2); c0 = ldC(e2);
    grad_pair<CLAMP>(a1, b1, c1, base1 + (e1 >> 4), pxf, pyf, T, Bk, dpr, dpg, dpb, last, slab + 9 * (e1 >> 4), clash_at(i + 1), sel, acc_lane, grp);
    e0 = e2;
    e1 = e3;
  }
}

__global__ __launch_bounds__(256) void render_bwd_kernel(int W, int H, int grid_x, int num_tiles,
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
  __shared__ float4 sA[BROUND + 1];
  __shared__ float4 sB[BROUND + 1];
  __shared__ float sC[BROUND + 4];
  __shared__ uint32_t sSlot[BROUND];                      // gradient-row slot of the staged instance
  __shared__ __attribute__((aligned(16))) uint16_t sList[16][BLIST];
  __shared__ __attribute__((aligned(16))) float sAcc[4][BROUND + 1][9];    // per wave: nine sums per staged instance (+ the dummy's)
  __shared__ uint4 sCnt[2];                               // the two staging waves' counts of the 16 mini-blocks
  __shared__ uint32_t sFlag[2];
  (void)num_tiles;
  const int tid = threadIdx.x;
  const int lane = tid & (WAVE - 1);
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = lane >> 4, q = lane & 15;
  const int tile = __builtin_amdgcn_readfirstlane((int)tile_order[blockIdx.x]);
  const int tile_x = tile % grid_x, tile_y = tile / grid_x;
  const int blk_r = 2 * (wid >> 1) + (grp >> 1), blk_c = 2 * (wid & 1) + (grp & 1);
  const int px = tile_x * TILE + 4 * blk_c + (q & 3), py = tile_y * TILE + 4 * blk_r + (q >> 2);
  const float tx0 = (float)(tile_x * TILE), ty0 = (float)(tile_y * TILE);
  const float pxf = (float)px, pyf = (float)py;
  const bool inside = px < W && py < H;
  const size_t HW = (size_t)W * H;
  const size_t pix = (size_t)py * W + px;

  // per pixel: T (running transmittance in front of the current instance), Bk = sum over the instances behind of
  // (c . dL_dpix) alpha T  +  T_final (bg . dL_dpix)
  float T = inside ? final_T[pix] : 0.0f;
  const uint32_t last = inside ? n_contrib[pix] : 0u;
  const float dpr = inside ? dL_dpix[pix] : 0.0f;
  const float dpg = inside ? dL_dpix[HW + pix] : 0.0f;
  const float dpb = inside ? dL_dpix[2 * HW + pix] : 0.0f;
  float Bk = T * (bg[0] * dpr + bg[1] * dpg + bg[2] * dpb);

  const uint2 range = ranges[tile];
  const uint32_t start = range.x;
  const uint32_t hi = min(range.y - range.x, tile_max[tile]);   // instances past the last contributor get no gradient
  if (tid == 0) {
    sA[BDUMMY] = make_float4(0.f, 0.f, 0.f, 0.f);
    sB[BDUMMY] = make_float4(0.f, 0.f, 0.f, 0.f);
    sC[BDUMMY] = 0.0f;
  }
  const uint16_t* mylist = &sList[4 * blk_r + blk_c][0];
  float* slab = &sAcc[wid][0][0];
  // where this lane's fold results go (row_fold9): banks (0,1,2,3) of n0 / n1 hold values (0,2,1,3) (+4)
  const int bank = q >> 2, sel = q & 3;
  const int vsel = (bank == 1) ? 2 : (bank == 2 ? 1 : bank);
  // accumulator entry this lane adds to: n0 -> values 0..3, n1 -> 4..7, n2 -> 8 (bank 0 only); -1: none
  const int acc_lane = sel == 0 ? vsel : (sel == 1 ? 4 + vsel : ((sel == 2 && bank == 0) ? 8 : -1));
  const bool stager = tid < BROUND;                         // waves 0 and 1 stage one instance per lane

  // rounds of BROUND list positions, back to front; ids are fetched two rounds ahead, GeomRec lines one round ahead
  const int nrounds = (int)((hi + BROUND - 1) / BROUND);
  Staged st;
  st.q0 = st.q1 = st.q2 = make_float4(0.f, 0.f, 0.f, 0.f);
  st.kk = st.isyy = 0.0f;
  st.rect_min = st.rect_wh = st.slot_base = 0;
  uint32_t id_next = 0;
  auto pos_of = [&](int round) { return min((uint32_t)(round < 0 ? 0 : round) * BROUND + (uint32_t)tid, hi - 1); };
  auto load_rec = [&](uint32_t id) {
    load_staged<true>(rec, id, st);          // q2.w = tile_mask
    st.slot_base = slot_base[id];
    const uint2 rr = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(rec + id) + 48);
    st.rect_min = rr.x;
    st.rect_wh = rr.y;
  };
  if (hi > 0 && stager) {
    load_rec(point_list[start + pos_of(nrounds - 1)]);
    id_next = point_list[start + pos_of(nrounds - 2)];
  }
  for (int round = nrounds - 1; round >= 0; --round) {
    const uint32_t base = (uint32_t)round * BROUND;
    // ---- (a) stage, cull, zero the accumulators, pad the lists ----------------------------------------------------
    const bool have = stager && base + tid < hi;
    {
      const uint4 d = make_uint4(16u * BDUMMY * 0x10001u, 16u * BDUMMY * 0x10001u, 16u * BDUMMY * 0x10001u, 16u * BDUMMY * 0x10001u);
      uint4* l4 = reinterpret_cast<uint4*>(&sList[0][0]);
      for (int c = tid; c < 16 * BLIST * 2 / 16; c += 256) l4[c] = d;
      float4* a4 = reinterpret_cast<float4*>(&sAcc[0][0][0]);
      for (int c = tid; c < 4 * (BROUND + 1) * 9 / 4; c += 256) a4[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    uint32_t m16 = 0;
    bool opaque = false;
    if (stager) {
      LdsRec lr;
      make_lds(st, lr);
      sA[tid] = lr.A;
      sB[tid] = lr.B;
      sC[tid] = st.q2.x;
      // gradient-row slot of this instance: the Gaussian's first slot + the rank of this tile among its instances
      const uint32_t rw = st.rect_wh & 0xffffu;
      const uint32_t bit = ((uint32_t)tile_y - rect_min_y(st.rect_min)) * rw + ((uint32_t)tile_x - rect_min_x(st.rect_min));
      sSlot[tid] = st.slot_base + bin_rank(st.rect_wh, __float_as_uint(st.q2.w), have ? bit : 0u);
      if (have) m16 = miniblock_mask(st.q0.x - tx0, st.q0.y - ty0, st.q0.z, st.kk, st.isyy, st.q2.y, st.q2.z);
      opaque = have && lr.B.y > ALPHA_MAX;
    }
    // ---- (b) ranks within the wave (waves 2, 3 hold empty masks) --------------------------------------------------
    uint32_t own[4], incl[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      own[r] = (((m16 >> (4 * r)) & 0xfu) * 0x00204081u) & 0x01010101u;
      incl[r] = wave_incl_scan_dpp(own[r]);
    }
    if (stager && lane == WAVE - 1) sCnt[wid] = make_uint4(incl[0], incl[1], incl[2], incl[3]);
    const bool wave_opaque = __builtin_amdgcn_ballot_w64(opaque) != 0ull;
    if (stager && lane == 0) sFlag[wid] = wave_opaque ? 1u : 0u;
    __syncthreads();
    // ---- (c) lists in REVERSE list order: position = total - 1 - (earlier wave's count + rank) ----------------------
    const uint4 c0 = sCnt[0], c1 = sCnt[1];
    {
      const uint32_t tot[4] = {c0.x + c1.x, c0.y + c1.y, c0.z + c1.z, c0.w + c1.w};     // <= 128 per byte
      const uint32_t offs[4] = {wid ? c0.x : 0u, wid ? c0.y : 0u, wid ? c0.z : 0u, wid ? c0.w : 0u};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t fwd = offs[r] + incl[r] - own[r];                                 // forward position, per byte
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if ((m16 >> (4 * r + k)) & 1u)
            sList[4 * r + k][((tot[r] >> (8 * k)) & 0xffu) - 1u - ((fwd >> (8 * k)) & 0xffu)] = (uint16_t)(16 * tid);
      }
    }
    const bool clamp = (sFlag[0] | sFlag[1]) != 0u;
    const uint32_t* c0w = reinterpret_cast<const uint32_t*>(&c0);
    const uint32_t* c1w = reinterpret_cast<const uint32_t*>(&c1);
    const uint32_t n_lane = ((c0w[blk_r] >> (8 * blk_c)) & 0xffu) + ((c1w[blk_r] >> (8 * blk_c)) & 0xffu);
    const uint32_t nmax = max(max((uint32_t)__builtin_amdgcn_readlane((int)n_lane, 0), (uint32_t)__builtin_amdgcn_readlane((int)n_lane, 16)),
                              max((uint32_t)__builtin_amdgcn_readlane((int)n_lane, 32), (uint32_t)__builtin_amdgcn_readlane((int)n_lane, 48)));
    // gathers of the next (earlier) round
    const uint32_t my_slot = stager ? sSlot[tid] : 0u;
    const bool staged_any = m16 != 0u;
    if (stager && round > 0) {
      load_rec(id_next);
      id_next = point_list[start + pos_of(round - 2)];
    }
    __syncthreads();
    // steps at which two of this wave's four lists hold the same instance (lane i looks at step i and at step 64 + i)
    unsigned long long clash_lo, clash_hi;
    {
      const uint16_t* l0 = &sList[4 * (2 * (wid >> 1)) + 2 * (wid & 1)][0];       // mini-blocks (r, c), (r, c+1),
      const uint16_t* l1 = l0 + BLIST;                                              // (r+1, c), (r+1, c+1)
      const uint16_t* l2 = l0 + 4 * BLIST;
      const uint16_t* l3 = l2 + BLIST;
      auto clash_step = [&](int i) {
        const uint32_t a = l0[i], b = l1[i], c = l2[i], d = l3[i];
        const uint32_t dm = 16u * BDUMMY;
        return (a != dm && (a == b || a == c || a == d)) || (b != dm && (b == c || b == d)) || (c != dm && c == d);
      };
      clash_lo = __builtin_amdgcn_ballot_w64(clash_step(lane));
      clash_hi = __builtin_amdgcn_ballot_w64(clash_step(64 + lane));
    }
    // ---- (d) every 16-lane group walks its own list, back to front -------------------------------------------------
    if (clamp) walk_lists_bwd<true>(mylist, nmax, base + 1u, (const char*)sA, (const char*)sB, (const char*)sC, slab, pxf, pyf, T,
                                    Bk, dpr, dpg, dpb, last, sel, acc_lane, grp, clash_lo, clash_hi);
    else       walk_lists_bwd<false>(mylist, nmax, base + 1u, (const char*)sA, (const char*)sB, (const char*)sC, slab, pxf, pyf, T,
                                     Bk, dpr, dpg, dpb, last, sel, acc_lane, grp, clash_lo, clash_hi);
    __syncthreads();
    // ---- (e) one row per staged instance: the four waves' sums in wave order -----------------------------------------
    if (staged_any) {
      float s9[9];
      bool nz = false;
#pragma unroll
      for (int v = 0; v < 9; ++v) {
        s9[v] = ((sAcc[0][tid][v] + sAcc[1][tid][v]) + sAcc[2][tid][v]) + sAcc[3][tid][v];
        nz = nz || s9[v] != 0.0f;
      }
      if (nz) {
        float4* dst = reinterpret_cast<float4*>(rows + my_slot);
        dst[0] = make_float4(s9[0], s9[1], -0.5f * s9[2], -s9[3]);      // Mx, My (first moments), dcxx, dcxy
        dst[1] = make_float4(-0.5f * s9[4], s9[5], s9[6], s9[7]);       // dcyy, dop, dr, dg
        dst[2] = make_float4(s9[8], 0.f, 0.f, 0.f);                     // db
        row_flags[my_slot] = 1;
      }
    }
    // the next round's prefill comes after this read of sAcc: one more barrier
    __syncthreads();
  }
}

void launch_render_fwd(int W, int H, const uint2* ranges, const uint32_t* point_list, const GeomRec* rec,
                       const float* bg, float* out_color, float* final_T, uint32_t* n_contrib, uint32_t* tile_max,
                       const uint32_t* tile_order, hipStream_t s, unsigned long long* stats) {
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  const int nblk = gx * gy;       // one 256-lane workgroup per tile, longest list first
  if (stats)
    hipLaunchKernelGGL((render_fwd_kernel<true, true>), dim3(nblk), dim3(256), 0, s, W, H, gx, gx * gy, tile_order, ranges,
                       point_list, rec, bg, out_color, final_T, n_contrib, tile_max, stats);
  else if (final_T && n_contrib && tile_max)
    hipLaunchKernelGGL((render_fwd_kernel<false, true>), dim3(nblk), dim3(256), 0, s, W, H, gx, gx * gy, tile_order, ranges,
                       point_list, rec, bg, out_color, final_T, n_contrib, tile_max, stats);
  else      // forward only: no per-pixel state for a backward
    hipLaunchKernelGGL((render_fwd_kernel<false, false>), dim3(nblk), dim3(256), 0, s, W, H, gx, gx * gy, tile_order, ranges,
                       point_list, rec, bg, out_color, final_T, n_contrib, tile_max, stats);
}
void launch_render_bwd(int W, int H, const uint2* ranges, const uint32_t* point_list, const GeomRec* rec,
                       const uint32_t* slot_base, const float* bg, const float* final_T, const uint32_t* n_contrib, const uint32_t* tile_max,
                       const float* dL_dpix, GradRow* rows, uint8_t* row_flags, const uint32_t* tile_order,
                       hipStream_t s) {
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  hipLaunchKernelGGL(render_bwd_kernel, dim3(gx * gy), dim3(256), 0, s, W, H, gx, gx * gy, tile_order, ranges, point_list, rec,
                     slot_base, bg, final_T, n_contrib, tile_max, dL_dpix, rows, row_flags);
}

}  // namespace gsr
