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
// BACKWARD: one wave per tile; every 4-lane DPP bank owns one 4x4 mini-block (a 2x2 pixel quad per lane) and walks its own
// list of the round's instances that reach it -- the reach masks are the forward's, handed over in list order.  The nine
// per-pair gradient sums are folded over the bank with DPP into an LDS slot per (list, position); the staging lane of an
// instance sums its slots in mini-block order and writes one row per (Gaussian, tile) instance -- no atomics;
// preprocess_bwd sums the rows.
#include "gsr_common.h"
#include "gsr_launch.h"
#include <type_traits>

namespace gsr {

constexpr int BATCH = WAVE;   // instances staged per round of the backward
constexpr int WAVES_PER_BLOCK = 4;
constexpr float LOG2E = 1.4426950408889634f;

// What one lane fetches for the instance it stages (GeomRec words 0..10, 14..15 and, for the backward, 11..13).
struct Staged {
  float4 q0, q1, q2;
  float kk, isyy;
  uint32_t rect_min, rect_wh, slot_base;
};

// The forward's staging loads skip the words it never reads -- 3 (cxy: kk replaces it), 4 (cyy) and 11 (tile_mask): a dead
// destination register of an in-flight load gets recycled by the compiler and forces an early s_waitcnt vmcnt right
// behind the prefetch, which exposes the whole gather latency.  (The backward has its own loader for the same reason.)
__device__ inline void load_staged(const GeomRec* __restrict__ rec, uint32_t id, Staged& s) {
  const float* f0 = reinterpret_cast<const float*>(rec + id);
  const float* f = f0 + 4;
  s.q0.x = f0[0];
  s.q0.y = f0[1];
  s.q0.z = f0[2];
  s.q1.y = f[1];
  s.q1.z = f[2];
  s.q1.w = f[3];
  s.q2.x = f[4];
  s.q2.y = f[5];
  s.q2.z = f[6];
  const float2 k = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(rec + id) + 56);
  s.kk = k.x;
  s.isyy = k.y;
}

// LDS image of a staged instance.  The quadratic form is kept as a completed square, pre-scaled by log2(e) so that
// the exponential is a bare v_exp_f32:
//    log2(e) * power = nka * u^2 + nkd * dy^2,   u = dx + kk * dy,   (dx, dy) = mean - pixel,
//    nka = -0.5 log2e cxx,  kk = cxy / cxx,  nkd = -0.5 log2e / cov_yy.
// Both terms are <= 0 in float32 whatever the rounding, so the reference's "power > 0 -> skip" guard (Appendix A.4)
// can never fire -- it only ever fired on rounding noise of the expanded form -- and costs no compare here.
// The opacity rides in the exponent: alpha = opacity * exp(power) = exp2(log2(opacity) + log2e * power) -- the addend
// of a multiply that becomes an fma, one instruction less per pair than the product (v_log_f32 is good to an ulp, the
// sum is at most ~8 in magnitude: alpha moves by < 5e-7 relative).
// Whether an instance can reach the 0.99 clamp (opacity > 0.99; alpha <= opacity otherwise, to the ulp of v_log / v_exp)
// rides in the SIGN of the stored nkd: the evaluation takes -|nkd| (source modifiers: free), the clamped copies of the
// walks read the sign.  The clamp is thus a property of the INSTANCE: which copy of a walk runs depends on what else
// shares the round (256 instances in the forward, 64 in the backward), and an instance must get the same alpha in both
// kernels whatever its neighbours are.
struct LdsRec {
  float4 A;   // x, y, nka, kk
  float4 B;   // +-|nkd| (+: opacity > 0.99), log2(opacity), r, g
};
__device__ inline void make_lds(const Staged& st, LdsRec& o) {
  // -|cxx|: a covariance whose float32 determinant came out negative (a needle thousands of pixels long) yields a
  // negative conic; upstream's power > 0 guard drops most pairs of such a splat, here it composes with |cxx| instead.
  // What matters is that alpha stays <= opacity: an alpha above 1 would un-park a saturated pixel (T (1 - alpha) > 0)
  o.A = make_float4(st.q0.x, st.q0.y, (-0.5f * LOG2E) * fabsf(st.q0.z), st.kk);
  const float nkd_abs = (0.5f * LOG2E) * fabsf(st.isyy);
  o.B = make_float4(st.q1.y > ALPHA_MAX ? nkd_abs : -nkd_abs, __builtin_amdgcn_logf(st.q1.y), st.q1.z, st.q1.w);
}
__device__ __forceinline__ float clamp_alpha(float alpha, float nkd_signed) {
  return nkd_signed > 0.0f ? fminf(ALPHA_MAX, alpha) : alpha;
}
// log2(alpha before the clamp).  The same five operations in the forward and in the backward: both must take the same
// alpha >= 1/255 decisions
__device__ __forceinline__ float pair_p2(float dx, float dy, float nka, float kk, float nkd_signed, float lo) {
#pragma clang fp contract(off)
  const float u = __builtin_fmaf(kk, dy, dx);
  const float s = nka * u;
  const float v = __builtin_fmaf(-fabsf(nkd_signed) * dy, dy, lo);
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
// `token` is what a contributing pair leaves in `last` (the list entry: converted to a list position once per round).
template <bool TRACK, bool CLAMP>
__device__ __forceinline__ void blend_pair(const float4 a, const float4 b, const float cb, const uint32_t token,
                                           const float pxf, const float pyf, float& T, float& Cr, float& Cg, float& Cb,
                                           uint32_t& last) {
  const float dx = a.x - pxf, dy = a.y - pyf;
  float alpha = __builtin_amdgcn_exp2f(pair_p2(dx, dy, a.z, a.w, b.x, b.y));
  if (CLAMP) alpha = clamp_alpha(alpha, b.x);
  if (alpha >= ALPHA_MIN) {
    const float test_T = __builtin_fmaf(-alpha, T, T);      // T (1 - alpha), rounded once
    if (!(test_T < T_STOP)) {
      const float w = alpha * T;
      Cr = fmaf(b.z, w, Cr);
      Cg = fmaf(b.w, w, Cg);
      Cb = fmaf(cb, w, Cb);
      if (TRACK) last = token;
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
__device__ __forceinline__ void walk_lists(const uint16_t* __restrict__ mylist, const uint32_t nmax,
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
    blend_pair<TRACK, CLAMP>(a0, b0, c0, e0, pxf, pyf, T, Cr, Cg, Cb, last);
    a0 = a1; b0 = b1; c0 = c1;
    e0 = e1; e1 = e2;
  }
}

template <bool STATS, bool TRACK>
__global__ __launch_bounds__(256, TRACK ? 7 : 8) void render_fwd_kernel(int W, int H, int grid_x, int cull_miniblocks,
                                                         const uint32_t* __restrict__ tile_order,
                                                         const uint2* __restrict__ ranges,
                                                         const uint32_t* __restrict__ point_list,
                                                         const GeomRec* __restrict__ rec,
                                                         const float* __restrict__ bg,
                                                         float* __restrict__ out_color,
                                                         float* __restrict__ final_T,
                                                         uint32_t* __restrict__ n_contrib,
                                                         uint32_t* __restrict__ tile_max,
                                                         uint16_t* __restrict__ inst_mask,
                                                         unsigned long long* __restrict__ stats) {
  __shared__ float4 sA[QROUND + 1];
  __shared__ float4 sB[QROUND + 1];
  __shared__ float sC[QROUND + 4];
  __shared__ __attribute__((aligned(16))) uint16_t sList[16][QLIST];
  __shared__ uint4 sCnt[4];             // per wave: counts of the 16 mini-blocks, one byte each
  __shared__ uint32_t sFlag[2][4];      // [0] any opacity > ALPHA_MAX in the round, [1] tile_max partials
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
    sB[QDUMMY] = make_float4(0.f, -__builtin_inff(), 0.f, 0.f);      // log2(opacity) = -inf: alpha = exp2(-inf) = 0
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
    load_staged(rec, point_list[start + min((uint32_t)tid, len - 1)], st);
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
    uint32_t m16 = have ? miniblock_mask(st.q0.x - tx0, st.q0.y - ty0, st.q0.z, st.kk, st.isyy, st.q2.y, st.q2.z) : 0u;
    if (!cull_miniblocks) m16 = have ? 0xffffu : 0u;      // debug (GSR_DEBUG_NO_MINIBLOCK_CULL): every pair is evaluated
    // the backward walks the same (instance, mini-block) pairs: it reads the masks instead of computing them again
    if (TRACK && have) inst_mask[start + base + tid] = (uint16_t)m16;
    // ---- (b) wave-level ranks: four packed registers of four 8-bit counters --------------------------------------
    uint32_t own[4], incl[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      own[r] = (((m16 >> (4 * r)) & 0xfu) * 0x00204081u) & 0x01010101u;
      incl[r] = wave_incl_scan_dpp(own[r]);
    }
    if (lane == WAVE - 1) sCnt[wid] = make_uint4(incl[0], incl[1], incl[2], incl[3]);
    const bool opaque = have && st.q1.y > ALPHA_MAX;
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
    // a mini-block whose 16 pixels are all parked (saturated, or outside the image) has nothing left to composite
    {
      const unsigned long long alive = __builtin_amdgcn_ballot_w64(T > 0.0f);
      if (((alive >> (16 * grp)) & 0xffffull) == 0ull) n_lane = 0;
    }
    const uint32_t nmax = max(max((uint32_t)__builtin_amdgcn_readlane((int)n_lane, 0), (uint32_t)__builtin_amdgcn_readlane((int)n_lane, 16)),
                              max((uint32_t)__builtin_amdgcn_readlane((int)n_lane, 32), (uint32_t)__builtin_amdgcn_readlane((int)n_lane, 48)));
    // gathers of the next round (clamped indices: lanes past the end re-read the last instance and are masked by `have`)
    load_staged(rec, id_next, st);
    id_next = point_list[start + min(base + 2u * QROUND + (uint32_t)tid, len - 1)];
    __syncthreads();
    // ---- (d) every 16-lane group walks its own list --------------------------------------------------------------
    if (STATS) { st_pairs += (q == 0) ? n_lane : 0u; st_evals += (lane == 0) ? nmax : 0u; }
    uint32_t last_e = 0xffffffffu;        // list entry (16 * slot) of the round's last contributor
    if (clamp) walk_lists<TRACK, true>(mylist, nmax, (const char*)sA, (const char*)sB, (const char*)sC, pxf, pyf, T, Cr, Cg, Cb, last_e);
    else       walk_lists<TRACK, false>(mylist, nmax, (const char*)sA, (const char*)sB, (const char*)sC, pxf, pyf, T, Cr, Cg, Cb, last_e);
    if (TRACK && last_e != 0xffffffffu) last = base + 1u + (last_e >> 4);
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

constexpr int MB_DUMMY = WAVE;                 // LDS slot of the all-zero record the lists are padded with
// ------------------------------------------------------------------------------------------------------------------
// Backward, sixteen lists per wave: every 4-lane DPP bank owns one 4x4 MINI-BLOCK (lane j of bank mb holds the 2x2 quad
// (2 (j & 1), 2 (j >> 1)) of mini-block mb, mb = 4 * block row + block column as miniblock_mask numbers them) and walks
// its own list: one wave instruction works on SIXTEEN (instance, mini-block) pairs.  For 7-pixel footprints a pair of a
// 4x4 block has half of its pixels above the alpha threshold (an 8x8 pair: a quarter), so the walk needs 0.64x the steps
// (tools/sim_bwd_lists.py) at the same cost per step.
// With sixteen banks the same instance is met by several banks in the same step all the time, so the banks do not add
// into a shared row: the fold over the bank's four lanes goes to an LDS slot of the pair's own, and after the walk the
// staging lane of every instance sums its slots in mini-block order (float; a fixed order per instance, so the row does
// not depend on what else shares the round).  Slots are handed out per pass by a prefix sum over the instances' pair
// counts (an instance's slots are contiguous); there are DYN_CAP of them and MB_WIN positions per list: a round that
// needs more is done in passes, a pass taking the instances from the back for which everything still fits (an
// instance's pairs are never split over passes).  13.3 KB of LDS per wave: three waves per SIMD, which the kernel needs
// (profiles/r03/ab_render_bwd_occupancy_mb16.txt, ab_bwd_mb16_dynamic_slots.txt).
// ------------------------------------------------------------------------------------------------------------------
#ifdef BWD_PROFILE
// variant build only (tools/bwd_profile.py): shader-clock cycles per section, summed over all waves
__device__ unsigned long long g_bwd_prof[8];
#define PROF_T(x) const unsigned long long x = __builtin_readcyclecounter()
#define PROF_ADD(k, a, b) prof[k] += (unsigned long long)((b) - (a))
#define PROF_CNT(k, n) prof[k] += (unsigned long long)(n)
#else
#define PROF_T(x)
#define PROF_ADD(k, a, b)
#define PROF_CNT(k, n)
#endif
constexpr int MB_WIN = 26;                   // list positions a pass can hold
constexpr int MB_LIST = MB_WIN + 4;          // list row: [0] takes the writes of the lanes that are not in the list, then the
                                             // entries, two entries of read-ahead: 60 bytes
constexpr int DYN_CAP = 253;                 // pairs a pass can hold: slot index and record index share a 16-bit list entry,
                                             // a byte each
struct MbLds {
  struct Rec { float4 A, B; float C; float pad[3]; } R[WAVE + 1];      // one 48-byte record: one address per list entry
  __attribute__((aligned(16))) uint16_t list[16][MB_LIST];
  // slots handed out per pass by a prefix sum over the instances' pair counts: [0] stays zero, 1 .. DYN_CAP are the pairs of
  // the pass (an instance's slots are contiguous, in mini-block order), DYN_CAP + 1 takes what the dummy steps write
  float4 s0[DYN_CAP + 2];
  float4 s1[DYN_CAP + 2];
  float s2[DYN_CAP + 2];
#ifdef BWD_LDS_PAD
  char occupancy_probe[BWD_LDS_PAD];
#endif
};
static_assert((16 * MB_LIST * 2) % 16 == 0 && 16 * MB_LIST * 2 / 16 <= WAVE, "the list block is padded by one 16-byte store per lane");

// sums of nine values over the four lanes of every DPP bank (all four lanes get them): 18 operations.  volatile: a DPP
// operand is read from OTHER lanes, the statement must stay where the whole wave executes it
__device__ __forceinline__ void quad_fold9(float (&v)[9]) {
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %8, %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %4, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %5, %5, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %6, %6, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %7, %7, %7 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %8, %8, %8 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
      : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]));
}

__device__ __forceinline__ void render_bwd_tile_mb16(const int tile, MbLds& L, int W, int H, int grid_x,
                                                     const uint2* __restrict__ ranges,
                                                     const uint32_t* __restrict__ point_list,
                                                     const GeomRec* __restrict__ rec,
                                                     const uint16_t* __restrict__ inst_mask,
                                                     const uint32_t* __restrict__ slot_base, const float* __restrict__ bg,
                                                     const float* __restrict__ final_T,
                                                     const uint32_t* __restrict__ n_contrib,
                                                     const uint32_t* __restrict__ tile_max,
                                                     const float* __restrict__ dL_dpix, GradRow* __restrict__ rows,
                                                     uint8_t* __restrict__ row_flags) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int mb = lane >> 2, j = lane & 3;
  const int tile_x = tile % grid_x, tile_y = tile / grid_x;
  const int px0 = tile_x * TILE + 4 * (mb & 3) + 2 * (j & 1), py0 = tile_y * TILE + 4 * (mb >> 2) + 2 * (j >> 1);
  // dx = mean - pixel is formed by ONE subtraction from the exact integer coordinate, as in the forward and the reference
  float pxf0 = (float)px0, pxf1 = (float)(px0 + 1), pyf0 = (float)py0, pyf1 = (float)(py0 + 1);
  asm volatile("" : "+v"(pxf0), "+v"(pxf1), "+v"(pyf0), "+v"(pyf1));      // keep them in registers
  const size_t HW = (size_t)W * H;
  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
  // per pixel: T (running transmittance in front of the current instance) and U = (sum over the instances behind of
  // (c . dL_dpix) alpha T  +  T_final (bg . dL_dpix)) / T -- the colour the pixel shows behind the current instance,
  // dotted with dL_dpix.  With D = c . dL_dpix - U:  dL_dalpha = T D  and  U <- U + alpha D  (a convex combination: U
  // never leaves the range of the c . dL_dpix values, and two instructions per pair fewer than carrying the
  // un-normalised sum, which needs it divided by 1 - alpha)
  float T[4], U[4], dpr[4], dpg[4], dpb[4];
  uint32_t last[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int px = px0 + (k & 1), py = py0 + (k >> 1);
    const bool in = px < W && py < H;
    const size_t pix = (size_t)py * W + px;
    T[k] = in ? final_T[pix] : 0.0f;
    last[k] = in ? n_contrib[pix] : 0u;
    dpr[k] = in ? dL_dpix[pix] : 0.0f;
    dpg[k] = in ? dL_dpix[HW + pix] : 0.0f;
    dpb[k] = in ? dL_dpix[2 * HW + pix] : 0.0f;
    U[k] = bg0 * dpr[k] + bg1 * dpg[k] + bg2 * dpb[k];
  }
  const uint2 range = ranges[tile];
  const uint32_t start = range.x;
  uint32_t hi = min(range.y - range.x, tile_max[tile]);   // instances past the last contributor get no gradient
  // the same per mini-block: lane l < 16 keeps the last contributor of mini-block l (a mini-block behind an opaque
  // surface leaves the lists long before the rest of the tile)
  uint32_t mbl;
  {
    uint32_t mx = max(max(last[0], last[1]), max(last[2], last[3]));
    mx = max(mx, (uint32_t)__shfl_xor((int)mx, 1, WAVE));
    mx = max(mx, (uint32_t)__shfl_xor((int)mx, 2, WAVE));
    mbl = (uint32_t)__shfl((int)mx, (4 * lane) & (WAVE - 1), WAVE);
    if (lane >= 16) mbl = 0u;
  }
  if (lane == 0) {
    L.R[MB_DUMMY].A = make_float4(0.f, 0.f, 0.f, 0.f);
    L.R[MB_DUMMY].B = make_float4(0.f, -__builtin_inff(), 0.f, 0.f);      // log2(opacity) = -inf: alpha = 0
    L.R[MB_DUMMY].C = 0.0f;
  }
  if (lane == 0) {
    L.s0[0] = make_float4(0.f, 0.f, 0.f, 0.f);
    L.s1[0] = make_float4(0.f, 0.f, 0.f, 0.f);
    L.s2[0] = 0.0f;
  }
  const uint16_t* mylist = &L.list[mb][1];

  Staged st;
  st.q0 = st.q1 = st.q2 = make_float4(0.f, 0.f, 0.f, 0.f);
  st.kk = st.isyy = 0.0f;
  st.rect_min = st.rect_wh = st.slot_base = 0;
  // unconditional, index-clamped staging loads (see the forward kernel): ids two rounds ahead,
  // records one round ahead, walking the list back to front
  auto load_id = [&](uint32_t lo, uint32_t top) { return point_list[start + min(lo + lane, top - 1)]; };
  // only the words this kernel reads (not cxy, cyy, the extents): the destination register of a load nobody reads gets
  // recycled by the compiler, and the write to it then has to wait for the load -- right behind the prefetch, exposing
  // the whole gather latency (see load_staged)
  auto load_rec = [&](uint32_t id) {
    const char* base = reinterpret_cast<const char*>(rec + id);
    const float* f = reinterpret_cast<const float*>(base);
    st.q0.x = f[0]; st.q0.y = f[1]; st.q0.z = f[2];                  // mean x, y, conic xx
    st.q1.y = f[5]; st.q1.z = f[6]; st.q1.w = f[7]; st.q2.x = f[8];  // opacity, r, g, b
    st.q2.w = f[11];                                                 // tile_mask
    const uint4 rr = *reinterpret_cast<const uint4*>(base + 48);     // rect_min, rect_wh, kk, 1 / cov_yy
    st.rect_min = rr.x;
    st.rect_wh = rr.y;
    st.kk = __uint_as_float(rr.z);
    st.isyy = __uint_as_float(rr.w);
    st.slot_base = slot_base[id];
  };
  // the forward's mini-block reach mask of the instance at a list position (coalesced 2-byte reads, one round ahead)
  auto load_mask = [&](uint32_t lo, uint32_t top) { return (uint32_t)inst_mask[start + min(lo + lane, top - 1)]; };
  uint32_t mask_next = 0;
  uint32_t lo = hi > BATCH ? hi - BATCH : 0u;
  uint32_t id_next = 0;
  if (hi > 0) {
    load_rec(load_id(lo, hi));
    mask_next = load_mask(lo, hi);
    const uint32_t lo2 = lo > BATCH ? lo - BATCH : 0u;
    id_next = load_id(lo2, max(lo, 1u));
  }

  // Rows of the most recent pass: every instance sums the slots of its pairs, first to last (= in mini-block order; a lane
  // that has run out of pairs reads the zero slot while others go on) and writes one row if it received anything
  // (all-zero sums: no row, the flag byte stays 0).
  bool pending = false;
  uint32_t p_slot = 0u, p_base = 0u, p_cnt = 0u;      // row of this lane's instance, its first slot, number of slots
  uint32_t p_chunks = 1u;                              // (uniform) groups of four slots the fullest instance of the pass has
  float p_lo2op = 0.0f;
  auto flush_rows = [&]() {
    __builtin_amdgcn_wave_barrier();
    {
      float r9[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      // four slots per trip to LDS (twelve reads in flight, then the adds in slot order); p_chunks = how many trips the
      // instance with the most pairs needs
      auto chunk = [&](const uint32_t c0) {
        float4 f[4], g[4];
        float hh[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#ifdef BWD_EXP_FLUSH_SLOT0      // attribution build (WRONG rows): every lane reads the zero slot -- no gather, no conflicts
          const uint32_t at = (c0 + t < p_cnt) ? 0u : 0u;
#else
          const uint32_t at = (c0 + t < p_cnt) ? p_base + c0 + t : 0u;
#endif
          f[t] = L.s0[at];
          g[t] = L.s1[at];
          hh[t] = L.s2[at];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          r9[0] += f[t].x; r9[1] += f[t].y; r9[2] += f[t].z; r9[3] += f[t].w;
          r9[4] += g[t].x; r9[5] += g[t].y; r9[6] += g[t].z; r9[7] += g[t].w;
          r9[8] += hh[t];
        }
      };
      chunk(0u);
      if (p_chunks > 1u) chunk(4u);
      if (p_chunks > 2u) chunk(8u);
      if (p_chunks > 3u) chunk(12u);
      uint32_t bits = 0u;
#pragma unroll
      for (int t = 0; t < 9; ++t) bits |= __float_as_uint(r9[t]);
      const bool nz = (bits << 1) != 0u;      // lanes outside the pass read nothing
#ifdef BWD_EXP_NO_ROWWRITE      // timing attribution (WRONG): the sums are formed, one lane in a thousand writes its row
      if (nz && (bits & 0x3ffu) == 0x155u) {
#else
      if (nz) {
#endif
        GradRow t;
        t.dmx = r9[0]; t.dmy = r9[1]; t.dcxx = -0.5f * r9[2]; t.dcxy = -r9[3];
        t.dcyy = -0.5f * r9[4]; t.dop = r9[5] * __builtin_amdgcn_exp2f(-p_lo2op); t.dr = r9[6]; t.dg = r9[7];
        t.db = r9[8];
        rows[p_slot] = t;
        row_flags[p_slot] = 1;
      }
    }
    __builtin_amdgcn_wave_barrier();
  };
#ifdef BWD_PROFILE
  unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  PROF_T(t_tile0);
  while (hi > 0) {
    PROF_T(t_r0);
    PROF_CNT(4, 1);
    const uint32_t cur_lo = lo;
    const bool have = lo + lane < hi;
    // mini-blocks that still have a contributor in this round (1-based indices cur_lo + 1 .. hi)
    const uint32_t active = (uint32_t)__builtin_amdgcn_ballot_w64(mbl > cur_lo);
    const uint32_t m = have ? (mask_next & active) : 0u;
    // the 0.99 clamp can only be reached by an opacity above it (to the ulp of v_log / v_exp)
    const bool clamp = __builtin_amdgcn_ballot_w64(m != 0u && st.q1.y > ALPHA_MAX) != 0ull;
    // gradient-row slot of this instance: the Gaussian's first slot + the rank of this tile among its instances
    const uint32_t rw = st.rect_wh & 0xffffu;
    const uint32_t bit = ((uint32_t)tile_y - rect_min_y(st.rect_min)) * rw + ((uint32_t)tile_x - rect_min_x(st.rect_min));
    const uint32_t slot = st.slot_base + bin_rank(st.rect_wh, __float_as_uint(st.q2.w), have ? bit : 0u);
    LdsRec lr;
    make_lds(st, lr);
    const float lo2op = lr.B.y;                      // log2(opacity): 1 / opacity for the row at the end of the pass
    __builtin_amdgcn_wave_barrier();
    L.R[lane].A = lr.A;
    L.R[lane].B = lr.B;
    L.R[lane].C = st.q2.x;
    hi = lo;
    lo = hi > BATCH ? hi - BATCH : 0u;
    {   // prefetch: record and mask of the next (earlier) round, ids of the one after it
      load_rec(id_next);
      mask_next = load_mask(lo, max(hi, 1u));
      const uint32_t lo2 = lo > BATCH ? lo - BATCH : 0u;
      id_next = load_id(lo2, max(lo, 1u));
    }
    unsigned long long remaining = __builtin_amdgcn_ballot_w64(m != 0u);
    PROF_T(t_r1);
    PROF_ADD(0, t_r0, t_r1);
    while (remaining != 0ull) {
      PROF_T(t_p0);
      PROF_CNT(5, 1);
      // ---- which instances this pass takes, their list positions ---------------------------------------------------
      const bool rem = (remaining >> lane) & 1ull;
      const uint32_t mr = rem ? m : 0u;
      // sfx[r], byte k: members of list 4 r + k among the remaining lanes >= this one (back to front: the highest lane
      // is the first entry).  Four packed 8-bit counters per register, as in the forward.
      uint32_t sfx[4], own[4], over = 0u;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        own[r] = (((mr >> (4 * r)) & 0xfu) * 0x00204081u) & 0x01010101u;
        const uint32_t incl = wave_incl_scan_dpp(own[r]);
        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, WAVE - 1);
        sfx[r] = tot - incl + own[r];
        over |= sfx[r] + (uint32_t)(127 - MB_WIN) * 0x01010101u;      // bit 7 of a byte: that list is past the window here
      }
      // slots: the pairs of the remaining instances above this one, plus its own, have to fit
      const uint32_t cnt = (uint32_t)__popc(mr);
      const uint32_t cincl = wave_incl_scan_dpp(cnt);
      const uint32_t above = (uint32_t)__builtin_amdgcn_readlane((int)cincl, WAVE - 1) - cincl;
      const bool inA = rem && (over & 0x80808080u) == 0u && above + cnt <= (uint32_t)DYN_CAP;
      const unsigned long long balA = __builtin_amdgcn_ballot_w64(inA);
      // list lengths of the pass = the counters at its lowest lane; the longest one sets the number of steps
      uint32_t nmax;
      {
        const int cA = __builtin_ctzll(balA);
        const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)sfx[0], cA), c1 = (uint32_t)__builtin_amdgcn_readlane((int)sfx[1], cA);
        const uint32_t c2 = (uint32_t)__builtin_amdgcn_readlane((int)sfx[2], cA), c3 = (uint32_t)__builtin_amdgcn_readlane((int)sfx[3], cA);
        const uint32_t csel = (lane & 8) ? ((lane & 4) ? c3 : c2) : ((lane & 4) ? c1 : c0);
        uint32_t len = (csel >> (8 * (lane & 3))) & 0xffu;            // lanes 0..15: length of list `lane`
        len = max(len, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)len, 0x128, 0xf, 0xf, false));   // row_ror:8
        len = max(len, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)len, 0x124, 0xf, 0xf, false));   // row_ror:4
        len = max(len, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)len, 0x122, 0xf, 0xf, false));   // row_ror:2
        len = max(len, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)len, 0x121, 0xf, 0xf, false));   // row_ror:1
        nmax = (uint32_t)__builtin_amdgcn_readlane((int)len, 0);
      }
      // idx[r], byte k: 1 + this instance's position in list 4 r + k, 0 for a list it is not in (or not in this pass): the
      // row index of its list entry -- index 0 is the junk entry, so the list build needs no branch or predicate
      uint32_t idx[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) idx[r] = inA ? (sfx[r] & (own[r] * 0xffu)) : 0u;
      __builtin_amdgcn_wave_barrier();
      {   // pad the lists with the dummy record
        const uint32_t dd = ((uint32_t)MB_DUMMY | ((uint32_t)(DYN_CAP + 1) << 8)) * 0x10001u;
        uint4* l4 = reinterpret_cast<uint4*>(&L.list[0][0]);
        if (lane < 16 * MB_LIST * 2 / 16) l4[lane] = make_uint4(dd, dd, dd, dd);
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          // entry: record index in the low byte, slot in the high byte (first slot + rank of the mini-block in the mask)
          const uint32_t sl = above + 1u + (uint32_t)__popc(m & ((1u << (4 * r + k)) - 1u));
#ifdef BWD_EXP_LIST_PRED       // A/B build: members only (EXEC-masked store) instead of 60 lanes writing the junk entry
          const uint32_t pp = (idx[r] >> (8 * k)) & 0xffu;
          if (pp) L.list[4 * r + k][pp] = (uint16_t)((sl << 8) | (uint32_t)lane);
#else
          L.list[4 * r + k][(idx[r] >> (8 * k)) & 0xffu] = (uint16_t)((sl << 8) | (uint32_t)lane);
#endif
        }
      __builtin_amdgcn_wave_barrier();
      // two copies of the walk: the 0.99 clamp costs an instruction per pixel and almost no round needs it
      auto walk = [&](auto clamped_c) {
        constexpr bool CLAMPED = decltype(clamped_c)::value;
        auto fetch = [&](uint32_t e, float4& a, float4& b, float& cb) {
#ifdef BWD_EXP_REC_BANK         // attribution build (WRONG gradients): bank mb reads record mb -- no gather conflicts
          const char* rp = reinterpret_cast<const char*>(L.R) + 48u * ((e & 0u) + (uint32_t)mb);
#else
          const char* rp = reinterpret_cast<const char*>(L.R) + 48u * (e & 0xffu);
#endif
          a = *reinterpret_cast<const float4*>(rp);
          b = *reinterpret_cast<const float4*>(rp + 16);
          cb = *reinterpret_cast<const float*>(rp + 32);
        };
        // one (instance, mini-block) pair per bank: list entry e0 (its record: a, b, cb), list position i
        auto step = [&](const float4 a, const float4 b, const float cb, const uint32_t e0, const uint32_t i) {
          const uint32_t pos1 = cur_lo + (e0 & 0xffu) + 1u;
          // per-lane partial sums over the quad; un-scaled forms (constants applied to the row):
          //   v0 = sum h dx, v1 = sum h dy (first moments), v2 = sum h dx^2, v3 = sum h dx dy, v4 = sum h dy^2 with
          //   h = opacity*G*dL_dalpha;  v5 = sum h (= opacity * dL_dopacity);  v6..8 = sum alpha*T*dL_dpix
          float v[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          float S = 0.f, Sx = 0.f, Sy = 0.f, Sxy = 0.f;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            // (no wave-wide skip of a quad position whose 64 pixels are all past their last contributor: with sixteen
            // mini-blocks in the wave that never happens, and the branch kept the four evaluations from being interleaved:
            // -5 % without it, profiles/r03/ab_bwd_mb16.txt)
            const float dx = a.x - ((k & 1) ? pxf1 : pxf0), dy = a.y - ((k >> 1) ? pyf1 : pyf0);
            const float ar = __builtin_amdgcn_exp2f(pair_p2(dx, dy, a.z, a.w, b.x, b.y));      // opacity * G
            // (the clamp is above the threshold: same test on either)
            const bool ok = pos1 <= last[k] && ar >= ALPHA_MIN;
            // lanes that do not contribute run the same instructions on alpha = 0: 1 / (1 - 0) = 1 and every product is 0
            const float arm = ok ? ar : 0.0f;
            const float am = CLAMPED ? clamp_alpha(arm, b.x) : arm;
            const float rcp = __builtin_amdgcn_rcpf(1.0f - am);
            T[k] *= rcp;                                   // transmittance in front of this instance
            const float D = fmaf(cb, dpb[k], fmaf(b.w, dpg[k], fmaf(b.z, dpr[k], -U[k])));   // c . dL_dpix - U
            const float dch = am * T[k];
            // opacity * G * dL_dalpha with dL_dalpha = T D: the clamp passes the gradient on (arm, not am).  Same
            // association in both copies of the walk: which one an instance meets depends on what else is in its round
            const float h = (CLAMPED ? arm * T[k] : dch) * D;
            U[k] = fmaf(am, D, U[k]);
            // moments of h about the quad's first pixel: the offsets of the other three are 0 / 1, so h itself is all that
            // is added per pixel (dx_k = dx_0 - (k & 1) and dy_k = dy_0 - (k >> 1) exactly: all four differences are exact)
            S += h;
            if (k & 1) Sx += h;
            if (k >> 1) Sy += h;
            if (k == 3) Sxy = h;
            v[6] = fmaf(dch, dpr[k], v[6]); v[7] = fmaf(dch, dpg[k], v[7]); v[8] = fmaf(dch, dpb[k], v[8]);
          }
          {   // sum h dx = dx0 S - Sx, sum h dx^2 = dx0 (dx0 S - 2 Sx) + Sx, sum h dx dy = dx0 (dy0 S - Sy) - dy0 Sx + Sxy, ...
            const float dx0 = a.x - pxf0, dy0 = a.y - pyf0;
            v[0] = fmaf(dx0, S, -Sx);
            v[1] = fmaf(dy0, S, -Sy);
            v[2] = fmaf(dx0, v[0] - Sx, Sx);
            v[4] = fmaf(dy0, v[1] - Sy, Sy);
            v[3] = fmaf(dx0, v[1], fmaf(-dy0, Sx, Sxy));
            v[5] = S;
          }
          quad_fold9(v);
          // the pair's nine sums: one slot per (list, position).  After the fold the bank's four lanes hold the same nine
          // values and all four write them (same address, same data): no EXEC change, the step stays one basic block
          // (-3 %, profiles/r03/ab_bwd_mb16_step.txt; two steps per iteration in one block: no better, the odd lengths cost
          // a step)
#ifdef BWD_EXP_SLOT_LANE        // attribution build (WRONG rows): every bank writes a slot of its own -- no store conflicts
          L.s0[1 + mb] = make_float4(v[0], v[1], v[2], v[3]);
          L.s1[1 + mb] = make_float4(v[4], v[5], v[6], v[7]);
          L.s2[1 + mb] = v[8];
#else
          L.s0[e0 >> 8] = make_float4(v[0], v[1], v[2], v[3]);
          L.s1[e0 >> 8] = make_float4(v[4], v[5], v[6], v[7]);
          L.s2[e0 >> 8] = v[8];
#endif
        };
        uint32_t e0 = mylist[0], e1 = mylist[1];
        float4 a, b, an, bn;
        float cb, cbn;
        fetch(e0, a, b, cb);
        for (uint32_t i = 0; i < nmax; ++i) {
          fetch(e1, an, bn, cbn);                       // the next pair's record, while this one is evaluated
          const uint32_t e2 = mylist[i + 2];
          step(a, b, cb, e0, i);
          e0 = e1; e1 = e2;
          a = an; b = bn; cb = cbn;
        }
      };
      PROF_T(t_p1);
      PROF_ADD(1, t_p0, t_p1);
      PROF_CNT(6, nmax);
#ifndef BWD_EXP_NO_FLUSH         // timing attribution (WRONG): no row sums, no rows
      if (pending) flush_rows();
#endif
      PROF_T(t_p1b);
      PROF_ADD(3, t_p1, t_p1b);
      if (clamp) walk(std::true_type{}); else walk(std::false_type{});
      __builtin_amdgcn_wave_barrier();
      PROF_T(t_p2);
      PROF_ADD(2, t_p1b, t_p2);
      // the rows of this pass are summed and written when its slots are about to be reused (flush_rows): the stores
      // are then in flight during a walk instead of in front of the next round's wait for its records
      pending = true;
      p_slot = slot;
      p_lo2op = lo2op;
      p_base = above + 1u;
      p_cnt = inA ? cnt : 0u;
      p_chunks = 1u + (__builtin_amdgcn_ballot_w64(p_cnt > 4u) != 0ull ? 1u : 0u) + (__builtin_amdgcn_ballot_w64(p_cnt > 8u) != 0ull ? 1u : 0u) +
                 (__builtin_amdgcn_ballot_w64(p_cnt > 12u) != 0ull ? 1u : 0u);
      remaining &= ~balA;
    }
  }
  if (pending) flush_rows();
#ifdef BWD_PROFILE
  {
    PROF_T(t_tile1);
    prof[7] = t_tile1 - t_tile0;
    if (lane == 0)
      for (int k = 0; k < 8; ++k) atomicAdd(&g_bwd_prof[k], prof[k]);
  }
#endif
}

// 13.3 KB of LDS per wave: three blocks of four waves per CU, three waves per SIMD
constexpr int BWD_WAVES = 3;
__global__ __launch_bounds__(WAVES_PER_BLOCK * WAVE, BWD_WAVES) void render_bwd_kernel(int W, int H, int grid_x, int num_tiles,
                                                          const uint32_t* __restrict__ tile_order,
                                                          const uint2* __restrict__ ranges,
                                                          const uint32_t* __restrict__ point_list,
                                                          const GeomRec* __restrict__ rec,
                                                          const uint16_t* __restrict__ inst_mask,
                                                          const uint32_t* __restrict__ slot_base,
                                                          const float* __restrict__ bg,
                                                          const float* __restrict__ final_T,
                                                          const uint32_t* __restrict__ n_contrib,
                                                          const uint32_t* __restrict__ tile_max,
                                                          const float* __restrict__ dL_dpix,
                                                          GradRow* __restrict__ rows,
                                                          uint8_t* __restrict__ row_flags) {
  __shared__ MbLds sL[WAVES_PER_BLOCK];
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
  const int slot = blockIdx.x * WAVES_PER_BLOCK + wid;
  if (slot >= num_tiles) return;
  const int tile = __builtin_amdgcn_readfirstlane((int)tile_order[slot]);
  render_bwd_tile_mb16(tile, sL[wid], W, H, grid_x, ranges, point_list, rec, inst_mask, slot_base, bg, final_T, n_contrib, tile_max,
                       dL_dpix, rows, row_flags);
}

void launch_render_fwd(int W, int H, const uint2* ranges, const uint32_t* point_list, const GeomRec* rec,
                       const float* bg, float* out_color, float* final_T, uint32_t* n_contrib, uint32_t* tile_max,
                       const uint32_t* tile_order, hipStream_t s, unsigned long long* stats, int cull, uint16_t* inst_mask) {
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  const int nblk = gx * gy;       // one 256-lane workgroup per tile, longest list first
  // cull = 0 (GsrParams.debug_flags & GSR_DEBUG_NO_MINIBLOCK_CULL, tests/test_gpu_miniblock_cull.py): every staged
  // instance enters all 16 lists; the image must not change by a bit (a dropped pair is a pair no pixel of which
  // passes the alpha test)
  if (stats)
    hipLaunchKernelGGL((render_fwd_kernel<true, true>), dim3(nblk), dim3(256), 0, s, W, H, gx, cull, tile_order, ranges,
                       point_list, rec, bg, out_color, final_T, n_contrib, tile_max, inst_mask, stats);
  else if (final_T && n_contrib && tile_max)      // inst_mask may be NULL only for a frame without instances
    hipLaunchKernelGGL((render_fwd_kernel<false, true>), dim3(nblk), dim3(256), 0, s, W, H, gx, cull, tile_order, ranges,
                       point_list, rec, bg, out_color, final_T, n_contrib, tile_max, inst_mask, stats);
  else      // forward only: no per-pixel state for a backward
    hipLaunchKernelGGL((render_fwd_kernel<false, false>), dim3(nblk), dim3(256), 0, s, W, H, gx, cull, tile_order, ranges,
                       point_list, rec, bg, out_color, final_T, n_contrib, tile_max, inst_mask, stats);
}
#ifdef BWD_PROFILE
extern "C" int gsr_debug_bwd_profile(unsigned long long* out8, int reset) {
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_bwd_prof), 64) != hipSuccess) return 2;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_bwd_prof), z, 64) != hipSuccess) return 3;
  }
  return 0;
}
#endif
void launch_render_bwd(int W, int H, const uint2* ranges, const uint32_t* point_list, const GeomRec* rec,
                       const uint32_t* slot_base, const float* bg, const float* final_T, const uint32_t* n_contrib, const uint32_t* tile_max,
                       const float* dL_dpix, GradRow* rows, uint8_t* row_flags, const uint32_t* tile_order,
                       hipStream_t s, const uint16_t* inst_mask) {
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  const int nblk = (gx * gy + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
  hipLaunchKernelGGL(render_bwd_kernel, dim3(nblk), dim3(WAVES_PER_BLOCK * WAVE), 0, s, W, H, gx, gx * gy, tile_order, ranges, point_list, rec,
                     inst_mask, slot_base, bg, final_T, n_contrib, tile_max, dL_dpix, rows, row_flags);
}

}  // namespace gsr
