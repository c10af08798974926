// Shared device-side definitions for the gfx950 Gaussian rasterizer kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/gsr.h"

namespace gsr {

constexpr int TILE = 16;            // tile edge in pixels (SURVEY Appendix A constants)
constexpr int WAVE = 64;            // gfx950 wavefront
constexpr float NEAR_Z = 0.2f;
constexpr float DILATION = 0.3f;
constexpr float FOV_GUARD = 1.3f;
constexpr float ALPHA_MAX = 0.99f;
constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float T_STOP = 0.0001f;

// Per-Gaussian render record: exactly one 64-byte line, so the per-tile gather in the
// compositing kernels touches one line per instance.
struct __attribute__((aligned(64))) GeomRec {
  float x, y, cxx, cxy;          // pixel-space mean, conic xx / xy
  float cyy, opacity, r, g;      // conic yy, opacity, colour
  float b, ext_x, ext_y;         // colour, conservative half-extent of the alpha >= 1/255 ellipse
  uint32_t tile_mask;            // which tiles of the rect hold instances (see BinInfo::mask); the per-Gaussian row slot
                                 //  lives in GeomLayout::slot_base (a 4-byte write into these records is a line RMW)
  uint32_t rect_min;             // tile rect min: x | y << 16 | SH clamp mask (r,g,b) << 29   (x, y < 8192 tiles)
  uint32_t rect_wh;              // tile rect width | height << 16
  // completed-square form of the quadratic: q = cxx (dx + kk dy)^2 + isyy dy^2 -- every term >= 0 in float32, taken
  // from the covariance rather than from cxx cyy - cxy^2 (which cancels for elongated splats)
  float kk;                      // cxy / cxx = -cov_xy / cov_yy
  float isyy;                    // 1 / cov_yy = cyy - cxy^2 / cxx
};
constexpr uint32_t RECT_Y_MASK = 0x1fffu;
__host__ __device__ inline uint32_t rect_min_x(uint32_t rect_min) { return rect_min & 0xffffu; }
__host__ __device__ inline uint32_t rect_min_y(uint32_t rect_min) { return (rect_min >> 16) & RECT_Y_MASK; }
__host__ __device__ inline uint32_t rect_clamp_flags(uint32_t rect_min) { return rect_min >> 29; }
static_assert(sizeof(GeomRec) == 64, "GeomRec must be one cache line");

// Compact per-Gaussian binning input (read by duplicateWithKeys without touching GeomRec).
// mask: for rects of at most 32 tiles, bit (ty - y0) * w + (tx - x0) is set when the tile receives an instance
// (all w*h low bits unless GSR_BINNING_TWO_LEVEL_CULLED dropped tiles the alpha >= 1/255 ellipse cannot reach);
// larger rects always emit every tile and carry 0xffffffff.  Invisible: rect_wh = 0 and mask = 0.
struct __attribute__((aligned(16))) BinInfo {
  uint32_t rect_min;   // x | y << 16
  uint32_t rect_wh;    // w | h << 16
  float depth;
  uint32_t mask;
};
constexpr uint32_t MASK_TILES = 32;
constexpr uint32_t ROWS_COOP = 64;   // splats with more gradient rows than this are pre-summed by sum_big_rows_kernel
// instances (tiles_touched) of a Gaussian
__host__ __device__ inline uint32_t bin_count(uint32_t rect_wh, uint32_t mask) {
  const uint32_t n = (rect_wh & 0xffffu) * (rect_wh >> 16);
#if defined(__HIP_DEVICE_COMPILE__)
  return n <= MASK_TILES ? (uint32_t)__popc(mask) : n;
#else
  return n <= MASK_TILES ? (uint32_t)__builtin_popcount(mask) : n;
#endif
}
__host__ __device__ inline uint32_t full_mask(uint32_t n) { return n >= 32u ? 0xffffffffu : ((1u << n) - 1u); }
// rank of rect tile `bit` among the Gaussian's instances (its gradient-row slot relative to slot_base)
__device__ inline uint32_t bin_rank(uint32_t rect_wh, uint32_t mask, uint32_t bit) {
  const uint32_t n = (rect_wh & 0xffffu) * (rect_wh >> 16);
  return n <= MASK_TILES ? (uint32_t)__popc(mask & ((1u << bit) - 1u)) : bit;
}
// rect tile of instance k (k < bin_count)
__device__ inline uint32_t bin_kth(uint32_t rect_wh, uint32_t mask, uint32_t k) {
  const uint32_t n = (rect_wh & 0xffffu) * (rect_wh >> 16);
  if (n > MASK_TILES) return k;
  uint32_t m = mask;
  for (uint32_t r = 0; r < k; ++r) m &= m - 1u;
  return (uint32_t)__ffs((int)m) - 1u;
}

// The rect of a small splat in 31 bits -- the second word of the depth sort's payload, so that the instance emission
// streams instead of gathering BinInfo in depth order: first tile (13 bits: at most 8191 tiles), width - 1 (2 bits),
// tile mask (16 bits, bit = y * width + x).  Rects wider than four tiles or of more than 16 carry PACK_FALLBACK.
constexpr uint32_t PACK_FALLBACK = 0x80000000u;
constexpr int PACK_W_SHIFT = 13, PACK_MASK_SHIFT = 15;
__host__ __device__ inline uint32_t pack_rect(uint32_t rect_min, uint32_t rect_wh, uint32_t mask, uint32_t grid_x) {
  const uint32_t w = rect_wh & 0xffffu, h = rect_wh >> 16;
  const uint32_t tile0 = (rect_min >> 16) * grid_x + (rect_min & 0xffffu);
  if (w == 0u || w > 4u || w * h > 16u || tile0 >= (1u << PACK_W_SHIFT)) return PACK_FALLBACK;
  return tile0 | ((w - 1u) << PACK_W_SHIFT) | ((mask & 0xffffu) << PACK_MASK_SHIFT);
}
// back to (rect_min, rect_wh, mask) in the BinInfo convention; the height is the tallest a 16-bit mask of that width
// allows (16 / w rows: every set bit stays inside, and bin_count / bin_kth only look at the mask for such rects)
__host__ __device__ inline void unpack_rect(uint32_t packed, uint32_t grid_x, uint32_t& rect_min, uint32_t& rect_wh,
                                            uint32_t& mask) {
  const uint32_t tile0 = packed & ((1u << PACK_W_SHIFT) - 1u);
  const uint32_t w = ((packed >> PACK_W_SHIFT) & 3u) + 1u;
  const uint32_t y0 = tile0 / grid_x;
  rect_min = (tile0 - y0 * grid_x) | (y0 << 16);
  rect_wh = w | ((16u / w) << 16);
  mask = packed >> PACK_MASK_SHIFT;
}

// Per-instance gradient row written by the compositing backward, summed per Gaussian by
// the preprocess backward (atomic-free, bitwise reproducible).
// 36 bytes, packed (round 3: the three padding words of a 48-byte row were a quarter of what render_bwd stores and
// preprocess_bwd reads per instance -- 0.1 GB at C4, 0.33 GB on the heavy-tailed scene).
struct GradRow {
  float dmx, dmy, dcxx, dcxy;   // dmx, dmy: first moments sum h*(mean - pixel); preprocess_bwd applies the conic
  float dcyy, dop, dr, dg;
  float db;
};
static_assert(sizeof(GradRow) == 36, "GradRow");

__host__ __device__ inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

constexpr int PRE_BLOCK = 256;      // Gaussians per preprocess / duplicate block (shared so the
                                    // hierarchical scan lines up)

// ---- workspace layouts (host + device agree through these helpers) -------------------------
constexpr int SORT_THREADS = 512;
#ifndef GSR_SORT_ITEMS
#define GSR_SORT_ITEMS 8          // A/B builds only (tools/build_variant_all.sh -DGSR_SORT_ITEMS=4): items per lane of a sort block
#endif
constexpr int SORT_ITEMS = GSR_SORT_ITEMS;
constexpr int SORT_TILE = SORT_THREADS * SORT_ITEMS;   // keys per sort block
constexpr int RADIX_BITS = 9;   // widest digit: 45-bit keys at 1080p (13 tile bits + 32 depth bits) sort in 5 passes
constexpr int RADIX = 1 << RADIX_BITS;

struct SortLayout {
  size_t hist, totals, totals_odd, bytes;
  uint32_t nblocks;
  __host__ __device__ explicit SortLayout(uint32_t n) {
    nblocks = (n + SORT_TILE - 1) / SORT_TILE;
    if (nblocks == 0) nblocks = 1;
    nblocks += RADIX;          // a segmented last pass (binning.hip: seg_block) has up to one partial block per segment more
    size_t o = 0;
    hist = o;   o = align_up(o + 4 * (size_t)RADIX * nblocks, 256);
    totals = o; o = align_up(o + 4 * (size_t)RADIX, 256);        // digit totals of passes 0, 2, ...
    totals_odd = o; o = align_up(o + 4 * (size_t)RADIX, 256);    // ... and of passes 1, 3, ...: a segmented pass reads the
    bytes = o;                                                   //     totals of the pass before it while writing its own
  }
};

struct GeomLayout {
  size_t rec, bin, offsets, slot_base, block_sums, block_offs, block_vis, block_vis_offs, block_range, total;
  size_t dkey_a, dkey_b, didx_a, didx_b, dsort, big_list, block_big, block_big_offs, bytes;   // depth sort of the visible Gaussians (capacity P)
  int nblocks;
  __host__ __device__ explicit GeomLayout(int P) {
    nblocks = (P + PRE_BLOCK - 1) / PRE_BLOCK;
    size_t o = 0;
    rec = o;        o = align_up(o + sizeof(GeomRec) * (size_t)P, 256);
    bin = o;        o = align_up(o + sizeof(BinInfo) * (size_t)P, 256);
    offsets = o;    o = align_up(o + 4 * (size_t)P, 256);
    slot_base = o;  o = align_up(o + 4 * (size_t)P, 256);   // first gradient-row slot of every Gaussian (index-major)
    block_sums = o; o = align_up(o + 4 * (size_t)(nblocks + 1), 256);
    block_offs = o; o = align_up(o + 4 * (size_t)(nblocks + 1), 256);
    block_vis = o;  o = align_up(o + 4 * (size_t)(nblocks + 1), 256);        // visible Gaussians per block
    block_vis_offs = o; o = align_up(o + 4 * (size_t)(nblocks + 1), 256);
    block_range = o; o = align_up(o + 8 * (size_t)(nblocks + 1), 256);       // per block: max(~depth bits), max(depth bits)
    total = o;      o = align_up(o + 64, 256);                               // the TOTAL_* words below
    dkey_a = o;     o = align_up(o + 4 * (size_t)P, 256);
    dkey_b = o;     o = align_up(o + 4 * (size_t)P, 256);
    didx_a = o;     o = align_up(o + 8 * (size_t)P, 256);    // payload: (index, packed rect)
    didx_b = o;     o = align_up(o + 8 * (size_t)P, 256);
    dsort = o;      o = align_up(o + SortLayout((uint32_t)(P > 0 ? P : 1)).bytes, 256);
    // Gaussians with more than ROWS_COOP instances: block b of preprocess_fwd lists its own at big_list[b * PRE_BLOCK ..],
    // block_big[b] of them (no counter to zero, no atomics); the scan kernel turns the counts into offsets and
    // sum_big_rows numbers the entries of the frame through the offsets
    big_list = o;   o = align_up(o + 4 * (size_t)nblocks * PRE_BLOCK, 256);
    block_big = o;  o = align_up(o + 4 * (size_t)(nblocks + 1), 256);
    block_big_offs = o; o = align_up(o + 4 * (size_t)(nblocks + 1), 256);
    bytes = o;
  }
};

// words of GeomLayout::total
enum {
  TOTAL_R = 0,              // instances (sum of tiles_touched)
  TOTAL_V = 1,              // visible Gaussians
  TOTAL_BIG = 2,            // entries of big_list
  TOTAL_DEPTH_INV_MIN = 4,  // ~(smallest depth key of the frame)
  TOTAL_DEPTH_MAX = 5,      // largest depth key
  TOTAL_TOP_PASS_N = 6,     // V when the depth sort needs its top-digit pass, else 0
  TOTAL_R_CLAMPED = 7       // min(R, capacity of the binning workspace)
};
constexpr int DEPTH_SORT_BITS = 24;   // see gsr_launch.h

struct ImageLayout {
  size_t final_T, n_contrib, ranges, tile_max, tile_order, bytes;
  int grid_x, grid_y, tiles;
  __host__ __device__ ImageLayout(int W, int H) {
    grid_x = (W + TILE - 1) / TILE;
    grid_y = (H + TILE - 1) / TILE;
    tiles = grid_x * grid_y;
    size_t o = 0;
    size_t px = (size_t)W * H;
    final_T = o;   o = align_up(o + 4 * px, 256);
    n_contrib = o; o = align_up(o + 4 * px, 256);
    ranges = o;    o = align_up(o + 8 * (size_t)tiles, 256);
    tile_max = o;  o = align_up(o + 4 * (size_t)tiles, 256);
    tile_order = o; o = align_up(o + 4 * (size_t)tiles, 256);   // tiles, longest list first
    bytes = o;
  }
};

// Binning workspace.  mode 0 (default): two-level binning -- visible Gaussians are sorted by depth (u32 key),
// instances are emitted in depth order and stably partitioned by tile id (u32 key); mode 1: upstream's layout,
// one (u64 tile<<32|depth, u32 index) pair per instance sorted on 32 + tile bits.
struct BinLayout {
  // mode 1
  size_t keys_a, keys_b, vals_a, vals_b;
  // mode 0
  size_t bsum2, boffs2, itile_a, itile_b, ig_a, ig_b;
  size_t sort, bytes;
  uint32_t nblocks2;
  __host__ __device__ BinLayout(uint32_t R, uint32_t V, int mode) {
    const size_t n = R ? R : 1, v = V ? V : 1;
    nblocks2 = (uint32_t)((v + PRE_BLOCK - 1) / PRE_BLOCK);
    size_t o = 0;
    keys_a = keys_b = vals_a = vals_b = 0;
    bsum2 = boffs2 = itile_a = itile_b = ig_a = ig_b = 0;
    if (mode == 1) {
      keys_a = o; o = align_up(o + 8 * n, 256);
      keys_b = o; o = align_up(o + 8 * n, 256);
      vals_a = o; o = align_up(o + 4 * n, 256);
      vals_b = o; o = align_up(o + 4 * n, 256);
      sort = o;   o = align_up(o + SortLayout((uint32_t)n).bytes, 256);
    } else {
      bsum2 = o;  o = align_up(o + 4 * (size_t)(nblocks2 + 1), 256);
      boffs2 = o; o = align_up(o + 4 * (size_t)(nblocks2 + 1) + 64, 256);
      itile_a = o; o = align_up(o + 4 * n, 256);
      itile_b = o; o = align_up(o + 4 * n, 256);
      ig_a = o;   o = align_up(o + 4 * n, 256);
      ig_b = o;   o = align_up(o + 4 * n, 256);
      sort = o;   o = align_up(o + SortLayout((uint32_t)n).bytes, 256);
    }
    bytes = o;
  }
};

struct BwdLayout {
  size_t rows, flags, bytes;
  __host__ __device__ BwdLayout(int P, uint32_t R) {
    (void)P;
    size_t n = R ? R : 1;
    size_t o = 0;
    rows = o;  o = align_up(o + sizeof(GradRow) * n, 256);
    flags = o; o = align_up(o + n, 256);
    bytes = o;
  }
};

__host__ __device__ inline int tile_bits(int tiles) {
  int b = 0;
  while ((1 << b) < tiles) ++b;   // ceil(log2 T): upstream getHigherMsb equivalent for the sort range
  return b;
}

// key bits of the two-level binning's tile sort: at least one, so that a pass always runs (a one-tile image too) and the
// digit totals, from which the tile ranges are derived, exist
__host__ __device__ inline int tile_sort_bits(int tiles) { const int b = tile_bits(tiles); return b < 1 ? 1 : b; }

// Smallest value of q(d) = cxx dx^2 + 2 cxy dx dy + cyy dy^2 over the rectangle [x0,x1] x [y0,y1] of offsets
// d = p - mean.  q is convex with its minimum (0) at the mean: inside the rectangle the answer is 0, otherwise the
// segment from any point of the rectangle to the mean lowers q all the way and leaves the rectangle through an edge
// that faces the mean -- so only the (at most two) facing edges need their 1-D minimum.
__device__ inline float qmin_rect(float cxx, float cxy, float cyy, float icxx, float icyy, float dx0, float dx1,
                                  float dy0, float dy1) {
#pragma clang fp contract(fast)   // also inside translation units built with -ffp-contract=off: a conservative test
  const float xe = fminf(dx1, fmaxf(dx0, 0.0f)), ye = fminf(dy1, fmaxf(dy0, 0.0f));   // rectangle point nearest the mean
  // vertical facing edge x = xe (exists when xe != 0): minimise over y; horizontal facing edge y = ye likewise
  const float yv = fminf(dy1, fmaxf(dy0, -cxy * xe * icyy));
  const float xh = fminf(dx1, fmaxf(dx0, -cxy * ye * icxx));
  const float qv = cxx * xe * xe + (2.0f * cxy * xe + cyy * yv) * yv;
  const float qh = cyy * ye * ye + (2.0f * cxy * ye + cxx * xh) * xh;
  // xe == 0: the mean lies within the x-range and the vertical "edge" degenerates to the line through the mean, whose
  // minimum over [y0,y1] is q(0, ye) >= the horizontal edge's minimum -- taking the min with it is harmless
  return (xe == 0.0f && ye == 0.0f) ? 0.0f : fminf(xe != 0.0f ? qv : qh, ye != 0.0f ? qh : qv);
}

// ||dL_dmeans2D.xy|| of the densification statistics: one fixed operation order (an explicit fma), so that the fused
// epilogue of preprocess_bwd (built without contraction) and the stand-alone kernel give the same bits
__device__ inline float densify_grad_norm(float gx, float gy) { return sqrtf(__builtin_fmaf(gx, gx, gy * gy)); }

// ---- wave helpers --------------------------------------------------------------------------
__device__ inline int lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0)); }

__device__ inline uint32_t wave_incl_scan_u32(uint32_t v) {
  const int lane = lane_id();
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) {
    uint32_t o = __shfl_up(v, d, WAVE);
    if (lane >= d) v += o;
  }
  return v;
}

__device__ inline uint32_t wave_reduce_add_u32(uint32_t v) {
#pragma unroll
  for (int d = WAVE / 2; d > 0; d >>= 1) v += __shfl_xor(v, d, WAVE);
  return v;
}

__device__ inline float wave_reduce_add_f32(float v) {
#pragma unroll
  for (int d = WAVE / 2; d > 0; d >>= 1) v += __shfl_xor(v, d, WAVE);
  return v;
}

}  // namespace gsr
