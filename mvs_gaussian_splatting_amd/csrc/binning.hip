// Tile binning: second level of the tiles_touched scan fused into
// duplicateWithKeys, a stable LSD radix sort of (u64 key, u32 value) pairs, identifyTileRanges.
// All integer work: results are bit-exact against oracle/rasterizer_ref.py:bin_ref.
#include "gsr_common.h"
#include "gsr_launch.h"

namespace gsr {

// ------------------------------------------------------------------------------------------
// duplicateWithKeys (A.3).  Block b owns Gaussians [b*256, b*256+256) -- the same partition as
// the preprocess kernel, so block_offs[b] (exclusive scan of the block totals) + an in-block
// exclusive scan of `tiles` gives every Gaussian's first slot without a separate scan pass.
// Emission order: y outer, x inner; key = tile << 32 | bits(depth); value = Gaussian index.
// Gaussians covering more than MASK_TILES tiles are emitted by the whole wave cooperatively.
// ------------------------------------------------------------------------------------------

__global__ __launch_bounds__(PRE_BLOCK) void duplicate_with_keys_kernel(int P, int grid_x,
                                                                        const BinInfo* __restrict__ bin,
                                                                        const uint32_t* __restrict__ block_offs,
                                                                        uint32_t* __restrict__ slot_base,
                                                                        uint32_t* __restrict__ point_offsets,
                                                                        uint64_t* __restrict__ keys,
                                                                        uint32_t* __restrict__ vals) {
  __shared__ uint32_t wave_tot[PRE_BLOCK / WAVE];
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
  const int idx = blockIdx.x * PRE_BLOCK + tid;
  BinInfo bi{0u, 0u, 0.0f, 0u};
  if (idx < P) bi = bin[idx];
  const uint32_t tiles = bin_count(bi.rect_wh, bi.mask);
  const uint32_t inc = wave_incl_scan_u32(tiles);
  if (lane == WAVE - 1) wave_tot[wid] = inc;
  __syncthreads();
  uint32_t base = block_offs[blockIdx.x];
#pragma unroll
  for (int w = 0; w < PRE_BLOCK / WAVE; ++w)
    if (w < wid) base += wave_tot[w];
  const uint32_t off = base + inc - tiles;   // exclusive
  if (idx < P) {
    point_offsets[idx] = off + tiles;        // inclusive scan, as upstream's point_offsets
    slot_base[idx] = off;
  }
  const uint32_t x0 = bi.rect_min & 0xffffu, y0 = bi.rect_min >> 16;
  const uint32_t w = bi.rect_wh & 0xffffu;
  const uint64_t dbits = (uint64_t)__float_as_uint(bi.depth);

  const uint32_t area = w * (bi.rect_wh >> 16);
  if (tiles && area <= MASK_TILES) {             // small rects: per lane, honouring the tile mask
    uint32_t o = off, bit = 0;
    const uint32_t h = bi.rect_wh >> 16;
    for (uint32_t y = y0; y < y0 + h; ++y)
      for (uint32_t x = x0; x < x0 + w; ++x, ++bit) {
        if (!((bi.mask >> bit) & 1u)) continue;
        keys[o] = ((uint64_t)(y * (uint32_t)grid_x + x) << 32) | dbits;
        vals[o] = (uint32_t)idx;
        ++o;
      }
  }
  // wave-cooperative path for large splats (every tile of the rect)
  unsigned long long big = __ballot(area > MASK_TILES);
  while (big) {
    const int src = __ffsll((long long)big) - 1;
    big &= big - 1;
    const uint32_t s_tiles = __shfl(tiles, src, WAVE);
    const uint32_t s_off = __shfl(off, src, WAVE);
    const uint32_t s_min = __shfl(bi.rect_min, src, WAVE);
    const uint32_t s_w = __shfl(w, src, WAVE);
    const uint32_t s_dbits = __shfl(__float_as_uint(bi.depth), src, WAVE);
    const uint32_t s_idx = (uint32_t)(idx - lane + src);
    const uint32_t sx0 = s_min & 0xffffu, sy0 = s_min >> 16;
    for (uint32_t k = (uint32_t)lane; k < s_tiles; k += WAVE) {
      const uint32_t y = sy0 + k / s_w, x = sx0 + k % s_w;
      keys[s_off + k] = ((uint64_t)(y * (uint32_t)grid_x + x) << 32) | (uint64_t)s_dbits;
      vals[s_off + k] = s_idx;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Two-level binning (mode 0).  Same point lists and tile ranges as sorting 64-bit (tile, depth) keys, for
// ~2.5x less memory traffic: the per-instance data that goes through a multi-pass sort shrinks from 12 bytes x
// 5-6 passes to 8 bytes x 2 passes, because the depth order is established once per Gaussian, not per instance.
//   1. compact_visible      : (depth bits - nearest depth bits, (index, packed rect)) of the visible Gaussians, index order
//   2. sort by depth        : stable LSD radix sort of V (u32 key, 8-byte payload) pairs on 24 bits (gsr_launch.h:
//                             DEPTH_SORT_BITS; the top 8 bits in a fourth pass only when a frame needs them)
//   3. count_tiles + scan   : tiles_touched in depth order (from the payload) -> first slot of every 256 Gaussians
//   4. emit_instances       : (tile id, index) per overlapped tile, y outer / x inner, in depth order
//   5. sort by tile         : stable sort on ceil(log2 T) bits -> lists ordered by (tile, depth, index)
// Stability of both sorts reproduces the tie order of a stable 64-bit sort of index-ordered pairs.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PRE_BLOCK) void compact_visible_kernel(int P, const BinInfo* __restrict__ bin,
                                                                    const uint32_t* __restrict__ block_vis_offs,
                                                                    const uint32_t* __restrict__ block_offs,
                                                                    uint32_t* __restrict__ slot_base,
                                                                    const uint32_t* __restrict__ depth_inv_min,
                                                                    int grid_x,
                                                                    uint32_t* __restrict__ dkey,
                                                                    uint2* __restrict__ dval,
                                                                    uint32_t* __restrict__ total, uint32_t capacity,
                                                                    int top_pass_enqueued) {
  __shared__ uint32_t wave_tot[PRE_BLOCK / WAVE];
  __shared__ uint32_t wave_tiles[PRE_BLOCK / WAVE];
  const uint32_t min_bits = ~*depth_inv_min;      // smallest depth key of the frame: keys are sorted relative to it
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    // counts the later stages read from the device (total = [R, V, big, -, ~min depth key, max depth key, ...]):
    //   [6] elements of the depth sort's top-digit pass: V when the frame spans more than 2^DEPTH_SORT_BITS depth keys
    //       (the pass then runs and the sorted payload ends in the other buffer), else 0 (its kernels exit at once);
    //       always 0 when the caller did not enqueue the pass (GsrParams::depth_span_lt24): the consumers then read the
    //       buffer the three regular passes ended in, whatever the span -- the caller discards such a frame
    //   [7] instances the binning workspace really receives: min(R, capacity)
    const uint32_t R = total[TOTAL_R], V = total[TOTAL_V], mx = total[TOTAL_DEPTH_MAX];
    const bool top = top_pass_enqueued && V > 0u && mx >= min_bits && ((mx - min_bits) >> DEPTH_SORT_BITS) != 0u;
    total[TOTAL_TOP_PASS_N] = top ? V : 0u;
    total[TOTAL_R_CLAMPED] = min(R, capacity);
  }
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
  const int idx = blockIdx.x * PRE_BLOCK + tid;
  BinInfo bi{0u, 0u, 0.0f, 0u};
  if (idx < P) bi = bin[idx];
  const uint32_t tiles = bin_count(bi.rect_wh, bi.mask);
  const uint32_t vis = tiles != 0u;
  const uint32_t inc = wave_incl_scan_u32(vis);
  const uint32_t tinc = wave_incl_scan_u32(tiles);
  if (lane == WAVE - 1) { wave_tot[wid] = inc; wave_tiles[wid] = tinc; }
  __syncthreads();
  uint32_t base = block_vis_offs[blockIdx.x];
  uint32_t tbase = block_offs[blockIdx.x];
#pragma unroll
  for (int w = 0; w < PRE_BLOCK / WAVE; ++w)
    if (w < wid) { base += wave_tot[w]; tbase += wave_tiles[w]; }
  if (vis) {
    const uint32_t o = base + inc - 1u;
    dkey[o] = __float_as_uint(bi.depth) - min_bits;     // order-preserving (positive floats, all >= the minimum)
    // the payload of the depth sort: the index and, for the common small rect, everything the instance emission needs
    // (so that no random gather of BinInfo follows the sort)
    dval[o] = make_uint2((uint32_t)idx, pack_rect(bi.rect_min, bi.rect_wh, bi.mask, (uint32_t)grid_x));
  }
  // slot range of this Gaussian's per-instance gradient rows: index-major (any bijection works), coalesced write
  if (slot_base && idx < P) slot_base[idx] = tbase + tinc - tiles;     // NULL: forward only
}

// tiles of a depth-sorted Gaussian: from the packed rect that travelled with it, or (rare: rects wider than four tiles
// or of more than 16) from its BinInfo
__device__ inline uint32_t sorted_tiles(uint2 v, const BinInfo* __restrict__ bin) {
  if (v.y & PACK_FALLBACK) {
    const BinInfo bi = bin[v.x];
    return bin_count(bi.rect_wh, bi.mask);
  }
  return (uint32_t)__popc(v.y >> PACK_MASK_SHIFT);
}

// The depth-sorted payload lives in `d3` after the three regular passes and in `d4` when the top-digit pass ran
// (total[TOTAL_TOP_PASS_N] != 0); V is read from the device (the grid may be sized for a capacity).
__global__ __launch_bounds__(PRE_BLOCK) void count_tiles_kernel(const uint32_t* __restrict__ total,
                                                                const uint2* __restrict__ d3,
                                                                const uint2* __restrict__ d4,
                                                                const BinInfo* __restrict__ bin,
                                                                uint32_t* __restrict__ block_sums2) {
  __shared__ uint32_t wave_tot[PRE_BLOCK / WAVE];
  const uint32_t V = total[TOTAL_V];
  const uint2* __restrict__ dval = total[TOTAL_TOP_PASS_N] ? d4 : d3;
  const uint32_t i = blockIdx.x * PRE_BLOCK + threadIdx.x;
  const uint32_t t = i < V ? sorted_tiles(dval[i], bin) : 0u;
  const uint32_t ws = wave_reduce_add_u32(t);
  if ((threadIdx.x & (WAVE - 1)) == 0) wave_tot[threadIdx.x / WAVE] = ws;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t s = 0;
#pragma unroll
    for (int w = 0; w < PRE_BLOCK / WAVE; ++w) s += wave_tot[w];
    block_sums2[blockIdx.x] = s;
  }
}

// Load-balanced expansion: the block's 256 Gaussians own a contiguous run of `total` instances; every lane
// takes output slots (not Gaussians), finds the owning Gaussian by binary search in the block's inclusive scan
// (LDS) and derives the tile from the slot's rank inside the rect -- fully coalesced stores, no divergence on
// the splat size (a per-Gaussian loop here ran at 0.5 TB/s).
// THREADS: 256, or 1024 for small frames (a 100 k-Gaussian frame has ~100 blocks of ~2000 instances: with 256 threads each
// block walks its run in eight trips while most of the chip idles); the first 256 threads load and scan, all emit.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void emit_instances_kernel(const uint32_t* __restrict__ total, int grid_x,
                                                                   const uint2* __restrict__ d3,
                                                                   const uint2* __restrict__ d4,
                                                                   const BinInfo* __restrict__ bin,
                                                                   const uint32_t* __restrict__ block_offs2,
                                                                   uint32_t* __restrict__ inst_tile,
                                                                   uint32_t* __restrict__ inst_g, uint32_t capacity) {
  // THREADS == 1024 (small frames, at most EMIT_WIDE_MAX_BLOCKS blocks): block_offs2 holds the block TOTALS of
  // count_tiles, not their scan -- every block sums the totals in front of it itself (one load per thread) and the scan
  // kernel between the two is not launched
  const uint32_t V = total[TOTAL_V];
  if (blockIdx.x * PRE_BLOCK >= V) return;      // grid sized for a capacity (uniform exit: no barrier crossed)
  const uint2* __restrict__ dval = total[TOTAL_TOP_PASS_N] ? d4 : d3;
  __shared__ uint32_t wave_tot[PRE_BLOCK / WAVE];
  __shared__ uint32_t s_incl[PRE_BLOCK];      // inclusive scan of tiles within the block
  __shared__ uint32_t s_g[PRE_BLOCK];
  __shared__ uint32_t s_min[PRE_BLOCK];       // rect min: x | y << 16
  __shared__ uint32_t s_wh[PRE_BLOCK];        // rect width | height << 16
  __shared__ uint32_t s_mask[PRE_BLOCK];      // tile mask (BinInfo::mask)
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
  const bool loader = THREADS == PRE_BLOCK || tid < PRE_BLOCK;
  const uint32_t i = blockIdx.x * PRE_BLOCK + tid;
  uint32_t g = 0, tiles = 0, mask = 0;
  uint2 rr = make_uint2(0u, 0u);
  if (loader && i < V) {
    const uint2 v = dval[i];
    g = v.x;
    if (v.y & PACK_FALLBACK) {
      const BinInfo bi = bin[g];
      mask = bi.mask;
      rr = make_uint2(bi.rect_min, bi.rect_wh);
    } else {
      unpack_rect(v.y, (uint32_t)grid_x, rr.x, rr.y, mask);
    }
    tiles = bin_count(rr.y, mask);
  }
  const uint32_t inc = wave_incl_scan_u32(tiles);
  if (loader && lane == WAVE - 1) wave_tot[wid] = inc;
  __syncthreads();
  if (loader) {
    uint32_t wbase = 0;
#pragma unroll
    for (int w = 0; w < PRE_BLOCK / WAVE; ++w)
      if (w < wid) wbase += wave_tot[w];
    s_incl[tid] = wbase + inc;
    s_g[tid] = g;
    s_min[tid] = rr.x;
    s_wh[tid] = rr.y;
    s_mask[tid] = mask;
  }
  __syncthreads();
  const uint32_t n_out = s_incl[PRE_BLOCK - 1];
  uint32_t base;
  if (THREADS == PRE_BLOCK) {
    base = block_offs2[blockIdx.x];
  } else {
    __shared__ uint32_t s_part[THREADS / WAVE];
    const uint32_t mine = (uint32_t)tid < blockIdx.x ? block_offs2[tid] : 0u;       // blockIdx.x < EMIT_WIDE_MAX_BLOCKS <= THREADS
    const uint32_t ws = wave_reduce_add_u32(mine);
    if (lane == 0) s_part[wid] = ws;
    __syncthreads();
    base = 0;
#pragma unroll
    for (int w = 0; w < THREADS / WAVE; ++w) base += s_part[w];
  }
  for (uint32_t o = (uint32_t)tid; o < n_out; o += THREADS) {
    // smallest j with s_incl[j] > o
    int lo = 0, hi = PRE_BLOCK - 1;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int mid = (lo + hi) >> 1;
      if (s_incl[mid] > o) hi = mid; else lo = mid + 1;
    }
    const int j = lo;
    const uint32_t first = j ? s_incl[j - 1] : 0u;
    const uint32_t wh = s_wh[j];
    const uint32_t k = bin_kth(wh, s_mask[j], o - first);   // tile inside the rect (y outer, x inner) of instance o - first
    const uint32_t w = max(wh & 0xffffu, 1u);
    uint32_t q = (uint32_t)(((float)k + 0.5f) * (1.0f / (float)w));   // k / w (exact for k < 2^20)
    const uint32_t mn = s_min[j];
    const uint32_t x = (mn & 0xffffu) + (k - q * w), y = (mn >> 16) + q;
    if (base + o < capacity) {      // an overflowing frame drops its tail (the caller sees num_rendered > capacity)
      inst_tile[base + o] = y * (uint32_t)grid_x + x;
      inst_g[base + o] = s_g[j];
    }
  }
}

// debug: the 64-bit keys a (tile, depth) sort would have produced, rebuilt from the two-level result
__global__ __launch_bounds__(256) void reconstruct_keys_kernel(uint32_t R, uint32_t P,
                                                               const uint32_t* __restrict__ tile_sorted,
                                                               const uint32_t* __restrict__ point_list,
                                                               const BinInfo* __restrict__ bin,
                                                               uint64_t* __restrict__ keys) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= R) return;
  // R may be the capacity of a gsr_forward frame: entries past the real count are uninitialised, never dereferenced
  const uint32_t g = point_list[i];
  keys[i] = g < P ? ((uint64_t)tile_sorted[i] << 32) | (uint64_t)__float_as_uint(bin[g].depth) : ~0ull;
}

// ------------------------------------------------------------------------------------------
// Radix sort, one digit of 6..9 bits per pass (sort_pass_plan), three kernels per pass (no inter-workgroup waiting):
//   hist    : per-block digit histogram      -> hist[digit][block]
//   rowscan : exclusive scan of every digit row over blocks, row totals -> totals[digit]
//   scatter : stable local ranking (wave match + LDS), exchange through LDS so that each
//             digit's run is written contiguously, global offset = digit base + row prefix.
// Item order inside a block tile: wave w owns [w*1024, w*1024+1024); item i of lane l is
// element i*64 + l of that range, so (i, l) lexicographic == memory order (stability).
// ------------------------------------------------------------------------------------------
// ---- segmented pass ------------------------------------------------------------------------------------------------
// The LAST pass of a two-pass sort can have its blocks aligned with the SEGMENTS the first pass left behind (segment L =
// the run of items whose first digit is L, lengths = the first pass's digit totals): block b then holds items of ONE
// first digit only, so the exclusive row prefixes of its histogram matrix ARE the boundaries of every full key's run --
// the tile ranges of the binning fall out of the sort (ranges_from_hist_kernel) and no pass over the sorted keys
// (identifyTileRanges) nor a memset is needed.  Costs ~one partial block per segment.
struct SegBlock { uint32_t base, count, nblocks; bool valid; };
// blocks of a segment of n items
__device__ inline uint32_t seg_blocks_of(uint32_t n) { return (n + (uint32_t)SORT_TILE - 1u) / (uint32_t)SORT_TILE; }
// (first item, item count) of block b and the number of blocks in use; every lane of the calling wave gets the result.
// seg_count <= 512 segments, eight per lane.
__device__ inline SegBlock seg_block(const uint32_t* __restrict__ seg_totals, int seg_count, uint32_t b) {
  const int lane = lane_id();
  uint32_t cnt[8], items = 0, blocks = 0;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int sgm = lane * 8 + u;
    cnt[u] = sgm < seg_count ? seg_totals[sgm] : 0u;
    items += cnt[u];
    blocks += seg_blocks_of(cnt[u]);
  }
  const uint32_t item_inc = wave_incl_scan_u32(items), blk_inc = wave_incl_scan_u32(blocks);
  SegBlock r;
  r.nblocks = (uint32_t)__shfl((int)blk_inc, WAVE - 1, WAVE);
  r.valid = b < r.nblocks;
  r.base = 0; r.count = 0;
  // the lane whose eight segments hold block b
  const unsigned long long owner = __ballot(b >= blk_inc - blocks && b < blk_inc);
  if (owner) {
    const int src = __ffsll((long long)owner) - 1;
    uint32_t ib = item_inc - items, bb = blk_inc - blocks, base = 0, count = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const uint32_t nb = seg_blocks_of(cnt[u]);
      if (b >= bb && b < bb + nb) {
        const uint32_t off = (b - bb) * (uint32_t)SORT_TILE;
        base = ib + off;
        count = min((uint32_t)SORT_TILE, cnt[u] - off);
      }
      ib += cnt[u]; bb += nb;
    }
    r.base = (uint32_t)__shfl((int)base, src, WAVE);
    r.count = (uint32_t)__shfl((int)count, src, WAVE);
  }
  return r;
}

template <typename KeyT>
__device__ inline uint32_t digit_of(KeyT k, int shift, uint32_t mask) { return (uint32_t)(k >> shift) & mask; }

template <typename KeyT, int BITS, int ITEMS>
__global__ __launch_bounds__(SORT_THREADS) void radix_hist_kernel(const KeyT* __restrict__ keys, uint32_t n,
                                                                  const uint32_t* __restrict__ n_dev, int shift,
                                                                  uint32_t mask, uint32_t nblocks,
                                                                  uint32_t* __restrict__ hist,
                                                                  const uint32_t* __restrict__ seg_totals = nullptr,
                                                                  int seg_count = 0) {
  constexpr int RADIX = 1 << BITS;
  __shared__ uint32_t h[RADIX];
  if (n_dev) n = *n_dev;     // element count known only on the device (grid sized for the capacity)
  uint32_t base = blockIdx.x * (SORT_THREADS * ITEMS);
  if (seg_totals) {
    const SegBlock sb = seg_block(seg_totals, seg_count, blockIdx.x);
    if (!sb.valid) return;
    base = sb.base;
    n = sb.base + sb.count;
  } else if (base >= n) return;                           // block past the count (grid sized for a capacity; n == 0: a
                                                           // predicated pass that does not run) -- rowscan and scatter skip
                                                           // the same columns
  const int tid = threadIdx.x;
#pragma unroll
  for (int d = tid; d < RADIX; d += SORT_THREADS) h[d] = 0;
  __syncthreads();
  KeyT kk[ITEMS];
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {       // all loads in flight before the first LDS atomic
    const uint32_t g = base + i * SORT_THREADS + tid;
    kk[i] = g < n ? keys[g] : (KeyT)0;
  }
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const uint32_t g = base + i * SORT_THREADS + tid;
    if (g < n) atomicAdd(&h[digit_of(kk[i], shift, mask)], 1u);
  }
  __syncthreads();
#pragma unroll
  for (int d = tid; d < RADIX; d += SORT_THREADS) hist[(size_t)d * nblocks + blockIdx.x] = h[d];
}

// one block per digit row; rounds of 2048 entries: every lane scans 8 contiguous entries in registers (two 16-byte
// loads), the block scans the 256 lane sums, the lane writes its 8 prefixes back
__global__ __launch_bounds__(256) void radix_rowscan_kernel(uint32_t* __restrict__ hist, uint32_t nblocks,
                                                            uint32_t* __restrict__ totals,
                                                            const uint32_t* __restrict__ n_dev,
                                                            const uint32_t* __restrict__ seg_totals = nullptr,
                                                            int seg_count = 0, uint2* __restrict__ runs_rel = nullptr,
                                                            int lo_bits = 0, uint32_t n_keys = 0) {
  constexpr uint32_t PER = 8;
  __shared__ uint32_t wave_tot[256 / WAVE];
  __shared__ uint32_t seg_first[RADIX + 1];    // segmented pass: first block of every segment; [seg_count] = blocks in use
  const uint32_t stride = nblocks;
  if (seg_totals) {           // segmented pass: the columns in use are the blocks its segments need
    if (threadIdx.x < WAVE) {
      uint32_t b8[8], bs = 0;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int sgm = (int)threadIdx.x * 8 + u;
        b8[u] = sgm < seg_count ? seg_blocks_of(seg_totals[sgm]) : 0u;
        bs += b8[u];
      }
      uint32_t br = wave_incl_scan_u32(bs) - bs;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int sgm = (int)threadIdx.x * 8 + u;
        if (sgm < seg_count) seg_first[sgm] = br;
        br += b8[u];
      }
      if (threadIdx.x == WAVE - 1) seg_first[seg_count] = br;     // the blocks in use (seg_count may be 512 = 64 x 8)
    }
    __syncthreads();
    nblocks = min(nblocks, seg_first[seg_count]);
    if (nblocks == 0) {
      // no item at all: every run of this row is empty
      if (runs_rel)
        for (int sgm = threadIdx.x; sgm < seg_count; sgm += 256) {
          const uint32_t key = ((uint32_t)blockIdx.x << lo_bits) | (uint32_t)sgm;
          if (key < n_keys) runs_rel[key] = make_uint2(0u, 0u);
        }
      if (threadIdx.x == 0) totals[blockIdx.x] = 0;
      return;
    }
  } else if (n_dev) {
    const uint32_t n = *n_dev;
    if (n == 0) {
      // no item on the device (a frame without a visible Gaussian, or a predicated pass that does not run): the row total
      // is still published -- the segmented pass and ranges_and_order_from_sort read totals[] whatever the count was, and
      // would otherwise rebuild the PREVIOUS frame's runs from what that frame left here
      if (threadIdx.x == 0) totals[blockIdx.x] = 0;
      return;
    }
    nblocks = min(nblocks, (n + SORT_TILE - 1) / SORT_TILE);   // columns past the device-side count are never written or read
  }
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
  uint32_t* row = hist + (size_t)blockIdx.x * stride;
  uint32_t carry = 0;
  for (uint32_t b0 = 0; b0 < nblocks; b0 += 256 * PER) {
    const uint32_t lo = b0 + (uint32_t)tid * PER;
    uint32_t v[PER];
    const bool full = lo + PER <= nblocks;
    if (full) {
      const uint4 q0 = *reinterpret_cast<const uint4*>(row + lo), q1 = *reinterpret_cast<const uint4*>(row + lo + 4);
      v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w; v[4] = q1.x; v[5] = q1.y; v[6] = q1.z; v[7] = q1.w;
    } else {
#pragma unroll
      for (uint32_t k = 0; k < PER; ++k) v[k] = lo + k < nblocks ? row[lo + k] : 0u;
    }
    uint32_t mine = 0;
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) mine += v[k];
    const uint32_t inc = wave_incl_scan_u32(mine);
    if (b0) __syncthreads();
    if (lane == WAVE - 1) wave_tot[wid] = inc;
    __syncthreads();
    uint32_t woff = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 256 / WAVE; ++w) {
      const uint32_t t = wave_tot[w];
      if (w < wid) woff += t;
      all += t;
    }
    uint32_t run = carry + woff + inc - mine;
    if (full) {
      uint4 q0, q1;
      q0.x = run; run += v[0]; q0.y = run; run += v[1]; q0.z = run; run += v[2]; q0.w = run; run += v[3];
      q1.x = run; run += v[4]; q1.y = run; run += v[5]; q1.z = run; run += v[6]; q1.w = run; run += v[7];
      *reinterpret_cast<uint4*>(row + lo) = q0;
      *reinterpret_cast<uint4*>(row + lo + 4) = q1;
    } else {
#pragma unroll
      for (uint32_t k = 0; k < PER; ++k) {
        if (lo + k < nblocks) row[lo + k] = run;
        run += v[k];
      }
    }
    carry += all;
  }
  if (tid == 0) totals[blockIdx.x] = carry;
  if (runs_rel) {
    // Segmented pass: the items of key (this row's digit << lo_bits | L) are those this row counts in the blocks of
    // segment L, so the key's run starts, relative to the row's first item, at the row prefix of the segment's first
    // block and ends at the prefix of the next segment's first block -- both just written.  (__syncthreads makes the
    // block's own global stores visible to it.)
    __syncthreads();
    for (int sgm = tid; sgm < seg_count; sgm += 256) {
      const uint32_t key = ((uint32_t)blockIdx.x << lo_bits) | (uint32_t)sgm;
      if (key < n_keys) {
        const uint32_t b0 = seg_first[sgm], b1 = seg_first[sgm + 1];
        const uint32_t x = b0 < nblocks ? row[b0] : carry, y = b1 < nblocks ? row[b1] : carry;
        runs_rel[key] = make_uint2(x, y);
      }
    }
  }
}

template <typename KeyT, typename ValT, int BITS, int ITEMS>
__global__ __launch_bounds__(SORT_THREADS) void radix_scatter_kernel(
    const KeyT* __restrict__ keys_in, const ValT* __restrict__ vals_in, KeyT* __restrict__ keys_out,
    ValT* __restrict__ vals_out, uint32_t n, const uint32_t* __restrict__ n_dev, int shift, uint32_t mask,
    uint32_t nblocks, const uint32_t* __restrict__ hist, const uint32_t* __restrict__ totals,
    const uint32_t* __restrict__ seg_totals = nullptr, int seg_count = 0) {
  constexpr int RADIX = 1 << BITS;
  constexpr int NW = SORT_THREADS / WAVE;
  if (n_dev) n = *n_dev;
  uint32_t base = blockIdx.x * (SORT_THREADS * ITEMS);
  if (seg_totals) {                                        // segmented pass: see seg_block
    const SegBlock sb = seg_block(seg_totals, seg_count, blockIdx.x);
    if (!sb.valid) return;
    base = sb.base;
    n = sb.base + sb.count;                                // the block's items end where its segment (or its 4096) ends
  } else if (base >= n) return;   // block beyond the device-side count (uniform: no barrier crossed)
  using WideT = typename std::conditional<(sizeof(ValT) > sizeof(KeyT)), ValT, KeyT>::type;
  __shared__ WideT xbuf_w[(SORT_THREADS * ITEMS)];         // exchange buffer: keys first, then reused for the values
  KeyT* xbuf = reinterpret_cast<KeyT*>(xbuf_w);
  __shared__ uint32_t wave_hist[NW][RADIX];   // per-wave digit counts, then exclusive wave prefixes
  __shared__ uint32_t digit_start[RADIX];     // first local slot of every digit
  __shared__ uint32_t global_base[RADIX];     // global position of the block's first item of the digit
  __shared__ uint32_t scan_tmp[NW];

  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
  const uint32_t wbase = base + wid * (WAVE * ITEMS);

#pragma unroll
  for (int w = 0; w < NW; ++w)
#pragma unroll
    for (int d = tid; d < RADIX; d += SORT_THREADS) wave_hist[w][d] = 0;
  __syncthreads();

  KeyT k[ITEMS];
  ValT v[ITEMS];
  uint32_t rank[ITEMS];
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const uint32_t g = wbase + i * WAVE + lane;
    const uint32_t gc = min(g, n - 1u);       // unconditional, index-clamped loads: all in flight together
    k[i] = keys_in[gc];
    v[i] = vals_in[gc];
  }
  // Stable rank of every item among the wave's items with the same digit: match masks by ballot, then the
  // wave's digit counter is read by every matching lane and bumped by the lowest one (LDS operations of a
  // wave execute in order, so all peers read before the leader writes).
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const uint32_t g = wbase + i * WAVE + lane;
    const bool ok = g < n;
    const uint32_t d = digit_of(k[i], shift, mask);
    unsigned long long peers = __ballot(ok);   // padding lanes never match real ones
    if (!ok) peers = ~peers;
#pragma unroll
    for (int b = 0; b < BITS; ++b) {
      const bool bit = (d >> b) & 1u;
      const unsigned long long bal = __ballot(bit);
      peers &= bit ? bal : ~bal;
    }
    const uint32_t cnt = (uint32_t)__popcll(peers);
    const uint32_t before = (uint32_t)__popcll(peers & lt_mask);
    uint32_t old = 0;
    if (ok) {
      old = wave_hist[wid][d];
      if (before == 0) wave_hist[wid][d] = old + cnt;
    }
    rank[i] = old + before;
  }
  __syncthreads();

  // digit totals over the waves -> exclusive wave prefixes + block-wide exclusive scan.
  // Thread t owns the BPT adjacent digits t*BPT .. t*BPT+BPT-1 (threads past the last digit own none).
  {
    constexpr int BPT = RADIX >= SORT_THREADS ? RADIX / SORT_THREADS : 1;
    const bool owner = tid * BPT < RADIX;
    uint32_t dsum[BPT], tot[BPT];
    uint32_t mine = 0, tmine = 0;
#pragma unroll
    for (int e = 0; e < BPT; ++e) {
      const int d = tid * BPT + e;
      uint32_t run = 0;
      dsum[e] = 0; tot[e] = 0;
      if (owner) {
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          const uint32_t c = wave_hist[w][d];
          wave_hist[w][d] = run;
          run += c;
        }
        dsum[e] = run;
        mine += run;
        tot[e] = totals[d];
        tmine += tot[e];
      }
    }
    const uint32_t inc = wave_incl_scan_u32(mine);
    if (lane == WAVE - 1) scan_tmp[wid] = inc;
    __syncthreads();
    uint32_t woff = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w)
      if (w < wid) woff += scan_tmp[w];
    uint32_t excl = woff + inc - mine;
    // global base: all smaller digits everywhere + this digit in earlier blocks
    const uint32_t t_inc = wave_incl_scan_u32(tmine);
    __syncthreads();
    if (lane == WAVE - 1) scan_tmp[wid] = t_inc;
    __syncthreads();
    uint32_t toff = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w)
      if (w < wid) toff += scan_tmp[w];
    uint32_t texcl = toff + t_inc - tmine;
    if (owner) {
#pragma unroll
      for (int e = 0; e < BPT; ++e) {
        const int d = tid * BPT + e;
        digit_start[d] = excl;
        global_base[d] = texcl + hist[(size_t)d * nblocks + blockIdx.x];
        excl += dsum[e];
        texcl += tot[e];
      }
    }
  }
  __syncthreads();

  // exchange through LDS so that every digit's run leaves the block contiguously; keys and values take
  // turns in the same buffer.  local slot = digit_start + (items of the digit in earlier waves) + rank in wave
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {   // the rank becomes the local slot in place
    const uint32_t g = wbase + i * WAVE + lane;
    const uint32_t d = digit_of(k[i], shift, mask);
    rank[i] = digit_start[d] + wave_hist[wid][d] + rank[i];
    if (g < n) xbuf[rank[i]] = k[i];
  }
  __syncthreads();
  const uint32_t count = min((uint32_t)(SORT_THREADS * ITEMS), n - base);
  uint32_t dst[ITEMS];
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const uint32_t s = i * SORT_THREADS + tid;
    dst[i] = 0;
    if (s < count) {
      const KeyT kk = xbuf[s];
      const uint32_t d = digit_of(kk, shift, mask);
      dst[i] = global_base[d] + (s - digit_start[d]);
      keys_out[dst[i]] = kk;
    }
  }
  __syncthreads();
  ValT* xv = reinterpret_cast<ValT*>(xbuf_w);
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const uint32_t g = wbase + i * WAVE + lane;
    if (g < n) xv[rank[i]] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const uint32_t s = i * SORT_THREADS + tid;
    if (s < count) vals_out[dst[i]] = xv[s];
  }
}

// ------------------------------------------------------------------------------------------
// identifyTileRanges (A.3): ranges[tile] = [first, last+1) of its run in the sorted keys.
// The caller zero-fills `ranges` (empty tiles stay (0,0)).
// ------------------------------------------------------------------------------------------
template <typename KeyT, int TILE_SHIFT>
__global__ __launch_bounds__(256) void identify_tile_ranges_kernel(uint32_t R, const uint32_t* __restrict__ r_dev,
                                                                   const KeyT* __restrict__ keys,
                                                                   uint2* __restrict__ ranges) {
  if (r_dev) R = *r_dev;      // instance count known only on the device (grid sized for the capacity)
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= R) return;
  const uint32_t cur = (uint32_t)(keys[i] >> TILE_SHIFT);
  if (i == 0) {
    ranges[cur].x = 0;
  } else {
    const uint32_t prev = (uint32_t)(keys[i - 1] >> TILE_SHIFT);
    if (cur != prev) {
      ranges[prev].y = i;
      ranges[cur].x = i;
    }
  }
  if (i == R - 1) ranges[cur].y = R;
}


// ------------------------------------------------------------------------------------------
// Tile processing order for the compositing kernels: longest instance list first (bucketed by a
// monotone 8-bit code of the length), so that the waves that start late pick up the short tiles.
// One block.  The order inside a bucket is arbitrary; no result depends on it.
// ------------------------------------------------------------------------------------------
__device__ inline uint32_t len_bucket(uint32_t len) {
  if (len < 16u) return len;
  const uint32_t e = 31u - (uint32_t)__clz((int)len);          // >= 4
  const uint32_t b = 16u + (e - 4u) * 8u + ((len >> (e - 3u)) & 7u);
  return b > 255u ? 255u : b;
}
// lanes of the wave that hold the same 8-bit code (the synthetic clouds put almost every tile into two or three
// buckets: one LDS atomic per lane would serialise)
__device__ inline unsigned long long match8(uint32_t code, bool ok) {
  unsigned long long peers = __ballot(ok);
  if (!ok) peers = ~peers;
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const bool bit = (code >> b) & 1u;
    const unsigned long long bal = __ballot(bit);
    peers &= bit ? bal : ~bal;
  }
  return peers;
}
__global__ __launch_bounds__(1024) void build_tile_order_kernel(int tiles, const uint2* __restrict__ ranges,
                                                                uint32_t* __restrict__ order) {
  constexpr int CHUNK = 8 * 1024;       // tiles ordered per pass (a 1080p frame has 8160)
  __shared__ uint32_t cnt[256];
  __shared__ uint32_t base[256];
  __shared__ uint8_t s_code[CHUNK];     // the length code of every tile of the chunk: the ranges are read ONCE
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (int c0 = 0; c0 < tiles; c0 += CHUNK) {
    if (c0 > 0) __syncthreads();        // the previous chunk's second pass is done with cnt / base / s_code
    if (tid < 256) cnt[tid] = 0;
    // one trip to memory for the whole chunk: eight independent loads per thread in flight (round 2 read every range
    // twice, in sixteen dependent trips: 14 us for the 8160 tiles of a 1080p frame, on the critical path of every forward)
    {
      uint2 r[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = c0 + u * 1024 + tid;
        r[u] = t < tiles ? ranges[t] : make_uint2(0u, 0u);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) s_code[u * 1024 + tid] = (uint8_t)(255u - len_bucket(r[u].y - r[u].x));
    }
    __syncthreads();
    const int n = min(CHUNK, tiles - c0);
    for (int t0 = 0; t0 < n; t0 += 1024) {
      const int t = t0 + tid;
      const bool ok = t < n;
      const uint32_t code = s_code[t0 + tid];
      const unsigned long long peers = match8(code, ok);
      if (ok && (peers & lt) == 0ull) atomicAdd(&cnt[code], (uint32_t)__popcll(peers));     // one atomic per group
    }
    __syncthreads();
    if (tid < WAVE) {     // exclusive scan of the 256 counts by one wave (4 per lane)
      uint32_t c[4], mine = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) { c[k] = cnt[4 * tid + k]; mine += c[k]; }
      uint32_t run = wave_incl_scan_u32(mine) - mine;
#pragma unroll
      for (int k = 0; k < 4; ++k) { base[4 * tid + k] = run; run += c[k]; }
    }
    __syncthreads();
    // a frame of more than 8192 tiles is ordered chunk by chunk (longest first inside each chunk): the order only
    // balances the tail of the compositing kernels, no result depends on it
    for (int t0 = 0; t0 < n; t0 += 1024) {
      const int t = t0 + tid;
      const bool ok = t < n;
      const uint32_t code = s_code[t0 + tid];
      const unsigned long long peers = match8(code, ok);
      uint32_t first = 0;
      const int leader = __ffsll((long long)peers) - 1;
      if (ok && lane == leader) first = atomicAdd(&base[code], (uint32_t)__popcll(peers));
      first = (uint32_t)__shfl((int)first, leader, WAVE);
      if (ok) order[c0 + first + (uint32_t)__popcll(peers & lt)] = (uint32_t)(c0 + t);
    }
  }
}

// Two-level binning: tile ranges AND tile order from what the tile sort left behind (one block, no pass over the keys,
// no memset).
//   one-pass sort (at most 512 tiles): ranges[t] = (digit_base[t], digit_base[t + 1]), digit_base = exclusive scan of
//     the digit totals;
//   two-pass sort with a segmented last pass (seg_block): the row scan of that pass left in ranges[t] the run of tile
//     t = hi << lo_bits | lo RELATIVE to the first item with top digit hi; adding digit_base[hi] completes it.
// An empty tile gets (0, 0), as in the zero-filled array upstream's identifyTileRanges writes into.
__global__ __launch_bounds__(1024) void ranges_and_order_from_sort_kernel(int tiles, const uint32_t* __restrict__ totals_last,
                                                                          int lo_bits, int hi_bits, bool relative,
                                                                          uint2* __restrict__ ranges,
                                                                          uint32_t* __restrict__ order) {
  constexpr int CHUNK = 8 * 1024;
  __shared__ uint32_t cnt[256];
  __shared__ uint32_t base[256];
  __shared__ uint8_t s_code[CHUNK];
  __shared__ uint32_t digit_base[RADIX + 1];   // exclusive scan of the last pass's digit totals
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  const unsigned long long lt = (1ull << lane) - 1ull;
  const int nhi = 1 << hi_bits;
  // the first chunk's relative runs are requested before the scan of the totals: one trip to memory for both
  uint2 r[8];
  {
    const int t_lo = tid * 8;
    if (relative && t_lo + 8 <= tiles) {
      const uint4* src = reinterpret_cast<const uint4*>(ranges + t_lo);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint4 q = src[u];
        r[2 * u] = make_uint2(q.x, q.y); r[2 * u + 1] = make_uint2(q.z, q.w);
      }
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u) r[u] = (relative && t_lo + u < tiles) ? ranges[t_lo + u] : make_uint2(0u, 0u);
    }
  }
  if (tid < WAVE) {
    uint32_t t8[8], ts = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int d = tid * 8 + u;
      t8[u] = d < nhi ? totals_last[d] : 0u;
      ts += t8[u];
    }
    uint32_t tr = wave_incl_scan_u32(ts) - ts;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int d = tid * 8 + u;
      if (d < nhi) digit_base[d] = tr;
      tr += t8[u];
    }
    if (tid == WAVE - 1) digit_base[nhi] = tr;      // nhi may be 512 = 64 x 8
  }
  __syncthreads();
  for (int c0 = 0; c0 < tiles; c0 += CHUNK) {
    const int t_lo = c0 + tid * 8;        // eight consecutive tiles per thread
    if (c0 > 0) {
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 8; ++u) r[u] = (relative && t_lo + u < tiles) ? ranges[t_lo + u] : make_uint2(0u, 0u);
    }
    if (tid < 256) cnt[tid] = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int t = t_lo + u;
      uint32_t x = 0, y = 0;
      if (t < tiles) {
        if (!relative) { x = digit_base[t]; y = digit_base[t + 1]; }
        else { const uint32_t db = digit_base[(uint32_t)t >> lo_bits]; x = db + r[u].x; y = db + r[u].y; }
      }
      r[u] = y > x ? make_uint2(x, y) : make_uint2(0u, 0u);
      s_code[tid * 8 + u] = (uint8_t)(255u - len_bucket(r[u].y - r[u].x));
    }
    if (t_lo + 8 <= tiles) {
      uint4* dst = reinterpret_cast<uint4*>(ranges + t_lo);
#pragma unroll
      for (int u = 0; u < 4; ++u) dst[u] = make_uint4(r[2 * u].x, r[2 * u].y, r[2 * u + 1].x, r[2 * u + 1].y);
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (t_lo + u < tiles) ranges[t_lo + u] = r[u];
    }
    __syncthreads();
    const int n = min(CHUNK, tiles - c0);
    for (int t0 = 0; t0 < n; t0 += 1024) {
      const int t = t0 + tid;
      const bool ok = t < n;
      const uint32_t code = s_code[t0 + tid];
      const unsigned long long peers = match8(code, ok);
      if (ok && (peers & lt) == 0ull) atomicAdd(&cnt[code], (uint32_t)__popcll(peers));     // one atomic per group
    }
    __syncthreads();
    if (tid < WAVE) {     // exclusive scan of the 256 bucket sizes by one wave (4 per lane)
      uint32_t b[4], m4 = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) { b[k] = cnt[4 * tid + k]; m4 += b[k]; }
      uint32_t rr = wave_incl_scan_u32(m4) - m4;
#pragma unroll
      for (int k = 0; k < 4; ++k) { base[4 * tid + k] = rr; rr += b[k]; }
    }
    __syncthreads();
    for (int t0 = 0; t0 < n; t0 += 1024) {
      const int t = t0 + tid;
      const bool ok = t < n;
      const uint32_t code = s_code[t0 + tid];
      const unsigned long long peers = match8(code, ok);
      uint32_t first = 0;
      const int leader = __ffsll((long long)peers) - 1;
      if (ok && lane == leader) first = atomicAdd(&base[code], (uint32_t)__popcll(peers));
      first = (uint32_t)__shfl((int)first, leader, WAVE);
      if (ok) order[c0 + first + (uint32_t)__popcll(peers & lt)] = (uint32_t)(c0 + t);
    }
  }
}

void launch_duplicate_with_keys(int P, int grid_x, const BinInfo* bin, const uint32_t* block_offs, uint32_t* slot_base,
                                uint32_t* point_offsets, uint64_t* keys, uint32_t* vals, hipStream_t s) {
  const int nb = (P + PRE_BLOCK - 1) / PRE_BLOCK;
  if (nb > 0)
    hipLaunchKernelGGL(duplicate_with_keys_kernel, dim3(nb), dim3(PRE_BLOCK), 0, s, P, grid_x, bin, block_offs, slot_base,
                       point_offsets, keys, vals);
}

template <typename KeyT, typename ValT, int BITS>
static void sort_pass(const KeyT* kin, const ValT* vin, KeyT* kout, ValT* vout, uint32_t n, const uint32_t* n_dev,
                      int shift, int nbits, const SortLayout& L, uint32_t* hist, uint32_t* totals, hipStream_t s,
                      const uint32_t* seg_totals = nullptr, int seg_count = 0, uint2* runs_rel = nullptr,
                      int lo_bits = 0, uint32_t n_keys = 0) {
  const uint32_t mask = (1u << nbits) - 1u;      // nbits <= BITS: digits above the mask do not occur
  // (8192-item tiles for 32-bit keys were measured: no gain once the digits are <= 8 bits wide)
  constexpr int ITEMS = SORT_ITEMS;
  // a segmented pass needs up to one partial block per segment more (SortLayout reserves the columns)
  const uint32_t nblocks = (n + SORT_THREADS * ITEMS - 1) / (SORT_THREADS * ITEMS) + (seg_totals ? (uint32_t)seg_count : 0u);
  (void)L;
  hipLaunchKernelGGL((radix_hist_kernel<KeyT, BITS, ITEMS>), dim3(nblocks), dim3(SORT_THREADS), 0, s, kin, n, n_dev, shift,
                     mask, nblocks, hist, seg_totals, seg_count);
  hipLaunchKernelGGL(radix_rowscan_kernel, dim3(1 << BITS), dim3(256), 0, s, hist, nblocks, totals, n_dev, seg_totals,
                     seg_count, runs_rel, lo_bits, n_keys);
  hipLaunchKernelGGL((radix_scatter_kernel<KeyT, ValT, BITS, ITEMS>), dim3(nblocks), dim3(SORT_THREADS), 0, s, kin, vin, kout,
                     vout, n, n_dev, shift, mask, nblocks, hist, totals, seg_totals, seg_count);
}

// Digit widths of the passes.  A block scatters 4096 items: with 2^w digits a digit's run leaves the block as
// 4096/2^w contiguous items, so narrow digits write long segments (w = 9: 32-byte fragments; w = 6..8: 64..256 bytes)
// and need fewer ballots to rank.  The pass count is ceil(end_bit / 9) (the fewest the 9-bit kernels allow); the bits
// are spread evenly over those passes, the wider digits LAST: 32 depth bits -> 8+8+8+8, 13 tile bits -> 6+7, 45-bit keys ->
// 9 x 5.  (The last pass of the tile sort sees the top bits of the tile id, which a scene with dense blobs concentrates on
// a few values: its LDS histogram atomics collide less over 128 bins than over 64 -- heavy-tailed C4 tile sort -19 us,
// uniform clouds unchanged.)
int sort_pass_plan(int end_bit, int widths[8]) {
  const int passes = sort_passes(end_bit);
  const int lo = end_bit / passes, extra = end_bit % passes;
  for (int p = 0; p < passes; ++p) widths[p] = lo + (p >= passes - extra ? 1 : 0);
  return passes;
}

// n = element count, or the capacity when n_dev (device-side count) is given.
// runs != NULL (one- and two-pass sorts only): the last pass is segmented and *runs describes where every key's run lies.
template <typename KeyT, typename ValT>
static bool sort_pairs_impl(KeyT* keys_a, ValT* vals_a, KeyT* keys_b, ValT* vals_b, uint32_t n, int end_bit,
                            void* scratch, hipStream_t s, const uint32_t* n_dev = nullptr, SortedRuns* runs = nullptr) {
  if (runs) runs->valid = false;
  if (n == 0 || end_bit <= 0) return false;
  // the rows of a segmented pass write one relative run per key value: every row must exist in the launch, i.e. the last
  // digit's kernels are instantiated for exactly its width -- true for the 6..9-bit kernels; narrower last digits run the
  // 6-bit kernels, whose 64 rows cover them
  const SortLayout L(n);
  uint32_t* hist = reinterpret_cast<uint32_t*>(static_cast<char*>(scratch) + L.hist);
  uint32_t* totals_pp[2] = {reinterpret_cast<uint32_t*>(static_cast<char*>(scratch) + L.totals),
                            reinterpret_cast<uint32_t*>(static_cast<char*>(scratch) + L.totals_odd)};
  int widths[8];
  const int passes = sort_pass_plan(end_bit, widths);
  KeyT* kin = keys_a; ValT* vin = vals_a; KeyT* kout = keys_b; ValT* vout = vals_b;
  int shift = 0;
  for (int pass = 0; pass < passes; ++pass) {
    const int w = widths[pass];
    uint32_t* totals = totals_pp[pass & 1];
    const bool seg = runs && passes == 2 && pass == 1;
    const uint32_t* st = seg ? totals_pp[0] : nullptr;
    const int sc = seg ? 1 << widths[0] : 0;
    uint2* rr = seg ? runs->runs_rel : nullptr;
    const int lb = widths[0];
    const uint32_t nk = seg ? runs->n_keys : 0u;
    if (w <= 6)      sort_pass<KeyT, ValT, 6>(kin, vin, kout, vout, n, n_dev, shift, w, L, hist, totals, s, st, sc, rr, lb, nk);
    else if (w == 7) sort_pass<KeyT, ValT, 7>(kin, vin, kout, vout, n, n_dev, shift, w, L, hist, totals, s, st, sc, rr, lb, nk);
    else if (w == 8) sort_pass<KeyT, ValT, 8>(kin, vin, kout, vout, n, n_dev, shift, w, L, hist, totals, s, st, sc, rr, lb, nk);
    else             sort_pass<KeyT, ValT, 9>(kin, vin, kout, vout, n, n_dev, shift, w, L, hist, totals, s, st, sc, rr, lb, nk);
    shift += w;
    KeyT* tk = kin; kin = kout; kout = tk;
    ValT* tv = vin; vin = vout; vout = tv;
  }
  if (runs && passes <= 2) {
    runs->valid = true;
    runs->totals_last = totals_pp[(passes - 1) & 1];
    runs->relative = passes == 2;
    runs->lo_bits = passes == 2 ? widths[0] : 0;
    runs->hi_bits = widths[passes - 1];
  }
  return (passes & 1) != 0;
}
bool launch_sort_pairs(uint64_t* keys_a, uint32_t* vals_a, uint64_t* keys_b, uint32_t* vals_b, uint32_t n,
                       int end_bit, void* scratch, hipStream_t s) {
  return sort_pairs_impl<uint64_t, uint32_t>(keys_a, vals_a, keys_b, vals_b, n, end_bit, scratch, s);
}
bool launch_sort_pairs_u32(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, uint32_t n,
                           int end_bit, void* scratch, hipStream_t s, const uint32_t* n_dev, SortedRuns* runs) {
  return sort_pairs_impl<uint32_t, uint32_t>(keys_a, vals_a, keys_b, vals_b, n, end_bit, scratch, s, n_dev, runs);
}
bool launch_sort_pairs_u32_v64(uint32_t* keys_a, uint2* vals_a, uint32_t* keys_b, uint2* vals_b, uint32_t n,
                               int end_bit, void* scratch, hipStream_t s, const uint32_t* n_dev) {
  return sort_pairs_impl<uint32_t, uint2>(keys_a, vals_a, keys_b, vals_b, n, end_bit, scratch, s, n_dev);
}

void launch_compact_visible(int P, const BinInfo* bin, const uint32_t* block_vis_offs, const uint32_t* block_offs,
                            uint32_t* slot_base, uint32_t* total, uint32_t capacity, int grid_x, uint32_t* dkey,
                            uint2* dval, hipStream_t s, bool top_pass_enqueued) {
  const int nb = (P + PRE_BLOCK - 1) / PRE_BLOCK;
  if (nb > 0)
    hipLaunchKernelGGL(compact_visible_kernel, dim3(nb), dim3(PRE_BLOCK), 0, s, P, bin, block_vis_offs, block_offs, slot_base,
                       total + TOTAL_DEPTH_INV_MIN, grid_x, dkey, dval, total, capacity, top_pass_enqueued ? 1 : 0);
}
// one more pass on bits [shift, shift + nbits) of 32-bit keys (nbits <= 8): the top digit of the depth sort
void launch_sort_extra_pass_u32(const uint32_t* kin, const uint2* vin, uint32_t* kout, uint2* vout, uint32_t n,
                                const uint32_t* n_dev, int shift, int nbits, void* scratch, hipStream_t s) {
  if (n == 0) return;
  const SortLayout L(n);
  uint32_t* hist = reinterpret_cast<uint32_t*>(static_cast<char*>(scratch) + L.hist);
  uint32_t* totals = reinterpret_cast<uint32_t*>(static_cast<char*>(scratch) + L.totals);
  sort_pass<uint32_t, uint2, 8>(kin, vin, kout, vout, n, n_dev, shift, nbits, L, hist, totals, s);
}
// v_cap: the number of visible Gaussians, or a capacity for it (the kernels read V from total[])
void launch_count_tiles(uint32_t v_cap, const uint32_t* total, const uint2* d3, const uint2* d4, const BinInfo* bin,
                        uint32_t* block_sums2, hipStream_t s) {
  const uint32_t nb = (v_cap + PRE_BLOCK - 1) / PRE_BLOCK;
  if (nb) hipLaunchKernelGGL(count_tiles_kernel, dim3(nb), dim3(PRE_BLOCK), 0, s, total, d3, d4, bin, block_sums2);
}
void launch_emit_instances(uint32_t v_cap, const uint32_t* total, int grid_x, const uint2* d3, const uint2* d4,
                           const BinInfo* bin, const uint32_t* block_offs2, uint32_t* inst_tile, uint32_t* inst_g,
                           uint32_t capacity, hipStream_t s) {
  const uint32_t nb = (v_cap + PRE_BLOCK - 1) / PRE_BLOCK;
  if (nb == 0) return;
  if (emit_is_wide(v_cap))      // fewer blocks than two per CU: four times the threads per block for the emission loop
    hipLaunchKernelGGL(emit_instances_kernel<1024>, dim3(nb), dim3(1024), 0, s, total, grid_x, d3, d4, bin, block_offs2,
                       inst_tile, inst_g, capacity);
  else
    hipLaunchKernelGGL(emit_instances_kernel<PRE_BLOCK>, dim3(nb), dim3(PRE_BLOCK), 0, s, total, grid_x, d3, d4, bin,
                       block_offs2, inst_tile, inst_g, capacity);
}
void launch_reconstruct_keys(uint32_t R, uint32_t P, const uint32_t* tile_sorted, const uint32_t* point_list,
                             const BinInfo* bin, uint64_t* keys, hipStream_t s) {
  if (R) hipLaunchKernelGGL(reconstruct_keys_kernel, dim3((R + 255) / 256), dim3(256), 0, s, R, P, tile_sorted, point_list, bin, keys);
}

void launch_build_tile_order(int tiles, const uint2* ranges, uint32_t* order, hipStream_t s) {
  hipLaunchKernelGGL(build_tile_order_kernel, dim3(1), dim3(1024), 0, s, tiles, ranges, order);
}
void launch_ranges_and_order_from_sort(int tiles, const SortedRuns& sr, uint2* ranges, uint32_t* order, hipStream_t s) {
  hipLaunchKernelGGL(ranges_and_order_from_sort_kernel, dim3(1), dim3(1024), 0, s, tiles, sr.totals_last, sr.lo_bits,
                     sr.hi_bits, sr.relative, ranges, order);
}

void launch_identify_tile_ranges(uint32_t R, const uint64_t* keys, uint2* ranges, hipStream_t s) {
  if (R == 0) return;
  hipLaunchKernelGGL((identify_tile_ranges_kernel<uint64_t, 32>), dim3((R + 255) / 256), dim3(256), 0, s, R,
                     (const uint32_t*)nullptr, keys, ranges);
}
void launch_identify_tile_ranges_u32(uint32_t R, const uint32_t* tiles, uint2* ranges, hipStream_t s,
                                     const uint32_t* r_dev) {
  if (R == 0) return;
  hipLaunchKernelGGL((identify_tile_ranges_kernel<uint32_t, 0>), dim3((R + 255) / 256), dim3(256), 0, s, R, r_dev, tiles,
                     ranges);
}

}  // namespace gsr
