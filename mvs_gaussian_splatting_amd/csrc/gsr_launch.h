// Host-side launchers, one per kernel; each is defined next to its kernel so that the
// translation units can be compiled with different floating-point contraction settings.
#pragma once
#include "gsr_common.h"

namespace gsr {

// preprocess.hip
// block_big / big_list: every block lists its Gaussians with more than ROWS_COOP instances at big_list[block * PRE_BLOCK ..]
// and writes how many (no counter to zero in front of the frame, no atomics); the scan launch below turns the counts
// into offsets
void launch_preprocess_fwd(const GsrParams& p, GeomRec* rec, BinInfo* bin, uint32_t* block_sums, uint32_t* block_vis,
                           int32_t* radii, uint32_t* block_big, uint32_t* big_list, uint2* block_range, hipStream_t s);
// folds the gradient rows of every listed Gaussian into its first row (wave-cooperative, fixed order); nb = blocks of
// preprocess_fwd, big_offs = the scanned block_big (nb + 1 entries)
void launch_sum_big_rows(const uint32_t* big_count, const uint32_t* big_offs, int nb, const uint32_t* big_list,
                         const GeomRec* rec, const uint32_t* slot_base, GradRow* rows, uint8_t* row_flags, hipStream_t s);
// exclusive scans of up to three per-block arrays in one launch (block 0: a, block 1: b, block 3: c; block 2 folds
// block_range, and c comes only with it); total_x = grand total
void launch_scan_block_sums(const uint32_t* sums_a, uint32_t* offs_a, uint32_t* total_a, const uint32_t* sums_b,
                            uint32_t* offs_b, uint32_t* total_b, int nb, hipStream_t s,
                            uint32_t* host_mirror = nullptr, const uint2* block_range = nullptr,
                            const uint32_t* sums_c = nullptr, uint32_t* offs_c = nullptr, uint32_t* total_c = nullptr);
void launch_preprocess_bwd(const GsrParams& p, const int32_t* radii, const GeomRec* rec, const uint32_t* slot_base,
                           const GradRow* rows,
                           const uint8_t* row_flags, const GsrGrads& g, hipStream_t s);

// binning.hip
void launch_duplicate_with_keys(int P, int grid_x, const BinInfo* bin, const uint32_t* block_offs, uint32_t* slot_base,
                                uint32_t* point_offsets, uint64_t* keys, uint32_t* vals, hipStream_t s);
// returns true when the sorted result ended in (keys_b, vals_b)
bool launch_sort_pairs(uint64_t* keys_a, uint32_t* vals_a, uint64_t* keys_b, uint32_t* vals_b, uint32_t n,
                       int end_bit, void* scratch, hipStream_t s);
// n_dev != NULL: the element count is read from device memory and n is the capacity the grid is sized for
// In / out of a sort that also delivers where every key value's run lies in the sorted array, without a pass over it
// (binning.hip: seg_block, radix_rowscan_kernel, ranges_and_order_from_sort_kernel).
struct SortedRuns {
  uint2* runs_rel;               // in: [n_keys] array the last pass's row scan writes RELATIVE runs into (two-pass sorts)
  uint32_t n_keys;               // in: number of key values (tiles)
  bool valid;                    // out: false = more than two passes (use identify_tile_ranges)
  bool relative;                 // out: runs_rel was written (two passes); false: one pass, the runs are the digit totals' scan
  const uint32_t* totals_last;   // out: digit totals of the last pass
  int lo_bits, hi_bits;          // out: digit widths of the first / last pass (lo_bits = 0: one pass)
};
// runs != NULL: sorts of at most two passes run their last pass segmented and fill *runs
bool launch_sort_pairs_u32(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, uint32_t n,
                           int end_bit, void* scratch, hipStream_t s, const uint32_t* n_dev = nullptr,
                           SortedRuns* runs = nullptr);
// r_dev != NULL: the instance count is read from device memory and R is the capacity the grid is sized for
void launch_identify_tile_ranges_u32(uint32_t R, const uint32_t* tiles, uint2* ranges, hipStream_t s,
                                     const uint32_t* r_dev = nullptr);
inline int sort_passes(int end_bit) { return (end_bit + RADIX_BITS - 1) / RADIX_BITS; }
void launch_identify_tile_ranges(uint32_t R, const uint64_t* keys, uint2* ranges, hipStream_t s);
void launch_build_tile_order(int tiles, const uint2* ranges, uint32_t* order, hipStream_t s);
// two-level binning: tile ranges ((0,0) for empty tiles) from the tile sort's histogram + the tile order, one kernel
void launch_ranges_and_order_from_sort(int tiles, const SortedRuns& sr, uint2* ranges, uint32_t* order, hipStream_t s);
// also derives the device-side counts of the later stages (GeomLayout::total[TOTAL_TOP_PASS_N / TOTAL_R_CLAMPED]);
// capacity: instances the binning workspace holds (0xffffffff when the host sizes it from the real count)
// top_pass_enqueued = false: total[TOTAL_TOP_PASS_N] stays 0 whatever the frame's depth span (nobody will run the pass)
void launch_compact_visible(int P, const BinInfo* bin, const uint32_t* block_vis_offs, const uint32_t* block_offs,
                            uint32_t* slot_base, uint32_t* total, uint32_t capacity, int grid_x, uint32_t* dkey,
                            uint2* dval, hipStream_t s, bool top_pass_enqueued = true);
// (32-bit key, 64-bit value) pairs: the depth sort, whose payload is (index, packed rect)
bool launch_sort_pairs_u32_v64(uint32_t* keys_a, uint2* vals_a, uint32_t* keys_b, uint2* vals_b, uint32_t n,
                               int end_bit, void* scratch, hipStream_t s, const uint32_t* n_dev = nullptr);
void launch_sort_extra_pass_u32(const uint32_t* kin, const uint2* vin, uint32_t* kout, uint2* vout, uint32_t n,
                                const uint32_t* n_dev, int shift, int nbits, void* scratch, hipStream_t s);
// The two-level binning sorts the visible Gaussians on (depth bits - smallest depth bits of the frame).  Three 8-bit
// passes (24 bits: up to two binades of depth, e.g. 3 .. 12) are enqueued before the host knows the counts; a frame
// that spans more gets the fourth 8-bit pass on bits 24..31: its kernels take their element count from
// total[TOTAL_TOP_PASS_N] (V or 0, set on the device), and the consumers pick the buffer the result ended in.
// (Three 9-bit passes were measured too: as slow as four 8-bit ones -- wider digits rank and scatter more slowly.)
// d3 / d4: the depth-sorted payload after the three regular passes / after the top-digit pass (chosen on the device)
void launch_count_tiles(uint32_t v_cap, const uint32_t* total, const uint2* d3, const uint2* d4, const BinInfo* bin,
                        uint32_t* block_sums2, hipStream_t s);
// Small frames (at most EMIT_WIDE_MAX_BLOCKS blocks of 256 depth-sorted Gaussians): emit_instances runs with 1024 threads
// per block and takes the block TOTALS of count_tiles as block_offs2 (it sums the totals in front of each block itself):
// the caller then skips the scan between the two kernels.
constexpr uint32_t EMIT_WIDE_MAX_BLOCKS = 512;
inline bool emit_is_wide(uint32_t v_cap) { return (v_cap + PRE_BLOCK - 1) / PRE_BLOCK <= EMIT_WIDE_MAX_BLOCKS; }
void launch_emit_instances(uint32_t v_cap, const uint32_t* total, int grid_x, const uint2* d3, const uint2* d4,
                           const BinInfo* bin, const uint32_t* block_offs2, uint32_t* inst_tile, uint32_t* inst_g,
                           uint32_t capacity, hipStream_t s);
void launch_reconstruct_keys(uint32_t R, uint32_t P, const uint32_t* tile_sorted, const uint32_t* point_list,
                             const BinInfo* bin, uint64_t* keys, hipStream_t s);

// render.hip
void launch_render_fwd(int W, int H, const uint2* ranges, const uint32_t* point_list, const GeomRec* rec,
                       const float* bg, float* out_color, float* final_T, uint32_t* n_contrib, uint32_t* tile_max,
                       const uint32_t* tile_order, hipStream_t s, unsigned long long* stats, int cull, uint16_t* inst_mask);
void launch_render_bwd(int W, int H, const uint2* ranges, const uint32_t* point_list, const GeomRec* rec,
                       const uint32_t* slot_base,
                       const float* bg, const float* final_T, const uint32_t* n_contrib, const uint32_t* tile_max,
                       const float* dL_dpix, GradRow* rows, uint8_t* row_flags, const uint32_t* tile_order,
                       hipStream_t s, const uint16_t* inst_mask);

// loss.hip
void launch_l1_dssim(const float* x, const float* gt, int C, int H, int W, float lambda, int dssim_mode, float* sums,
                     float* dL_dx, float* maps, hipStream_t s);

// splat2d.hip (BASELINE config 1)
struct Splat2dLayout {
  size_t rec, flag, pre, gmask, partial, bytes;
  int chunks, per_chunk;
  Splat2dLayout(int N, int H, int W);
};
int splat2d_max_kernel_size();
void launch_splat2d_fwd(int N, int K, int H, int W, const float* sx, const float* sy, const float* rho,
                        const float* coords, const float* colours, const float* ax, void* ws, float* out,
                        hipStream_t s);
void launch_splat2d_bwd(int N, int K, int H, int W, const float* sx, const float* sy, const float* rho, const float* ax,
                        void* ws, const float* dL_dout, float* d_sx, float* d_sy, float* d_rho, float* d_coords,
                        float* d_colours, hipStream_t s);

// densify.hip.  counts = {kept originals, kept clones, kept children per copy, split-selected}
struct DensifyLayout {
  size_t flags, block_counts, block_offs, totals, pos, bytes;
  int nblocks;
  explicit DensifyLayout(int P);
};
void launch_densify_plan(int P, const float* accum, const float* denom, const float* scaling, const float* opacity,
                         float thr, float pde, float min_opacity, float ws_limit, int use_ws, void* ws, hipStream_t s);
void launch_densify_gather_rows(int P, int w, const float* src, const void* ws, const uint32_t counts[4], int zero_new,
                                float* dst, hipStream_t s);
void launch_densify_split_children(int P, const float* xyz, const float* scaling, const float* rotation,
                                   const float* noise, const void* ws, const uint32_t counts[4], float* dst_xyz,
                                   float* dst_scaling, hipStream_t s);

// knn.hip
size_t knn_workspace_bytes(int N);
void launch_knn3(const float* pts, int N, float* mean_dist2, void* ws, hipStream_t s);

// aux.hip
size_t l1_loss_workspace_bytes();
void launch_l1_loss(const float* x, const float* gt, size_t n, float scale, float* loss_sum, float* dL_dx,
                    float* partials, hipStream_t s);
void launch_densify_stats(int P, const float* dL_dmeans2D, const int32_t* radii, float* accum, float* denom,
                          float* max_radii2D, hipStream_t s);
void launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* visible, hipStream_t s);
void launch_unpack_geom(int P, const GeomRec* rec, const BinInfo* bin, const uint32_t* block_offs, float* xy,
                        float* conic_opacity, float* rgb, float* depth, uint32_t* tiles, uint32_t* point_offsets,
                        uint32_t* rect, uint32_t* clamped, hipStream_t s);

}  // namespace gsr
