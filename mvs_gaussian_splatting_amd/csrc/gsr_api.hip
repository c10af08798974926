// extern "C" entry points of libgsr_hip.so (declared in include/gsr.h).  Host logic only:
// argument validation, workspace carving, kernel sequencing on the caller's stream.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "gsr_common.h"
#include "gsr_launch.h"

using namespace gsr;

namespace {
thread_local std::string g_err = "";

int fail(int code, const char* msg) {
  g_err = msg;
  return code;
}
int hip_fail(hipError_t e, const char* where) {
  g_err = std::string(where) + ": " + hipGetErrorString(e);
  return (int)e;
}
#define GSR_HIP(expr)                                        \
  do {                                                       \
    hipError_t _e = (expr);                                  \
    if (_e != hipSuccess) return hip_fail(_e, #expr);        \
  } while (0)

// after a batch of launches: surface launch errors; in debug mode also synchronise
int check(const GsrParams* p, hipStream_t s, const char* where) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, where);
  if (p && p->debug) {
    e = hipStreamSynchronize(s);
    if (e != hipSuccess) return hip_fail(e, where);
  }
  return 0;
}

int validate(const GsrParams* p) {
  if (!p) return fail(GSR_E_BADARG, "params is NULL");
  if (p->P < 0 || p->width <= 0 || p->height <= 0) return fail(GSR_E_BADARG, "bad P / image size");
  if (p->width > 8191 * TILE || p->height > 8191 * TILE) return fail(GSR_E_BADARG, "image too large (13-bit tile coordinates)");
  if (p->P == 0) return 0;
  if (!p->means3D || !p->opacities || !p->viewmatrix || !p->projmatrix || !p->bg)
    return fail(GSR_E_BADARG, "means3D / opacities / viewmatrix / projmatrix / bg must be non-NULL");
  if ((p->shs == nullptr) == (p->colors_precomp == nullptr))
    return fail(GSR_E_BADARG, "provide exactly one of shs / colors_precomp");
  const bool sr = p->scales != nullptr || p->rotations != nullptr;
  if (sr && (p->scales == nullptr || p->rotations == nullptr))
    return fail(GSR_E_BADARG, "scales and rotations must be given together");
  if (sr == (p->cov3D_precomp != nullptr))
    return fail(GSR_E_BADARG, "provide exactly one of (scales, rotations) / cov3D_precomp");
  if (p->shs) {
    if (p->D < 0 || p->D > 3) return fail(GSR_E_BADARG, "sh degree must be 0..3");
    if (p->M < (p->D + 1) * (p->D + 1)) return fail(GSR_E_BADARG, "M smaller than (D+1)^2");
    if (!p->campos) return fail(GSR_E_BADARG, "campos required with shs");
    if (((uintptr_t)p->shs & 15u) != 0 && p->M == 16 && !p->shs_rest)
      return fail(GSR_E_ALIGN, "shs must be 16-byte aligned");
    if (p->shs_rest) {
      if (p->M != 16) return fail(GSR_E_BADARG, "split SH inputs (shs_rest) require M == 16");
      if (((uintptr_t)p->shs_rest & 15u) != 0) return fail(GSR_E_ALIGN, "shs_rest must be 16-byte aligned");
    }
  } else if (p->shs_rest) {
    return fail(GSR_E_BADARG, "shs_rest given without shs");
  }
  if (p->binning_mode != GSR_BINNING_TWO_LEVEL && p->binning_mode != GSR_BINNING_KEYS64 &&
      p->binning_mode != GSR_BINNING_TWO_LEVEL_CULLED)
    return fail(GSR_E_BADARG, "unknown binning_mode");
  if ((p->act_flags & (GSR_ACT_SCALE_EXP | GSR_ACT_ROT_NORMALIZE)) && !p->scales)
    return fail(GSR_E_BADARG, "scale / rotation activations need the scales + rotations inputs");
  if (p->rotations && ((uintptr_t)p->rotations & 15u) != 0) return fail(GSR_E_ALIGN, "rotations must be 16-byte aligned");
  return 0;
}

// ---- opt-in stage timers: event pairs recorded on the caller's stream ----------------------------
// Forward and backward of one frame arrive on different OS threads (autograd engine): every access takes `mu`.
struct Profile {
  struct Span { int stage; hipEvent_t a, b; };
  std::mutex mu;
  std::vector<Span> used;
  std::vector<hipEvent_t> pool;
  hipEvent_t get() {
    std::lock_guard<std::mutex> lock(mu);
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
  }
  void push(const Span& sp) {
    std::lock_guard<std::mutex> lock(mu);
    used.push_back(sp);
  }
};
// The count read-back of the two-call forward waits on one event per frame: it is created once per (thread, device)
// and lives as long as the thread (a fresh hipEventCreate / Destroy per frame cost ~10 us of host time).
struct ThreadEvent {
  int dev = -1;
  hipEvent_t e = nullptr;
  ~ThreadEvent() { if (e) (void)hipEventDestroy(e); }
  hipError_t get(hipEvent_t* out) {
    int d = 0;
    hipError_t err = hipGetDevice(&d);
    if (err != hipSuccess) return err;
    if (e && d != dev) { (void)hipEventDestroy(e); e = nullptr; }
    if (!e) {
      err = hipEventCreateWithFlags(&e, hipEventDisableTiming);
      if (err != hipSuccess) { e = nullptr; return err; }
      dev = d;
    }
    *out = e;
    return hipSuccess;
  }
};
thread_local ThreadEvent g_count_event;

// ---- opt-in roctx ranges: resolved at run time, so that the library has no link-time dependency
// on a profiler; one process-wide switch (gsr_enable_markers)
typedef int (*roctx_push_fn)(const char*);
typedef int (*roctx_pop_fn)(void);
std::atomic<roctx_push_fn> g_roctx_push{nullptr};
std::atomic<roctx_pop_fn> g_roctx_pop{nullptr};
std::mutex g_roctx_mu;
const char* const kStageNames[GSR_STAGE_COUNT] = {"preprocess_fwd", "scan_block_sums", "duplicate_with_keys", "radix_sort",
                                                  "identify_tile_ranges", "render_fwd", "render_bwd", "preprocess_bwd"};
const char* const kStageMarkers[GSR_STAGE_COUNT] = {"gsr:preprocess_fwd", "gsr:scan_block_sums", "gsr:duplicate_with_keys",
                                                    "gsr:radix_sort", "gsr:identify_tile_ranges", "gsr:render_fwd",
                                                    "gsr:render_bwd", "gsr:preprocess_bwd"};

// one stage of a call: HIP-event pair on the stream when a profile handle is attached, roctx range when markers are on
struct StageTimer {
  Profile* pr; hipStream_t s; Profile::Span sp; roctx_pop_fn pop;
  StageTimer(const GsrParams* p, int stage, hipStream_t st)
      : pr(p ? static_cast<Profile*>(p->profile) : nullptr), s(st), pop(nullptr) {
    if (roctx_push_fn push = g_roctx_push.load(std::memory_order_acquire)) {
      (void)push(kStageMarkers[stage]);
      pop = g_roctx_pop.load(std::memory_order_acquire);
    }
    if (!pr) return;
    sp.stage = stage; sp.a = pr->get(); sp.b = pr->get();
    if (sp.a) (void)hipEventRecord(sp.a, s);
  }
  ~StageTimer() {
    if (pr) {
      if (sp.b) (void)hipEventRecord(sp.b, s);
      pr->push(sp);
    }
    if (pop) (void)pop();
  }
};

template <typename T>
T* at(void* base, size_t off) { return reinterpret_cast<T*>(static_cast<char*>(base) + off); }
template <typename T>
const T* at(const void* base, size_t off) { return reinterpret_cast<const T*>(static_cast<const char*>(base) + off); }

// where the sorted point list (and the sorted tile ids / keys) of a frame ended up
struct SortedViews {
  const uint32_t* point_list;
  const uint32_t* tile_sorted;   // mode 0
  const uint64_t* keys_sorted;   // mode 1
  // the sort's other payload buffer, free once the sort is done: render_fwd leaves the mini-block reach mask of every
  // instance (list order, 2 bytes each) there for render_bwd
  uint16_t* inst_mask;
};
SortedViews sorted_views(const void* bin_ws, uint32_t R, uint32_t V, int W, int H, int mode) {
  const ImageLayout I(W, H);
  const BinLayout B(R, V, mode);
  SortedViews v{nullptr, nullptr, nullptr, nullptr};
  char* ws = const_cast<char*>(static_cast<const char*>(bin_ws));
  if (mode == GSR_BINNING_KEYS64) {
    const bool in_b = (sort_passes(32 + tile_bits(I.tiles)) & 1) != 0;
    v.point_list = at<uint32_t>(bin_ws, in_b ? B.vals_b : B.vals_a);
    v.keys_sorted = at<uint64_t>(bin_ws, in_b ? B.keys_b : B.keys_a);
    v.inst_mask = reinterpret_cast<uint16_t*>(ws + (in_b ? B.vals_a : B.vals_b));
  } else {
    const bool in_b = (sort_passes(tile_sort_bits(I.tiles)) & 1) != 0;
    v.point_list = at<uint32_t>(bin_ws, in_b ? B.ig_b : B.ig_a);
    v.tile_sorted = at<uint32_t>(bin_ws, in_b ? B.itile_b : B.itile_a);
    v.inst_mask = reinterpret_cast<uint16_t*>(ws + (in_b ? B.ig_a : B.ig_b));
  }
  return v;
}
}  // namespace

extern "C" {

int gsr_abi_version(void) { return GSR_ABI_VERSION; }
const char* gsr_last_error(void) { return g_err.c_str(); }
const char* gsr_build_info(void) { return "libgsr_hip gfx950 wave64 tile16 radix6-9 binning:culled|two_level|keys64 (HIP " __DATE__ ")"; }

size_t gsr_geom_bytes(int32_t P) { return GeomLayout(P < 0 ? 0 : P).bytes; }
size_t gsr_image_bytes(int32_t width, int32_t height) { return ImageLayout(width, height).bytes; }
size_t gsr_binning_bytes(uint32_t num_rendered, uint32_t num_visible, int32_t, int32_t, int32_t mode) {
  return BinLayout(num_rendered, num_visible, mode).bytes;
}
size_t gsr_backward_bytes(int32_t P, uint32_t num_rendered) { return BwdLayout(P, num_rendered).bytes; }
size_t gsr_sort_scratch_bytes(uint32_t n) { return SortLayout(n).bytes; }

}  // extern "C"

static void enqueue_depth_top_pass(const GsrParams* p, void* geom_ws, hipStream_t s);

// Stage 1 on the stream: preprocess, scan of the block totals (counts -> total[] and the pinned mirror), `counted`
// recorded behind the scan, then -- two-level modes -- compaction of the visible Gaussians and their depth sort.
// None of it needs a host-side count: the grids are sized for P and the kernels read V from the device.
// top_pass: WITH = enqueued here, decides on the device (gsr_forward); LATER = the host enqueues it behind stage 1 for the
// frames that need it (two-call path, which knows the span); NEVER = the caller vouches for a narrow frame
// (GsrParams::depth_span_lt24): the device-side switch stays off whatever the span.
enum TopPass { TOP_PASS_WITH, TOP_PASS_LATER, TOP_PASS_NEVER };
static int enqueue_stage1(const GsrParams* p, void* geom_ws, int32_t* radii, hipStream_t s, uint32_t capacity,
                          hipEvent_t counted, TopPass top_pass) {
  const GeomLayout L(p->P);
  {
    StageTimer t(p, GSR_STAGE_PREPROCESS_FWD, s);
    launch_preprocess_fwd(*p, at<GeomRec>(geom_ws, L.rec), at<BinInfo>(geom_ws, L.bin),
                          at<uint32_t>(geom_ws, L.block_sums), at<uint32_t>(geom_ws, L.block_vis), radii,
                          at<uint32_t>(geom_ws, L.block_big), at<uint32_t>(geom_ws, L.big_list),
                          at<uint2>(geom_ws, L.block_range), s);
  }
  if (int rc = check(p, s, "preprocess_fwd")) return rc;
  uint32_t* total = at<uint32_t>(geom_ws, L.total);
  {
    StageTimer t(p, GSR_STAGE_SCAN, s);
    launch_scan_block_sums(at<uint32_t>(geom_ws, L.block_sums), at<uint32_t>(geom_ws, L.block_offs), total + TOTAL_R,
                           at<uint32_t>(geom_ws, L.block_vis), at<uint32_t>(geom_ws, L.block_vis_offs), total + TOTAL_V,
                           L.nblocks, s, p->counts_pinned, at<uint2>(geom_ws, L.block_range),
                           at<uint32_t>(geom_ws, L.block_big), at<uint32_t>(geom_ws, L.block_big_offs), total + TOTAL_BIG);
  }
  if (int rc = check(p, s, "scan_block_sums")) return rc;
  if (counted) GSR_HIP(hipEventRecord(counted, s));
  if (p->binning_mode != GSR_BINNING_KEYS64) {
    StageTimer t(p, GSR_STAGE_SORT, s);
    launch_compact_visible(p->P, at<BinInfo>(geom_ws, L.bin), at<uint32_t>(geom_ws, L.block_vis_offs),
                           at<uint32_t>(geom_ws, L.block_offs),
                           p->forward_only ? nullptr : at<uint32_t>(geom_ws, L.slot_base), total, capacity,
                           (p->width + TILE - 1) / TILE, at<uint32_t>(geom_ws, L.dkey_a), at<uint2>(geom_ws, L.didx_a), s,
                           top_pass != TOP_PASS_NEVER);
    launch_sort_pairs_u32_v64(at<uint32_t>(geom_ws, L.dkey_a), at<uint2>(geom_ws, L.didx_a),
                              at<uint32_t>(geom_ws, L.dkey_b), at<uint2>(geom_ws, L.didx_b), (uint32_t)p->P,
                              DEPTH_SORT_BITS, at<char>(geom_ws, L.dsort), s, total + TOTAL_V);
    if (top_pass == TOP_PASS_WITH) enqueue_depth_top_pass(p, geom_ws, s);
  }
  return check(p, s, "depth_sort");
}

// The depth sort's fourth pass (bits 24..31 of the relative depth key).  Its element count is total[TOTAL_TOP_PASS_N]:
// V for a frame that spans more than 2^24 float32 steps of depth, 0 otherwise (the three kernels then exit at once).
// The consumers (count_tiles / emit_instances) read the same word to pick the buffer the sorted payload ended in.
static void enqueue_depth_top_pass(const GsrParams* p, void* geom_ws, hipStream_t s) {
  const GeomLayout L(p->P);
  const bool in_b = (sort_passes(DEPTH_SORT_BITS) & 1) != 0;
  launch_sort_extra_pass_u32(at<uint32_t>(geom_ws, in_b ? L.dkey_b : L.dkey_a), at<uint2>(geom_ws, in_b ? L.didx_b : L.didx_a),
                             at<uint32_t>(geom_ws, in_b ? L.dkey_a : L.dkey_b), at<uint2>(geom_ws, in_b ? L.didx_a : L.didx_b),
                             (uint32_t)p->P, at<uint32_t>(geom_ws, L.total) + TOTAL_TOP_PASS_N, DEPTH_SORT_BITS,
                             32 - DEPTH_SORT_BITS, at<char>(geom_ws, L.dsort), s);
}

// Stage 2 on the stream: instance emission, tile sort, tile ranges, compositing.  (r_cap, v_cap) are what the binning
// workspace is laid out for: the real counts in the two-call forward (device_counts = false), the caller's capacity
// and P in gsr_forward (device_counts = true: every kernel takes the real counts from total[]).
static int enqueue_stage2(const GsrParams* p, void* geom_ws, void* bin_ws, size_t bin_ws_bytes, void* img_ws,
                          uint32_t r_cap, uint32_t v_cap, bool device_counts, float* out_color, hipStream_t s) {
  const ImageLayout I(p->width, p->height);
  uint2* ranges = at<uint2>(img_ws, I.ranges);
  // two-level modes: the ranges come out of the tile sort's own histogram (launch_ranges_and_order_from_sort: no memset,
  // no pass over the sorted keys); 64-bit key mode, or nothing to bin: upstream's identifyTileRanges into a zero-filled array
  SortedRuns runs;
  runs.valid = false;
  runs.runs_rel = ranges;
  runs.n_keys = (uint32_t)I.tiles;
  const uint32_t* point_list = nullptr;
  uint16_t* inst_mask = nullptr;     // the sort's spare payload buffer (see SortedViews)
  const GeomRec* rec = nullptr;
  if (p->P > 0 && r_cap > 0) {
    if (!geom_ws || !bin_ws) return fail(GSR_E_BADARG, "geom_ws / bin_ws is NULL");
    if (v_cap == 0 || v_cap > (uint32_t)p->P || (!device_counts && v_cap > r_cap))
      return fail(GSR_E_BADARG, "num_visible inconsistent with num_rendered / P");
    const int mode = p->binning_mode;
    const BinLayout B(r_cap, v_cap, mode);
    if (bin_ws_bytes < B.bytes) return fail(GSR_E_CAPACITY, "binning workspace too small for num_rendered / num_visible");
    if (((uintptr_t)bin_ws & 255u) != 0) return fail(GSR_E_ALIGN, "bin_ws must be 256-byte aligned");
    const GeomLayout L(p->P);
    rec = at<GeomRec>(geom_ws, L.rec);
    const BinInfo* bin = at<BinInfo>(geom_ws, L.bin);
    const uint32_t* total = at<uint32_t>(geom_ws, L.total);
    const int tb = mode == GSR_BINNING_KEYS64 ? tile_bits(I.tiles) : tile_sort_bits(I.tiles);
    if (mode == GSR_BINNING_KEYS64) {
      if (device_counts) return fail(GSR_E_BADARG, "gsr_forward supports the two-level binning modes only");
      uint64_t* ka = at<uint64_t>(bin_ws, B.keys_a);
      uint64_t* kb = at<uint64_t>(bin_ws, B.keys_b);
      uint32_t* va = at<uint32_t>(bin_ws, B.vals_a);
      uint32_t* vb = at<uint32_t>(bin_ws, B.vals_b);
      {
        StageTimer t(p, GSR_STAGE_DUPLICATE, s);
        launch_duplicate_with_keys(p->P, I.grid_x, bin, at<uint32_t>(geom_ws, L.block_offs),
                                   at<uint32_t>(geom_ws, L.slot_base), at<uint32_t>(geom_ws, L.offsets), ka, va, s);
      }
      if (int rc = check(p, s, "duplicate_with_keys")) return rc;
      bool in_b;
      {
        StageTimer t(p, GSR_STAGE_SORT, s);
        in_b = launch_sort_pairs(ka, va, kb, vb, r_cap, 32 + tb, at<char>(bin_ws, B.sort), s);
      }
      if (int rc = check(p, s, "sort_pairs")) return rc;
      {
        StageTimer t(p, GSR_STAGE_RANGES, s);
        GSR_HIP(hipMemsetAsync(ranges, 0, 8 * (size_t)I.tiles, s));
        launch_identify_tile_ranges(r_cap, in_b ? kb : ka, ranges, s);
      }
      point_list = in_b ? vb : va;
      inst_mask = reinterpret_cast<uint16_t*>(in_b ? va : vb);
    } else {
      uint32_t* ita = at<uint32_t>(bin_ws, B.itile_a);
      uint32_t* itb = at<uint32_t>(bin_ws, B.itile_b);
      uint32_t* iga = at<uint32_t>(bin_ws, B.ig_a);
      uint32_t* igb = at<uint32_t>(bin_ws, B.ig_b);
      uint32_t* bsum2 = at<uint32_t>(bin_ws, B.bsum2);
      uint32_t* boffs2 = at<uint32_t>(bin_ws, B.boffs2);
      // depth-sorted payload: produced by stage 1 in the geometry workspace, in d3 or (after the top-digit pass) in d4
      const bool in_b3 = (sort_passes(DEPTH_SORT_BITS) & 1) != 0;
      const uint2* d3 = at<uint2>(geom_ws, in_b3 ? L.didx_b : L.didx_a);
      const uint2* d4 = at<uint2>(geom_ws, in_b3 ? L.didx_a : L.didx_b);
      const uint32_t cap = device_counts ? r_cap : 0xffffffffu;
      {
        StageTimer t(p, GSR_STAGE_DUPLICATE, s);   // instances emitted in depth order
        launch_count_tiles(v_cap, total, d3, d4, bin, bsum2, s);
        if (emit_is_wide(v_cap)) {      // small frame: the emission sums the block totals itself, no scan launch
          launch_emit_instances(v_cap, total, I.grid_x, d3, d4, bin, bsum2, ita, iga, cap, s);
        } else {
          launch_scan_block_sums(bsum2, boffs2, boffs2 + B.nblocks2 + 1, nullptr, nullptr, nullptr, (int)B.nblocks2, s);
          launch_emit_instances(v_cap, total, I.grid_x, d3, d4, bin, boffs2, ita, iga, cap, s);
        }
      }
      if (int rc = check(p, s, "emit_instances")) return rc;
      const uint32_t* r_dev = device_counts ? total + TOTAL_R_CLAMPED : nullptr;
      bool in_b;
      {
        StageTimer t(p, GSR_STAGE_SORT, s);     // stable partition by tile id
        in_b = launch_sort_pairs_u32(ita, iga, itb, igb, r_cap, tb, at<char>(bin_ws, B.sort), s, r_dev, &runs);
      }
      if (int rc = check(p, s, "tile_sort")) return rc;
      if (!runs.valid) {       // more than 2^18 tiles: three passes -- read the ranges off the sorted keys
        StageTimer t(p, GSR_STAGE_RANGES, s);
        GSR_HIP(hipMemsetAsync(ranges, 0, 8 * (size_t)I.tiles, s));
        launch_identify_tile_ranges_u32(r_cap, in_b ? itb : ita, ranges, s, r_dev);
      }
      point_list = in_b ? igb : iga;
      inst_mask = reinterpret_cast<uint16_t*>(in_b ? iga : igb);
    }
    if (int rc = check(p, s, "identify_tile_ranges")) return rc;
  } else {
    GSR_HIP(hipMemsetAsync(ranges, 0, 8 * (size_t)I.tiles, s));      // nothing to bin: every tile is empty
  }
  {
    StageTimer t(p, GSR_STAGE_RANGES, s);
    if (runs.valid) launch_ranges_and_order_from_sort(I.tiles, runs, ranges, at<uint32_t>(img_ws, I.tile_order), s);
    else launch_build_tile_order(I.tiles, ranges, at<uint32_t>(img_ws, I.tile_order), s);
  }
  {
    StageTimer t(p, GSR_STAGE_RENDER_FWD, s);
    const bool track = !p->forward_only;
    launch_render_fwd(p->width, p->height, ranges, point_list, rec, p->bg, out_color,
                      track ? at<float>(img_ws, I.final_T) : nullptr, track ? at<uint32_t>(img_ws, I.n_contrib) : nullptr,
                      track ? at<uint32_t>(img_ws, I.tile_max) : nullptr, at<uint32_t>(img_ws, I.tile_order), s, nullptr,
                      (p->debug_flags & GSR_DEBUG_NO_MINIBLOCK_CULL) ? 0 : 1, track ? inst_mask : nullptr);
  }
  return check(p, s, "render_fwd");
}

extern "C" {

int gsr_forward_preprocess(const GsrParams* p, void* geom_ws, int32_t* radii, void* stream, uint32_t* num_rendered,
                           uint32_t* num_visible) {
  if (int rc = validate(p)) return rc;
  if (!num_rendered || !num_visible) return fail(GSR_E_BADARG, "num_rendered / num_visible is NULL");
  *num_rendered = 0;
  *num_visible = 0;
  if (p->P == 0) return 0;
  if (!geom_ws || !radii) return fail(GSR_E_BADARG, "geom_ws / radii is NULL");
  if (((uintptr_t)geom_ws & 255u) != 0) return fail(GSR_E_ALIGN, "geom_ws must be 256-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const GeomLayout L(p->P);
  hipEvent_t counted = nullptr;
  if (p->counts_pinned) GSR_HIP(g_count_event.get(&counted));
  // the first half of the two-level binning is enqueued before the read-back, so that the GPU sorts while the host
  // round-trips; the top-digit pass follows only for the frames that need it (the host knows once the counts are in)
  if (int rc = enqueue_stage1(p, geom_ws, radii, s, 0xffffffffu, counted, TOP_PASS_LATER)) return rc;
  uint32_t depth_min = 0, depth_max = 0;
  if (counted) {
    const hipError_t e = hipEventSynchronize(counted);     // waits for the scan kernel only
    if (e != hipSuccess) return hip_fail(e, "hipEventSynchronize(counted)");
    *num_rendered = p->counts_pinned[0];
    *num_visible = p->counts_pinned[1];
    depth_min = p->counts_pinned[2];
    depth_max = p->counts_pinned[3];
  } else {
    uint32_t host[6] = {0, 0, 0, 0, 0, 0};
    GSR_HIP(hipMemcpyAsync(host, at<uint32_t>(geom_ws, L.total), sizeof(host), hipMemcpyDeviceToHost, s));
    GSR_HIP(hipStreamSynchronize(s));
    *num_rendered = host[TOTAL_R];
    *num_visible = host[TOTAL_V];
    depth_min = ~host[TOTAL_DEPTH_INV_MIN];
    depth_max = host[TOTAL_DEPTH_MAX];
  }
  if (p->binning_mode != GSR_BINNING_KEYS64 && *num_visible > 0 && depth_max >= depth_min &&
      ((uint64_t)depth_max - depth_min) >> DEPTH_SORT_BITS) {
    StageTimer t(p, GSR_STAGE_SORT, s);
    enqueue_depth_top_pass(p, geom_ws, s);
    if (int rc = check(p, s, "depth_sort_top_digit")) return rc;
  }
  return 0;
}

int gsr_forward_render(const GsrParams* p, void* geom_ws, void* bin_ws, size_t bin_ws_bytes, void* img_ws,
                       uint32_t R, uint32_t V, float* out_color, void* stream) {
  if (int rc = validate(p)) return rc;
  if (!img_ws || !out_color) return fail(GSR_E_BADARG, "img_ws / out_color is NULL");
  return enqueue_stage2(p, geom_ws, bin_ws, bin_ws_bytes, img_ws, R, V, false, out_color, static_cast<hipStream_t>(stream));
}

int gsr_forward(const GsrParams* p, void* geom_ws, void* bin_ws, size_t bin_ws_bytes, uint32_t capacity, void* img_ws,
                int32_t* radii, float* out_color, void* counts_event, void* stream) {
  if (int rc = validate(p)) return rc;
  if (!img_ws || !out_color) return fail(GSR_E_BADARG, "img_ws / out_color is NULL");
  if (p->P > 0 && p->binning_mode == GSR_BINNING_KEYS64)
    return fail(GSR_E_BADARG, "gsr_forward supports the two-level binning modes only");
  if (p->P > 0 && !p->counts_pinned)
    return fail(GSR_E_BADARG, "gsr_forward needs counts_pinned (the caller detects an overflow of `capacity` there)");
  if (p->P > 0 && capacity == 0) return fail(GSR_E_BADARG, "capacity is 0");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (p->P > 0) {
    if (!geom_ws || !radii) return fail(GSR_E_BADARG, "geom_ws / radii is NULL");
    if (((uintptr_t)geom_ws & 255u) != 0) return fail(GSR_E_ALIGN, "geom_ws must be 256-byte aligned");
    // the depth sort's fourth pass: enqueued (and decided on the device) unless the caller vouches for a narrow frame
    if (int rc = enqueue_stage1(p, geom_ws, radii, s, capacity, static_cast<hipEvent_t>(counts_event),
                                p->depth_span_lt24 ? TOP_PASS_NEVER : TOP_PASS_WITH))
      return rc;
  } else if (counts_event) {
    GSR_HIP(hipEventRecord(static_cast<hipEvent_t>(counts_event), s));
  }
  return enqueue_stage2(p, geom_ws, bin_ws, bin_ws_bytes, img_ws, p->P > 0 ? capacity : 0u, (uint32_t)p->P, true, out_color, s);
}

int gsr_event_create(void** event) {
  if (!event) return fail(GSR_E_BADARG, "event is NULL");
  hipEvent_t e = nullptr;
  GSR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  *event = e;
  return 0;
}
int gsr_event_destroy(void* event) {
  if (event) GSR_HIP(hipEventDestroy(static_cast<hipEvent_t>(event)));
  return 0;
}
int gsr_event_wait(void* event) {
  if (!event) return fail(GSR_E_BADARG, "event is NULL");
  GSR_HIP(hipEventSynchronize(static_cast<hipEvent_t>(event)));
  return 0;
}
int gsr_event_query(void* event, int32_t* done) {
  if (!event || !done) return fail(GSR_E_BADARG, "NULL argument");
  const hipError_t e = hipEventQuery(static_cast<hipEvent_t>(event));
  if (e != hipSuccess && e != hipErrorNotReady) return hip_fail(e, "hipEventQuery");
  *done = e == hipSuccess ? 1 : 0;
  return 0;
}

int gsr_enable_markers(int32_t on) {
  std::lock_guard<std::mutex> lock(g_roctx_mu);
  if (!on) {
    g_roctx_push.store(nullptr, std::memory_order_release);
    return 0;
  }
  if (g_roctx_push.load(std::memory_order_acquire)) return 0;
  static const char* const libs[] = {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so",
                                     "libroctx64.so.4"};
  for (const char* name : libs) {
    void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (!h) continue;
    roctx_push_fn push = reinterpret_cast<roctx_push_fn>(dlsym(h, "roctxRangePushA"));
    roctx_pop_fn pop = reinterpret_cast<roctx_pop_fn>(dlsym(h, "roctxRangePop"));
    if (push && pop) {
      g_roctx_pop.store(pop, std::memory_order_release);
      g_roctx_push.store(push, std::memory_order_release);
      return 0;
    }
  }
  return fail(GSR_E_BADARG, "no roctx library found (librocprofiler-sdk-roctx.so / libroctx64.so)");
}

int gsr_backward(const GsrParams* p, const int32_t* radii, const void* geom_ws, const void* bin_ws, const void* img_ws,
                 uint32_t R, uint32_t V, const float* dL_dout_color, void* bwd_ws, size_t bwd_ws_bytes,
                 const GsrGrads* grads, void* stream) {
  if (p && p->forward_only) return fail(GSR_E_BADARG, "the forward ran with forward_only = 1: no state for a backward");
  if (int rc = validate(p)) return rc;
  if (!grads) return fail(GSR_E_BADARG, "grads is NULL");
  if (p->P == 0) return 0;
  if (!radii || !geom_ws || !img_ws || !dL_dout_color || !bwd_ws) return fail(GSR_E_BADARG, "NULL workspace / input");
  if (!grads->dL_dmeans3D || !grads->dL_dmeans2D || !grads->dL_dopacities)
    return fail(GSR_E_BADARG, "dL_dmeans3D / dL_dmeans2D / dL_dopacities must be non-NULL");
  if (p->shs && !grads->dL_dshs) return fail(GSR_E_BADARG, "dL_dshs required with shs");
  if (p->shs_rest && !grads->dL_dshs_rest) return fail(GSR_E_BADARG, "dL_dshs_rest required with shs_rest");
  if (p->shs_rest && ((uintptr_t)grads->dL_dshs_rest & 15u) != 0) return fail(GSR_E_ALIGN, "dL_dshs_rest must be 16-byte aligned");
  if (p->shs && !p->shs_rest && p->M == 16 && ((uintptr_t)grads->dL_dshs & 15u) != 0)
    return fail(GSR_E_ALIGN, "dL_dshs must be 16-byte aligned");
  if (p->colors_precomp && !grads->dL_dcolors) return fail(GSR_E_BADARG, "dL_dcolors required with colors_precomp");
  {
    const int n_stats = (grads->stats_xyz_gradient_accum != nullptr) + (grads->stats_denom != nullptr) +
                        (grads->stats_max_radii2D != nullptr);
    if (n_stats != 0 && n_stats != 3) return fail(GSR_E_BADARG, "the three stats_* pointers must be given together");
  }
  const BwdLayout Wl(p->P, R);
  if (bwd_ws_bytes < Wl.bytes) return fail(GSR_E_CAPACITY, "backward workspace too small");
  if (((uintptr_t)bwd_ws & 255u) != 0) return fail(GSR_E_ALIGN, "bwd_ws must be 256-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const GeomLayout L(p->P);
  const ImageLayout I(p->width, p->height);
  GradRow* rows = at<GradRow>(bwd_ws, Wl.rows);
  uint8_t* flags = at<uint8_t>(bwd_ws, Wl.flags);
  const GeomRec* rec = at<GeomRec>(geom_ws, L.rec);
  if (R > 0) {
    if (!bin_ws) return fail(GSR_E_BADARG, "bin_ws is NULL");
    const SortedViews sv = sorted_views(bin_ws, R, V, p->width, p->height, p->binning_mode);
    const uint32_t* point_list = sv.point_list;
    GSR_HIP(hipMemsetAsync(flags, 0, R, s));
    {
      StageTimer t(p, GSR_STAGE_RENDER_BWD, s);
      launch_render_bwd(p->width, p->height, at<uint2>(img_ws, I.ranges), point_list, rec,
                        at<uint32_t>(geom_ws, L.slot_base), p->bg,
                        at<float>(img_ws, I.final_T), at<uint32_t>(img_ws, I.n_contrib),
                        at<uint32_t>(img_ws, I.tile_max), dL_dout_color, rows, flags,
                        at<uint32_t>(img_ws, I.tile_order), s, sv.inst_mask);
    }
    if (int rc = check(p, s, "render_bwd")) return rc;
  }
  {
    StageTimer t(p, GSR_STAGE_PREPROCESS_BWD, s);
    launch_sum_big_rows(at<uint32_t>(geom_ws, L.total) + TOTAL_BIG, at<uint32_t>(geom_ws, L.block_big_offs), L.nblocks,
                        at<uint32_t>(geom_ws, L.big_list), rec, at<uint32_t>(geom_ws, L.slot_base), rows, flags, s);
    launch_preprocess_bwd(*p, radii, rec, at<uint32_t>(geom_ws, L.slot_base), rows, flags, *grads, s);
  }
  return check(p, s, "preprocess_bwd");
}

int gsr_profile_create(void** handle) {
  if (!handle) return fail(GSR_E_BADARG, "handle is NULL");
  *handle = new Profile();
  return 0;
}
int gsr_profile_destroy(void* handle) {
  Profile* pr = static_cast<Profile*>(handle);
  if (!pr) return 0;
  for (auto& sp : pr->used) { if (sp.a) (void)hipEventDestroy(sp.a); if (sp.b) (void)hipEventDestroy(sp.b); }
  for (auto e : pr->pool) (void)hipEventDestroy(e);
  delete pr;
  return 0;
}
int gsr_profile_collect(void* handle, double* ms_sum, uint32_t* counts) {
  Profile* pr = static_cast<Profile*>(handle);
  if (!pr || !ms_sum || !counts) return fail(GSR_E_BADARG, "NULL argument");
  std::vector<Profile::Span> spans;
  {
    std::lock_guard<std::mutex> lock(pr->mu);
    spans.swap(pr->used);
  }
  hipError_t err = hipSuccess;
  for (auto& sp : spans) {
    if (sp.a && sp.b && err == hipSuccess) {
      err = hipEventSynchronize(sp.b);
      float ms = 0.f;
      if (err == hipSuccess) err = hipEventElapsedTime(&ms, sp.a, sp.b);
      if (err == hipSuccess && sp.stage >= 0 && sp.stage < GSR_STAGE_COUNT) { ms_sum[sp.stage] += ms; counts[sp.stage] += 1; }
    }
    std::lock_guard<std::mutex> lock(pr->mu);     // the events go back to the pool on every path
    if (sp.a) pr->pool.push_back(sp.a);
    if (sp.b) pr->pool.push_back(sp.b);
  }
  if (err != hipSuccess) return hip_fail(err, "gsr_profile_collect");
  return 0;
}
const char* gsr_stage_name(int32_t stage) {
  return (stage >= 0 && stage < GSR_STAGE_COUNT) ? kStageNames[stage] : "?";
}

// Re-runs the forward compositing with work counters (debug / tuning only):
// stats[0] instances in all tile lists, [1] instances staged into LDS, [2] instances visited after the
// sub-block cull, [3] sub-block evaluations, [4] evaluations with >= 1 contributing lane, [5] sum of tile_max.
int gsr_debug_render_stats(const GsrParams* p, const void* geom_ws, const void* bin_ws, void* img_ws, uint32_t R,
                           uint32_t V, float* out_color, unsigned long long* stats /* device [8], zeroed by caller */,
                           void* stream) {
  if (int rc = validate(p)) return rc;
  if (!geom_ws || !bin_ws || !img_ws || !out_color || !stats || R == 0) return fail(GSR_E_BADARG, "NULL argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const ImageLayout I(p->width, p->height);
  const GeomLayout L(p->P);
  const SortedViews sv = sorted_views(bin_ws, R, V, p->width, p->height, p->binning_mode);
  launch_render_fwd(p->width, p->height, at<uint2>(img_ws, I.ranges), sv.point_list,
                    at<GeomRec>(geom_ws, L.rec), p->bg, out_color, at<float>(img_ws, I.final_T),
                    at<uint32_t>(img_ws, I.n_contrib), at<uint32_t>(img_ws, I.tile_max),
                    at<uint32_t>(img_ws, I.tile_order), s, stats,
                    (p->debug_flags & GSR_DEBUG_NO_MINIBLOCK_CULL) ? 0 : 1, sv.inst_mask);
  return check(p, s, "render_stats");
}

int gsr_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, uint8_t* visible, void* stream) {
  if (P < 0) return fail(GSR_E_BADARG, "P < 0");
  if (P == 0) return 0;
  if (!means3D || !viewmatrix || !visible) return fail(GSR_E_BADARG, "NULL argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  launch_mark_visible(P, means3D, viewmatrix, visible, s);
  return check(nullptr, s, "mark_visible");
}

int gsr_sort_pairs_u64(uint64_t* keys, uint32_t* vals, uint64_t* keys_tmp, uint32_t* vals_tmp, uint32_t n,
                       int32_t end_bit, void* scratch, void* stream, int32_t* result_in_tmp) {
  if (!result_in_tmp) return fail(GSR_E_BADARG, "result_in_tmp is NULL");
  *result_in_tmp = 0;
  if (n == 0) return 0;
  if (!keys || !vals || !keys_tmp || !vals_tmp || !scratch) return fail(GSR_E_BADARG, "NULL buffer");
  if (end_bit < 0 || end_bit > 64) return fail(GSR_E_BADARG, "end_bit out of range");
  hipStream_t s = static_cast<hipStream_t>(stream);
  *result_in_tmp = launch_sort_pairs(keys, vals, keys_tmp, vals_tmp, n, end_bit, scratch, s) ? 1 : 0;
  return check(nullptr, s, "sort_pairs");
}

int gsr_debug_read_geom(const void* geom_ws, int32_t P, float* xy, float* conic_opacity, float* rgb, float* depth,
                        uint32_t* tiles_touched, uint32_t* point_offsets, uint32_t* rect, uint32_t* clamped,
                        void* stream) {
  if (!geom_ws || P < 0) return fail(GSR_E_BADARG, "bad geom_ws / P");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const GeomLayout L(P);
  launch_unpack_geom(P, at<GeomRec>(geom_ws, L.rec), at<BinInfo>(geom_ws, L.bin), at<uint32_t>(geom_ws, L.block_offs), xy,
                     conic_opacity, rgb, depth, tiles_touched, point_offsets, rect, clamped, s);
  return check(nullptr, s, "unpack_geom");
}

int gsr_debug_read_binning(const void* geom_ws, int32_t P, const void* bin_ws, uint32_t R, uint32_t V, int32_t width,
                           int32_t height, int32_t mode, uint64_t* keys_sorted, uint32_t* point_list, void* stream) {
  if (R == 0) return 0;
  if (!bin_ws || !geom_ws) return fail(GSR_E_BADARG, "bin_ws / geom_ws is NULL");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const SortedViews v = sorted_views(bin_ws, R, V, width, height, mode);
  if (keys_sorted) {
    if (mode == GSR_BINNING_KEYS64) {
      GSR_HIP(hipMemcpyAsync(keys_sorted, v.keys_sorted, 8 * (size_t)R, hipMemcpyDeviceToDevice, s));
    } else {
      const GeomLayout L(P);
      launch_reconstruct_keys(R, (uint32_t)P, v.tile_sorted, v.point_list, at<BinInfo>(geom_ws, L.bin), keys_sorted, s);
    }
  }
  if (point_list) GSR_HIP(hipMemcpyAsync(point_list, v.point_list, 4 * (size_t)R, hipMemcpyDeviceToDevice, s));
  return check(nullptr, s, "read_binning");
}

int gsr_debug_read_counts(const void* geom_ws, int32_t P, uint32_t out_host[8], void* stream) {
  if (!geom_ws || !out_host || P < 0) return fail(GSR_E_BADARG, "bad geom_ws / out / P");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const GeomLayout L(P);
  GSR_HIP(hipMemcpyAsync(out_host, at<uint32_t>(geom_ws, L.total), 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  GSR_HIP(hipStreamSynchronize(s));
  return 0;
}

int gsr_debug_read_image(const void* img_ws, int32_t width, int32_t height, float* final_T, uint32_t* n_contrib,
                         uint32_t* ranges, void* stream) {
  if (!img_ws) return fail(GSR_E_BADARG, "img_ws is NULL");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const ImageLayout I(width, height);
  const size_t px = (size_t)width * height;
  if (final_T) GSR_HIP(hipMemcpyAsync(final_T, at<float>(img_ws, I.final_T), 4 * px, hipMemcpyDeviceToDevice, s));
  if (n_contrib) GSR_HIP(hipMemcpyAsync(n_contrib, at<uint32_t>(img_ws, I.n_contrib), 4 * px, hipMemcpyDeviceToDevice, s));
  if (ranges) GSR_HIP(hipMemcpyAsync(ranges, at<uint32_t>(img_ws, I.ranges), 8 * (size_t)I.tiles, hipMemcpyDeviceToDevice, s));
  return 0;
}

size_t gsr_l1_loss_workspace_bytes(void) { return l1_loss_workspace_bytes(); }
int gsr_l1_loss_fwd_bwd(const float* x, const float* gt, size_t n, float scale, float* loss_sum, float* dL_dx,
                        void* workspace, void* stream) {
  if (!x || !gt || !loss_sum || !workspace) return fail(GSR_E_BADARG, "NULL input");
  if ((((uintptr_t)x | (uintptr_t)gt | (uintptr_t)dL_dx) & 15u) != 0) return fail(GSR_E_ALIGN, "16-byte alignment required");
  hipStream_t s = static_cast<hipStream_t>(stream);
  launch_l1_loss(x, gt, n, scale, loss_sum, dL_dx, static_cast<float*>(workspace), s);
  return check(nullptr, s, "l1_loss");
}

size_t gsr_l1_dssim_workspace_bytes(int32_t C, int32_t H, int32_t W) {
  if (C <= 0 || H <= 0 || W <= 0) return 0;
  const size_t blocks = (size_t)((W + 15) / 16) * (size_t)((H + 15) / 16) * (size_t)C;      // 16x16 output tiles
  return sizeof(float) * (3 * (size_t)C * (size_t)H * (size_t)W + 2 * blocks);
}
int gsr_l1_dssim_loss_fwd_bwd(const float* x, const float* gt, int32_t C, int32_t H, int32_t W, float lambda_dssim,
                              int32_t dssim_mode, float* sums, float* dL_dx, void* workspace, void* stream) {
  if (!x || !gt || !sums || !dL_dx || !workspace) return fail(GSR_E_BADARG, "NULL argument");
  if (C <= 0 || H <= 0 || W <= 0 || C > 65535) return fail(GSR_E_BADARG, "bad image shape");
  if (dssim_mode != GSR_DSSIM_ONE_MINUS_MEAN && dssim_mode != GSR_DSSIM_CLAMPED_HALF)
    return fail(GSR_E_BADARG, "unknown dssim_mode");
  hipStream_t s = static_cast<hipStream_t>(stream);
  launch_l1_dssim(x, gt, C, H, W, lambda_dssim, dssim_mode, sums, dL_dx, static_cast<float*>(workspace), s);
  return check(nullptr, s, "l1_dssim");
}

size_t gsr_splat2d_workspace_bytes(int32_t N, int32_t H, int32_t W) {
  return (N < 0 || H <= 0 || W <= 0) ? 0 : Splat2dLayout(N, H, W).bytes;
}
static int splat2d_validate(int32_t N, int32_t K, int32_t H, int32_t W, const void* ws, size_t ws_bytes) {
  if (N < 0 || H <= 0 || W <= 0 || K <= 0) return fail(GSR_E_BADARG, "bad N / K / image size");
  if (K > H || K > W) return fail(GSR_E_BADARG, "Kernel size should be smaller or equal to the image size.");
  if (K > splat2d_max_kernel_size()) return fail(GSR_E_BADARG, "kernel size above 2048");
  if (!ws) return fail(GSR_E_BADARG, "NULL workspace");
  if (((uintptr_t)ws & 255u) != 0) return fail(GSR_E_ALIGN, "workspace must be 256-byte aligned");
  if (ws_bytes < Splat2dLayout(N, H, W).bytes) return fail(GSR_E_CAPACITY, "splat2d workspace too small");
  return 0;
}
int gsr_splat2d_forward(int32_t N, int32_t K, int32_t H, int32_t W, const float* sigma_x, const float* sigma_y,
                        const float* rho, const float* coords, const float* colours, const float* ax, void* workspace,
                        size_t workspace_bytes, float* out, int32_t* not_pd_host, void* stream) {
  if (int rc = splat2d_validate(N, K, H, W, workspace, workspace_bytes)) return rc;
  if (!out || !ax) return fail(GSR_E_BADARG, "NULL argument");
  if (N > 0 && (!sigma_x || !sigma_y || !rho || !coords || !colours)) return fail(GSR_E_BADARG, "NULL input");
  hipStream_t s = static_cast<hipStream_t>(stream);
  launch_splat2d_fwd(N, K, H, W, sigma_x, sigma_y, rho, coords, colours, ax, workspace, out, s);
  if (int rc = check(nullptr, s, "splat2d_forward")) return rc;
  if (not_pd_host) {
    int flag = 0;
    GSR_HIP(hipMemcpyAsync(&flag, static_cast<char*>(workspace) + Splat2dLayout(N, H, W).flag, 4, hipMemcpyDeviceToHost, s));
    GSR_HIP(hipStreamSynchronize(s));
    *not_pd_host = flag;
  }
  return 0;
}
int gsr_splat2d_backward(int32_t N, int32_t K, int32_t H, int32_t W, const float* sigma_x, const float* sigma_y,
                         const float* rho, const float* ax, void* workspace, size_t workspace_bytes,
                         const float* dL_dout, float* dL_dsigma_x, float* dL_dsigma_y, float* dL_drho,
                         float* dL_dcoords, float* dL_dcolours, void* stream) {
  if (int rc = splat2d_validate(N, K, H, W, workspace, workspace_bytes)) return rc;
  if (!dL_dout || !ax) return fail(GSR_E_BADARG, "NULL argument");
  if (N > 0 && (!sigma_x || !sigma_y || !rho || !dL_dsigma_x || !dL_dsigma_y || !dL_drho || !dL_dcoords || !dL_dcolours))
    return fail(GSR_E_BADARG, "NULL input / gradient");
  hipStream_t s = static_cast<hipStream_t>(stream);
  launch_splat2d_bwd(N, K, H, W, sigma_x, sigma_y, rho, ax, workspace, dL_dout, dL_dsigma_x, dL_dsigma_y, dL_drho,
                     dL_dcoords, dL_dcolours, s);
  return check(nullptr, s, "splat2d_backward");
}

size_t gsr_knn3_workspace_bytes(int32_t N) { return knn_workspace_bytes(N); }
int gsr_dist2_knn3(const float* points, int32_t N, float* mean_dist2, void* workspace, size_t workspace_bytes,
                   void* stream) {
  if (N < 0) return fail(GSR_E_BADARG, "N < 0");
  if (N == 0) return 0;
  if (!points || !mean_dist2 || !workspace) return fail(GSR_E_BADARG, "NULL argument");
  if (workspace_bytes < knn_workspace_bytes(N)) return fail(GSR_E_CAPACITY, "knn workspace too small");
  if (((uintptr_t)workspace & 255u) != 0) return fail(GSR_E_ALIGN, "workspace must be 256-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  launch_knn3(points, N, mean_dist2, workspace, s);
  return check(nullptr, s, "knn3");
}

int gsr_densify_stats(int32_t P, const float* dL_dmeans2D, const int32_t* radii, float* xyz_gradient_accum, float* denom,
                      float* max_radii2D, void* stream) {
  if (P < 0) return fail(GSR_E_BADARG, "P < 0");
  if (P == 0) return 0;
  if (!dL_dmeans2D || !radii || !xyz_gradient_accum || !denom || !max_radii2D) return fail(GSR_E_BADARG, "NULL input");
  hipStream_t s = static_cast<hipStream_t>(stream);
  launch_densify_stats(P, dL_dmeans2D, radii, xyz_gradient_accum, denom, max_radii2D, s);
  return check(nullptr, s, "densify_stats");
}

size_t gsr_densify_workspace_bytes(int32_t P) { return P < 0 ? 0 : DensifyLayout(P).bytes; }
int gsr_densify_plan(int32_t P, const float* xyz_gradient_accum, const float* denom, const float* scaling_raw,
                     const float* opacity_raw, float grad_threshold, float percent_dense_extent, float min_opacity,
                     float max_world_scale, void* workspace, size_t workspace_bytes, uint32_t counts_host[4],
                     void* stream) {
  if (P < 0) return fail(GSR_E_BADARG, "P < 0");
  if (!counts_host) return fail(GSR_E_BADARG, "NULL counts_host");
  counts_host[0] = counts_host[1] = counts_host[2] = counts_host[3] = 0;
  if (P == 0) return 0;
  if (!xyz_gradient_accum || !denom || !scaling_raw || !opacity_raw || !workspace) return fail(GSR_E_BADARG, "NULL input");
  if (((uintptr_t)workspace & 255u) != 0) return fail(GSR_E_ALIGN, "workspace must be 256-byte aligned");
  const DensifyLayout L(P);
  if (workspace_bytes < L.bytes) return fail(GSR_E_CAPACITY, "densify workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  launch_densify_plan(P, xyz_gradient_accum, denom, scaling_raw, opacity_raw, grad_threshold, percent_dense_extent,
                      min_opacity, max_world_scale, max_world_scale >= 0.0f ? 1 : 0, workspace, s);
  if (int rc = check(nullptr, s, "densify_plan")) return rc;
  GSR_HIP(hipMemcpyAsync(counts_host, static_cast<char*>(workspace) + L.totals, 16, hipMemcpyDeviceToHost, s));
  GSR_HIP(hipStreamSynchronize(s));
  return 0;
}
int gsr_densify_gather_rows(int32_t P, int32_t row_floats, const float* src, const void* workspace,
                            const uint32_t counts[4], int32_t zero_new, float* dst, void* stream) {
  if (P < 0 || row_floats <= 0) return fail(GSR_E_BADARG, "bad P / row_floats");
  if (P == 0) return 0;
  if (!src || !workspace || !counts) return fail(GSR_E_BADARG, "NULL argument");
  if (!dst && counts[0] + counts[1] + counts[2] > 0) return fail(GSR_E_BADARG, "NULL dst");
  hipStream_t s = static_cast<hipStream_t>(stream);
  launch_densify_gather_rows(P, row_floats, src, workspace, counts, zero_new, dst, s);
  return check(nullptr, s, "densify_gather_rows");
}
int gsr_densify_split_children(int32_t P, const float* xyz, const float* scaling_raw, const float* rotation_raw,
                               const float* noise, const void* workspace, const uint32_t counts[4], float* dst_xyz,
                               float* dst_scaling, void* stream) {
  if (P < 0) return fail(GSR_E_BADARG, "P < 0");
  if (!counts) return fail(GSR_E_BADARG, "NULL counts");
  if (P == 0 || counts[2] == 0) return 0;
  if (!xyz || !scaling_raw || !rotation_raw || !noise || !workspace || !dst_xyz || !dst_scaling)
    return fail(GSR_E_BADARG, "NULL argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  launch_densify_split_children(P, xyz, scaling_raw, rotation_raw, noise, workspace, counts, dst_xyz, dst_scaling, s);
  return check(nullptr, s, "densify_split_children");
}

}  // extern "C"
