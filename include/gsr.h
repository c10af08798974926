/*
 * gsr.h — C ABI of the MI355X-native differentiable Gaussian rasterizer (libgsr_hip.so).
 *
 * This is the drop-in boundary for the reference's `diff_gaussian_rasterization._C` pybind
 * module (absent from /root/reference: un-vendored submodule, .gitmodules:4-6).  The
 * reference binds it at
 *     gaussian_renderer/__init__.py:15      (import of GaussianRasterizationSettings/GaussianRasterizer)
 *     gaussian_renderer/__init__.py:42-57   (settings tuple + module construction, every frame)
 *     gaussian_renderer/__init__.py:257-265 (the call)
 * and reaches the backward through loss.backward() at train.py:107.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer marked "device" is HIP device memory owned
 *     by the caller (PyTorch on the Python side); the library never allocates or frees device
 *     memory and keeps no global mutable state (forward and backward arrive on different OS
 *     threads: train.py:107 runs backward on an autograd engine thread);
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it; the only host
 *     synchronisation is the read-back of (num_rendered, num_visible) in gsr_forward_preprocess();
 *   - return value 0 = success, >0 = hipError_t, <0 = library error (GSR_E_*); the message is
 *     available per thread from gsr_last_error();
 *   - all float tensors are contiguous fp32; matrices are the row-vector-convention 4x4s the
 *     reference builds at scene/cameras.py:54-57 (p_view = [x,y,z,1] @ viewmatrix).
 */
#ifndef GSR_H_
#define GSR_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSR_ABI_VERSION 13

enum {
  GSR_OK = 0,
  GSR_E_BADARG = -1,    /* null / inconsistent arguments (both-or-neither of shs/colors, scales+rotations/cov3D) */
  GSR_E_CAPACITY = -2,  /* binning workspace smaller than gsr_binning_bytes(num_rendered) */
  GSR_E_ALIGN = -3      /* a pointer violates the documented alignment */
};

/* One call's inputs.  Replaces the positional arguments of the reference-side
 * `_C.rasterize_gaussians(bg, means3D, colors, opacity, scales, rotations, scale_modifier,
 * cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, H, W, sh, degree, campos,
 * prefiltered, debug)` implied by gaussian_renderer/__init__.py:42-55,257-265. */
typedef struct GsrParams {
  int32_t P;              /* number of Gaussians */
  int32_t M;              /* SH coefficients stored per Gaussian in `shs` ((max_sh_degree+1)^2); 0 with colors_precomp */
  int32_t D;              /* active SH degree evaluated (0..3), settings.sh_degree */
  int32_t width, height;  /* settings.image_width / image_height */
  float tan_fovx, tan_fovy;
  float scale_modifier;
  int32_t prefiltered;    /* accepted and ignored: the reference always passes False (gaussian_renderer/__init__.py:53);
                             culled Gaussians are simply skipped whatever its value */
  int32_t debug;          /* 1: synchronise and check after every kernel */
  const float* means3D;        /* device [P,3] */
  const float* shs;            /* device [P,M,3] or NULL (16-byte aligned) */
  const float* colors_precomp; /* device [P,3] or NULL */
  const float* opacities;      /* device [P] (the reference passes [P,1]) */
  const float* scales;         /* device [P,3] or NULL */
  const float* rotations;      /* device [P,4] (w,x,y,z) or NULL (16-byte aligned) */
  const float* cov3D_precomp;  /* device [P,6] (xx,xy,xz,yy,yz,zz) or NULL */
  const float* viewmatrix;     /* device [16] */
  const float* projmatrix;     /* device [16] */
  const float* campos;         /* device [3] */
  const float* bg;             /* device [3] */
  void* profile;               /* NULL, or a handle from gsr_profile_create(): HIP-event stage timers */
  /* Fused-input extension (removes the caller's torch.cat / exp / normalize / sigmoid of
   * scene/gaussian_model.py:151-183 and their autograd): */
  const float* shs_rest;       /* NULL, or device [P,M-1,3] (16-byte aligned): then `shs` is [P,1,3] (f_dc) */
  int32_t act_flags;           /* GSR_ACT_*: inputs are RAW parameters, the activation (and its gradient) is applied here */
  int32_t binning_mode;        /* GSR_BINNING_*; same value in every call of a frame */
  uint32_t* counts_pinned;     /* NULL, or HOST-PINNED, device-accessible uint32[4] (hipHostMalloc / torch pin_memory):
                                  the scan kernel stores (num_rendered, num_visible, smallest and largest depth key) there and gsr_forward_preprocess
                                  waits on an event behind that kernel only, so the depth sort it has already
                                  enqueued keeps the GPU busy while the host sizes and launches stage 2 */
  int32_t forward_only;        /* 1: no gsr_backward will follow (inference): the compositing kernel does not track the
                                  last contributor and the per-pixel / per-Gaussian state the backward reads (final
                                  transmittance, contributor counts, gradient-row slots) is not written.  Same image. */
  int32_t debug_flags;         /* GSR_DEBUG_* bits: switches the tests use (never read from the environment) */
  uint8_t* visible_out;        /* NULL, or device [P] (ABI v12): the forward also stores radii[i] > 0 there -- the
                                  `visibility_filter = radii > 0` of the reference's render()
                                  (gaussian_renderer/__init__.py:311) without a pass of its own over the radii */
  int32_t depth_span_lt24;     /* gsr_forward only (ABI v13).  1: the caller expects the frame's depth keys (float32 bits of the
                                  view depths of the visible Gaussians) to span fewer than 2^24 steps -- counts_pinned[3] -
                                  counts_pinned[2] < 2^24, about two binades of depth -- and the depth sort's fourth pass (three
                                  launches that find nothing to do on such a frame) is NOT enqueued.  The caller must check
                                  the two words once the counts event has completed, exactly as it checks the capacity: a
                                  frame that spans more is sorted on its low 24 key bits only (no out-of-bounds access,
                                  but lists in the wrong depth order) and must be discarded / redone with 0.
                                  0: the pass is enqueued and decides on the device (any frame is right). */
} GsrParams;

enum {
  GSR_DEBUG_NO_MINIBLOCK_CULL = 1  /* forward compositing: every staged instance enters all 16 mini-block lists (the
                                      image must not change by a bit: tests/test_gpu_miniblock_cull.py) */
};

enum {
  GSR_BINNING_TWO_LEVEL = 0,   /* depth-sort the visible Gaussians (u32 keys), emit instances in depth order, stable
                                  partition by tile id (u32 keys): same lists as KEYS64 for ~2.5x less sort traffic */
  GSR_BINNING_KEYS64 = 1,      /* upstream layout: duplicateWithKeys + radix sort of u64 tile<<32|depth keys */
  GSR_BINNING_TWO_LEVEL_CULLED = 2 /* TWO_LEVEL minus the (Gaussian, tile) instances whose tile the alpha >= 1/255
                                  ellipse provably cannot reach (rects of > 32 tiles shrink to the ellipse's bounding box;
                                  rects of <= 32 tiles get an exact per-tile ellipse test with a safety margin): such
                                  instances are rejected pixel by pixel by the alpha test anyway,
                                  so colour, radii and every gradient are bit-identical to the other modes; only the
                                  internal lists (and n_contrib positions) are shorter */
};

enum {
  GSR_ACT_SCALE_EXP = 1,       /* scales    = exp(raw)              scene/gaussian_model.py:34,151-153 */
  GSR_ACT_ROT_NORMALIZE = 2,   /* rotations = raw / max(|raw|,1e-12) scene/gaussian_model.py:42,167-169 */
  GSR_ACT_OPACITY_SIGMOID = 4  /* opacity   = sigmoid(raw)          scene/gaussian_model.py:39,181-183 */
};

/* Gradient outputs of the backward.  Replaces the tuple returned by the reference-side
 * `_C.rasterize_gaussians_backward(...)`.  Every buffer is written in full by the call
 * (no caller zero-fill required); NULL is allowed for the members that do not apply
 * (dL_dshs without shs, dL_dcolors without colors_precomp, dL_dscales/dL_drotations with
 * cov3D_precomp, dL_dcov3D without it). */
typedef struct GsrGrads {
  float* dL_dmeans3D;   /* device [P,3] */
  float* dL_dmeans2D;   /* device [P,3] (x,y in NDC units scaled by 0.5*W / 0.5*H; z = 0) */
  float* dL_dshs;       /* device [P,M,3] */
  float* dL_dcolors;    /* device [P,3] */
  float* dL_dopacities; /* device [P] */
  float* dL_dscales;    /* device [P,3] */
  float* dL_drotations; /* device [P,4] */
  float* dL_dcov3D;     /* device [P,6] */
  float* dL_dshs_rest;  /* device [P,M-1,3]; required with shs_rest (dL_dshs is then [P,1,3]) */
  /* Fused densification statistics (scene/gaussian_model.py:775-777 + train.py:130): all three
   * NULL, or all three set -- then the per-Gaussian backward kernel, which holds dL_dmeans2D and the radius in
   * registers, also does  xyz_gradient_accum[i] += ||dL_dmeans2D[i].xy||, denom[i] += 1,
   * max_radii2D[i] = max(max_radii2D[i], radii[i])  for every Gaussian with radii[i] > 0 (same arithmetic as
   * gsr_densify_stats, which stays available as the stand-alone step). */
  float* stats_xyz_gradient_accum; /* device [P] (the reference keeps [P,1]) */
  float* stats_denom;              /* device [P] */
  float* stats_max_radii2D;        /* device [P] */
} GsrGrads;

/* ---- introspection ------------------------------------------------------------------ */
int gsr_abi_version(void);
const char* gsr_last_error(void);       /* thread-local, never NULL */
const char* gsr_build_info(void);       /* "gfx950 ..." */

/* ---- workspace sizing (bytes; all workspaces must be 256-byte aligned) ----------------- */
size_t gsr_geom_bytes(int32_t P);                        /* per-Gaussian state (upstream "geomBuffer") */
size_t gsr_image_bytes(int32_t width, int32_t height);   /* per-pixel + per-tile state ("imgBuffer") */
size_t gsr_binning_bytes(uint32_t num_rendered, uint32_t num_visible, int32_t width, int32_t height,
                         int32_t binning_mode);          /* keys/values/sort scratch ("binningBuffer") */
size_t gsr_backward_bytes(int32_t P, uint32_t num_rendered); /* per-instance gradient rows + flags */

/* ---- forward -------------------------------------------------------------------------- */
/* Stage 1: preprocess (cull, project, cov3D->cov2D->conic, radius, tile rect, SH->RGB) and the
 * prefix sum of tiles_touched.  Writes radii[P] (int32), *num_rendered (instances) and *num_visible (host).
 * Blocks the calling thread until the two counts have been read back (the one sync of the forward). */
int gsr_forward_preprocess(const GsrParams* p, void* geom_ws, int32_t* radii, void* stream,
                           uint32_t* num_rendered, uint32_t* num_visible);

/* Stage 2: tile binning (see binning_mode), identifyTileRanges, per-tile compositing.
 * Writes out_color[3,H,W].  bin_ws must hold gsr_binning_bytes(num_rendered, num_visible, W, H, mode). */
int gsr_forward_render(const GsrParams* p, void* geom_ws, void* bin_ws, size_t bin_ws_bytes,
                       void* img_ws, uint32_t num_rendered, uint32_t num_visible, float* out_color, void* stream);

/* Both stages in one call with NO host synchronisation (SURVEY §7 "hard parts": caller-provided capacity + overflow
 * flag instead of upstream's per-forward num_rendered read-back, gaussian_renderer/__init__.py:257-265).
 *   capacity     : instances the binning workspace can hold; bin_ws must hold
 *                  gsr_binning_bytes(capacity, P, W, H, mode) bytes.  Every launch is sized for (capacity, P) and the
 *                  kernels take the real counts from device memory.
 *   p->counts_pinned (required): receives (num_rendered, num_visible, min depth key, max depth key) when the scan
 *                  kernel has run; `counts_event` (a handle from gsr_event_create, or NULL) is recorded right behind it.
 *   Overflow     : when num_rendered > capacity the instances past the capacity are dropped (no out-of-bounds
 *                  access, but the image and every gradient of the frame are INCOMPLETE): the caller must compare
 *                  counts_pinned[0] with its capacity once the event has completed and discard / redo the frame.
 *   Depth span   : with p->depth_span_lt24 = 1 the same check covers counts_pinned[3] - counts_pinned[2] < 2^24 (see the field).
 *   Backward     : call gsr_backward with num_rendered = capacity and num_visible = P (the values the workspaces
 *                  were laid out with); gsr_backward_bytes(P, capacity) sizes its workspace.
 * Only GSR_BINNING_TWO_LEVEL / _CULLED (the 64-bit key mode keeps the two-call path). */
int gsr_forward(const GsrParams* p, void* geom_ws, void* bin_ws, size_t bin_ws_bytes, uint32_t capacity, void* img_ws,
                int32_t* radii, float* out_color, void* counts_event, void* stream);
/* events for the deferred count check (plain HIP events without timing; usable from any thread) */
int gsr_event_create(void** event);
int gsr_event_destroy(void* event);
int gsr_event_wait(void* event);                 /* blocks the calling thread until the event has completed */
int gsr_event_query(void* event, int32_t* done); /* *done = 1 when completed; never blocks */

/* ---- backward -------------------------------------------------------------------------
 * geom_ws, bin_ws and img_ws as the forward of the SAME frame left them: besides the sorted point list, bin_ws holds the
 * 16-bit mini-block reach mask of every instance, which the forward's compositing stores in list order (in the tile
 * sort's spare payload buffer) when p->forward_only is 0 and the backward's compositing reads instead of re-deriving. */
int gsr_backward(const GsrParams* p, const int32_t* radii, const void* geom_ws, const void* bin_ws,
                 const void* img_ws, uint32_t num_rendered, uint32_t num_visible,
                 const float* dL_dout_color /* [3,H,W] */,
                 void* bwd_ws, size_t bwd_ws_bytes, const GsrGrads* grads, void* stream);

/* `_C.mark_visible(means3D, viewmatrix, projmatrix)` of the upstream module (unused by the reference): visible[i] = 1
 * when Gaussian i passes the near-plane test of the preprocess stage (view z > 0.2). */
int gsr_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, uint8_t* visible, void* stream);

/* ---- unit entry points (each stage callable on its own; used by the parity tests) ------- */
/* keys_out/vals_out receive the result; *_tmp are scratch of the same size; bits sorted: [0,end_bit) */
size_t gsr_sort_scratch_bytes(uint32_t n);
int gsr_sort_pairs_u64(uint64_t* keys, uint32_t* vals, uint64_t* keys_tmp, uint32_t* vals_tmp,
                       uint32_t n, int32_t end_bit, void* scratch, void* stream,
                       int32_t* result_in_tmp /* host out: 1 if the sorted data ended in *_tmp */);
/* copies internal state out for inspection (any pointer may be NULL) */
int gsr_debug_read_geom(const void* geom_ws, int32_t P, float* xy /*[P,2]*/, float* conic_opacity /*[P,4]*/,
                        float* rgb /*[P,3]*/, float* depth /*[P]*/, uint32_t* tiles_touched /*[P]*/,
                        uint32_t* point_offsets /*[P]*/, uint32_t* rect /*[P,4] x0,y0,x1,y1*/,
                        uint32_t* clamped /*[P]*/, void* stream);
/* keys_sorted are the (tile<<32|depth) keys of the sorted instances (rebuilt from the result in two-level mode).
 * (num_rendered, num_visible) = what the binning workspace was laid out for: after gsr_forward that is (capacity, P),
 * and only the first real-count entries of the outputs are meaningful. */
int gsr_debug_read_binning(const void* geom_ws, int32_t P, const void* bin_ws, uint32_t num_rendered,
                           uint32_t num_visible, int32_t width, int32_t height, int32_t binning_mode,
                           uint64_t* keys_sorted, uint32_t* point_list, void* stream);
/* the device-side counters of a frame: out[0] num_rendered, [1] num_visible, [2] Gaussians with more than 64 instances
 * (their gradient rows are pre-summed cooperatively), [3] reserved, [4] ~(smallest depth key), [5] largest depth key,
 * [6] element count of the depth sort's top-digit pass (0: the frame's depths fit 24 key bits), [7] instances the binning
 * workspace received (min(num_rendered, capacity)).  Synchronises the stream. */
int gsr_debug_read_counts(const void* geom_ws, int32_t P, uint32_t out_host[8], void* stream);
int gsr_debug_read_image(const void* img_ws, int32_t width, int32_t height, float* final_T,
                         uint32_t* n_contrib, uint32_t* ranges /*[T,2]*/, void* stream);

/* forward compositing re-run with work counters; stats = device u64[8], zeroed by the caller:
 * [0] instances in all tile lists, [1] staged into LDS, [2] visited after the sub-block cull,
 * [3] sub-block evaluations, [4] evaluations with at least one contributing lane, [5] sum of per-tile last contributor */
int gsr_debug_render_stats(const GsrParams* p, const void* geom_ws, const void* bin_ws, void* img_ws,
                           uint32_t num_rendered, uint32_t num_visible, float* out_color, unsigned long long* stats,
                           void* stream);

/* ---- stage timers (opt-in; HIP events recorded on the call's stream around each stage) ------ */
enum {
  GSR_STAGE_PREPROCESS_FWD = 0, GSR_STAGE_SCAN, GSR_STAGE_DUPLICATE, GSR_STAGE_SORT, GSR_STAGE_RANGES,
  GSR_STAGE_RENDER_FWD, GSR_STAGE_RENDER_BWD, GSR_STAGE_PREPROCESS_BWD, GSR_STAGE_COUNT
};
int gsr_profile_create(void** handle);
int gsr_profile_destroy(void* handle);
/* waits for the recorded events, adds their elapsed times into ms_sum[GSR_STAGE_COUNT] and the number of
 * recorded intervals into counts[GSR_STAGE_COUNT], then clears the handle for reuse */
int gsr_profile_collect(void* handle, double* ms_sum, uint32_t* counts);
const char* gsr_stage_name(int32_t stage);
/* roctx ranges ("gsr:<stage>") around the same stages, for rocprofv3 --marker-trace.  Off by default;
 * on = 1 loads librocprofiler-sdk-roctx.so (or libroctx64.so) at run time and returns GSR_E_BADARG if neither is
 * there.  Process-wide switch, safe to call from any thread. */
int gsr_enable_markers(int32_t on);

/* ---- caller-side steps of the train loop ---------------------------- */
/* L1 loss (utils/loss_utils.py:17-18) forward + gradient in one pass:
 * loss_sum[0] = sum|x-gt| (written, not accumulated; the caller divides by n), dL_dx = sign(x-gt) * scale.
 * `workspace`: device scratch of gsr_l1_loss_workspace_bytes() for the per-block partial sums, which are added in a
 * fixed order (no atomics): the loss is bitwise reproducible from run to run. */
size_t gsr_l1_loss_workspace_bytes(void);
int gsr_l1_loss_fwd_bwd(const float* x, const float* gt, size_t n, float scale, float* loss_sum,
                        float* dL_dx, void* workspace, void* stream);
/* The reference's training loss (train.py:99-101) in two kernels: (1-lambda)*L1 + lambda*(1 - SSIM) with SSIM as
 * utils/loss_utils.py:23-63 (11x11 Gaussian window, sigma 1.5, zero padding, mean over C*H*W).
 * dssim_mode GSR_DSSIM_ONE_MINUS_MEAN: sums[0] += sum|x-gt|, sums[1] += sum SSIM
 *   (caller zero-fills; loss = (1-lambda)*sums[0]/n + lambda*(1 - sums[1]/n));
 * dssim_mode GSR_DSSIM_CLAMPED_HALF (the 2D script's combined_loss, 2d_gaussian_splatting.py:196-202):
 *   sums[1] += sum clamp((1-SSIM)/2, 0, 1); loss = (1-lambda)*sums[0]/n + lambda*sums[1]/n.
 * dL_dx receives the full gradient of that loss; workspace: gsr_l1_dssim_workspace_bytes (three derivative maps +
 * one pair of partial sums per 16x16 tile; the sums are formed in a fixed order, so the value is reproducible). */
#define GSR_DSSIM_ONE_MINUS_MEAN 0
#define GSR_DSSIM_CLAMPED_HALF 1
size_t gsr_l1_dssim_workspace_bytes(int32_t C, int32_t H, int32_t W);
int gsr_l1_dssim_loss_fwd_bwd(const float* x, const float* gt, int32_t C, int32_t H, int32_t W, float lambda_dssim,
                              int32_t dssim_mode, float* sums, float* dL_dx, void* workspace, void* stream);

/* BASELINE config 1: `generate_2D_gaussian_splatting(kernel_size, sigma_x, sigma_y, rho, coords, colours, image_size)`
 * (2D-Gaussian-Splatting-main/2d_gaussian_splatting.py:44-123) without the N x 3 x H x W intermediate.
 *   sigma_x, sigma_y, rho [N]; coords [N,2] (normalised translation, x then y); colours [N,3];
 *   ax [K]: the kernel-grid abscissae `-5 + 10 * linspace(0, 1, K)` as the host framework evaluates them (:66-71);
 *   out [3,H,W] channel-major (the reference returns `.permute(1, 2, 0)` of exactly that buffer, :120-121).
 * The workspace (gsr_splat2d_workspace_bytes, 256-byte aligned) carries the forward state to gsr_splat2d_backward.
 * not_pd_host, when non-NULL, makes the call synchronise the stream and receive 1 if any covariance has det <= 0
 * (the reference raises ValueError there, :59-61).  Returns GSR_E_BADARG if K > min(H, W) (:93-94) or K > 2048. */
size_t gsr_splat2d_workspace_bytes(int32_t N, int32_t H, int32_t W);
int gsr_splat2d_forward(int32_t N, int32_t K, int32_t H, int32_t W, const float* sigma_x, const float* sigma_y,
                        const float* rho, const float* coords, const float* colours, const float* ax, void* workspace,
                        size_t workspace_bytes, float* out, int32_t* not_pd_host, void* stream);
/* dL_dout [3,H,W] -> gradients of all five inputs (overwritten, not accumulated; deterministic). */
int gsr_splat2d_backward(int32_t N, int32_t K, int32_t H, int32_t W, const float* sigma_x, const float* sigma_y,
                         const float* rho, const float* ax, void* workspace, size_t workspace_bytes,
                         const float* dL_dout, float* dL_dsigma_x, float* dL_dsigma_y, float* dL_drho,
                         float* dL_dcoords, float* dL_dcolours, void* stream);
/* Replacement for `simple_knn._C.distCUDA2(points[N,3]) -> meanDist2[N]` (scene/gaussian_model.py:21,210; the
 * submodule is absent from the reference): mean of the squared distances to the 3 nearest OTHER points, exact. */
size_t gsr_knn3_workspace_bytes(int32_t N);
int gsr_dist2_knn3(const float* points, int32_t N, float* mean_dist2, void* workspace, size_t workspace_bytes,
                   void* stream);
/* scene/gaussian_model.py:775-777 + train.py:130 fused: for radii>0:
 * xyz_gradient_accum += ||dL_dmeans2D.xy||, denom += 1, max_radii2D = max(max_radii2D, radii) */
int gsr_densify_stats(int32_t P, const float* dL_dmeans2D /*[P,3]*/, const int32_t* radii,
                      float* xyz_gradient_accum, float* denom, float* max_radii2D, void* stream);

/* scene/gaussian_model.py:750-772 `densify_and_prune` (plain branch: clone :580-610, split :506-578 with N = 2,
 * postfix :466-504, prune :401-449) as one plan and one read-once / write-once pass per tensor.
 *   gsr_densify_plan: classifies every Gaussian from xyz_gradient_accum / denom (NaN -> 0), the RAW scaling [P,3] and
 *     RAW opacity [P]; percent_dense_extent = percent_dense * scene_extent; max_world_scale = 0.1 * extent, or < 0
 *     when the caller's max_screen_size is None / 0 (then only the opacity test prunes, :759-764).  Synchronises the
 *     stream and returns counts_host = {kept originals, kept clones, kept children PER COPY, split-selected}.
 *     The output has counts[0] + counts[1] + 2*counts[2] rows in the reference's order:
 *     [kept originals | clones | first children | second children].
 *   gsr_densify_gather_rows: dst[rows_out, row_floats] <- src[P, row_floats]; zero_new = 1 stores zeros in the appended
 *     rows (Adam exp_avg / exp_avg_sq of new points, :458-459).
 *   gsr_densify_split_children: overwrites the children's rows of dst_xyz / dst_scaling with
 *     R(rotation) (exp(scaling) * noise) + xyz and log(exp(scaling) / 1.6); noise [2*counts[3], 3] holds the standard-
 *     normal draws of `torch.normal(mean=0, std=stds)` (:537-539) in the reference's order (all first samples of the
 *     selected Gaussians, then all second samples). */
size_t gsr_densify_workspace_bytes(int32_t P);
int gsr_densify_plan(int32_t P, const float* xyz_gradient_accum, const float* denom, const float* scaling_raw,
                     const float* opacity_raw, float grad_threshold, float percent_dense_extent, float min_opacity,
                     float max_world_scale, void* workspace, size_t workspace_bytes, uint32_t counts_host[4],
                     void* stream);
int gsr_densify_gather_rows(int32_t P, int32_t row_floats, const float* src, const void* workspace,
                            const uint32_t counts[4], int32_t zero_new, float* dst, void* stream);
int gsr_densify_split_children(int32_t P, const float* xyz, const float* scaling_raw, const float* rotation_raw,
                               const float* noise, const void* workspace, const uint32_t counts[4], float* dst_xyz,
                               float* dst_scaling, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GSR_H_ */
