// Caller-side steps of the train loop that sit directly on either side of the rasterizer:
//   L1 loss + its gradient          utils/loss_utils.py:17-18, train.py:99      
//   densification statistics        scene/gaussian_model.py:775-777, train.py:130 (§8 a13)
// plus the unpack kernels behind the gsr_debug_read_* inspection entry points.
#include "gsr_common.h"
#include "gsr_launch.h"

namespace gsr {

// partials[block] = sum over the block's elements of |x - gt| ; dL_dx = sign(x - gt) * scale.  Streaming, 16 B per lane.
// No atomics: the block partials are summed in block order by l1_loss_finish_kernel, so the loss is bitwise
// reproducible from run to run (like loss.hip's ordered sums and the atomic-free backward).
constexpr int L1_THREADS = 1024;
constexpr int L1_MAX_BLOCKS = 512;     // two 16-wave blocks per CU
__global__ __launch_bounds__(L1_THREADS) void l1_loss_kernel(const float* __restrict__ x, const float* __restrict__ gt,
                                                             size_t n, float scale, float* __restrict__ partials,
                                                             float* __restrict__ dL_dx) {
  __shared__ float wsum[L1_THREADS / WAVE];
  const size_t n4 = n / 4;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  float acc = 0.0f;
  auto sgn = [](float d) { return d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f); };
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 a = reinterpret_cast<const float4*>(x)[i];
    const float4 b = reinterpret_cast<const float4*>(gt)[i];
    const float d0 = a.x - b.x, d1 = a.y - b.y, d2 = a.z - b.z, d3 = a.w - b.w;
    acc += fabsf(d0) + fabsf(d1) + fabsf(d2) + fabsf(d3);
    if (dL_dx) reinterpret_cast<float4*>(dL_dx)[i] = make_float4(sgn(d0) * scale, sgn(d1) * scale, sgn(d2) * scale, sgn(d3) * scale);
  }
  for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float d = x[i] - gt[i];
    acc += fabsf(d);
    if (dL_dx) dL_dx[i] = sgn(d) * scale;
  }
  acc = wave_reduce_add_f32(acc);
  if ((threadIdx.x & (WAVE - 1)) == 0) wsum[threadIdx.x / WAVE] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.0f;
#pragma unroll
    for (int w = 0; w < L1_THREADS / WAVE; ++w) s += wsum[w];
    partials[blockIdx.x] = s;
  }
}
// one wave: lane l sums partials l, l+64, ... in order, then a fixed butterfly
__global__ __launch_bounds__(WAVE) void l1_loss_finish_kernel(const float* __restrict__ partials, int blocks,
                                                              float* __restrict__ loss_sum) {
  float acc = 0.0f;
  for (int i = threadIdx.x; i < blocks; i += WAVE) acc += partials[i];
  acc = wave_reduce_add_f32(acc);
  if (threadIdx.x == 0) loss_sum[0] = acc;
}

__global__ __launch_bounds__(256) void densify_stats_kernel(int P, const float* __restrict__ dL_dmeans2D,
                                                            const int32_t* __restrict__ radii,
                                                            float* __restrict__ accum, float* __restrict__ denom,
                                                            float* __restrict__ max_radii2D) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const int32_t r = radii[i];
  if (r > 0) {
    const float gx = dL_dmeans2D[3 * (size_t)i], gy = dL_dmeans2D[3 * (size_t)i + 1];
    accum[i] += densify_grad_norm(gx, gy);
    denom[i] += 1.0f;
    max_radii2D[i] = fmaxf(max_radii2D[i], (float)r);
  }
}

// markVisible of the upstream module (unused by the reference, kept for API completeness): the near-plane test of
// the preprocess stage, view z > 0.2.
__global__ __launch_bounds__(256) void mark_visible_kernel(int P, const float* __restrict__ means3D,
                                                           const float* __restrict__ viewmatrix,
                                                           uint8_t* __restrict__ visible) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const float x = means3D[3 * (size_t)i], y = means3D[3 * (size_t)i + 1], z = means3D[3 * (size_t)i + 2];
  const float vz = viewmatrix[2] * x + viewmatrix[6] * y + viewmatrix[10] * z + viewmatrix[14];
  visible[i] = vz > NEAR_Z ? 1 : 0;
}

__global__ __launch_bounds__(PRE_BLOCK) void unpack_geom_kernel(int P, const GeomRec* __restrict__ rec,
                                                                const BinInfo* __restrict__ bin,
                                                                const uint32_t* __restrict__ block_offs, float* xy,
                                                                float* conic_opacity, float* rgb, float* depth,
                                                                uint32_t* tiles, uint32_t* point_offsets,
                                                                uint32_t* rect, uint32_t* clamped) {
  __shared__ uint32_t wave_tot[PRE_BLOCK / WAVE];
  const int i = blockIdx.x * PRE_BLOCK + threadIdx.x;
  {   // upstream's point_offsets: inclusive scan of tiles_touched in index order (block level + in-block)
    const uint32_t t = i < P ? bin_count(bin[i].rect_wh, bin[i].mask) : 0u;
    const uint32_t inc = wave_incl_scan_u32(t);
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
    if (lane == WAVE - 1) wave_tot[wid] = inc;
    __syncthreads();
    uint32_t base = block_offs[blockIdx.x];
    for (int w = 0; w < wid; ++w) base += wave_tot[w];
    if (point_offsets && i < P) point_offsets[i] = base + inc;
  }
  if (i >= P) return;
  const BinInfo b = bin[i];
  const bool vis = b.rect_wh != 0;      // has a GeomRec (radius > 0 and a non-empty rect)
  GeomRec g;
  if (vis) g = rec[i];
  if (xy) { xy[2 * i] = vis ? g.x : 0.f; xy[2 * i + 1] = vis ? g.y : 0.f; }
  if (conic_opacity) {
    conic_opacity[4 * i] = vis ? g.cxx : 0.f; conic_opacity[4 * i + 1] = vis ? g.cxy : 0.f;
    conic_opacity[4 * i + 2] = vis ? g.cyy : 0.f; conic_opacity[4 * i + 3] = vis ? g.opacity : 0.f;
  }
  if (rgb) { rgb[3 * i] = vis ? g.r : 0.f; rgb[3 * i + 1] = vis ? g.g : 0.f; rgb[3 * i + 2] = vis ? g.b : 0.f; }
  if (depth) depth[i] = vis ? b.depth : 0.f;
  if (tiles) tiles[i] = bin_count(b.rect_wh, b.mask);
  if (rect) {
    const uint32_t x0 = b.rect_min & 0xffffu, y0 = b.rect_min >> 16;
    rect[4 * i] = vis ? x0 : 0u; rect[4 * i + 1] = vis ? y0 : 0u;
    rect[4 * i + 2] = vis ? x0 + (b.rect_wh & 0xffffu) : 0u; rect[4 * i + 3] = vis ? y0 + (b.rect_wh >> 16) : 0u;
  }
  if (clamped) clamped[i] = vis ? rect_clamp_flags(g.rect_min) : 0u;
}


size_t l1_loss_workspace_bytes() { return sizeof(float) * L1_MAX_BLOCKS; }
void launch_l1_loss(const float* x, const float* gt, size_t n, float scale, float* loss_sum, float* dL_dx,
                    float* partials, hipStream_t s) {
  size_t blocks = (n / 4 + L1_THREADS - 1) / L1_THREADS;
  if (blocks > (size_t)L1_MAX_BLOCKS) blocks = L1_MAX_BLOCKS;
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL(l1_loss_kernel, dim3((unsigned)blocks), dim3(L1_THREADS), 0, s, x, gt, n, scale, partials, dL_dx);
  hipLaunchKernelGGL(l1_loss_finish_kernel, dim3(1), dim3(WAVE), 0, s, partials, (int)blocks, loss_sum);
}
void launch_densify_stats(int P, const float* dL_dmeans2D, const int32_t* radii, float* accum, float* denom,
                          float* max_radii2D, hipStream_t s) {
  if (P <= 0) return;
  hipLaunchKernelGGL(densify_stats_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, dL_dmeans2D, radii, accum, denom,
                     max_radii2D);
}
void launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* visible, hipStream_t s) {
  if (P <= 0) return;
  hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, viewmatrix, visible);
}
void launch_unpack_geom(int P, const GeomRec* rec, const BinInfo* bin, const uint32_t* block_offs, float* xy,
                        float* conic_opacity, float* rgb, float* depth, uint32_t* tiles, uint32_t* point_offsets,
                        uint32_t* rect, uint32_t* clamped, hipStream_t s) {
  if (P <= 0) return;
  hipLaunchKernelGGL(unpack_geom_kernel, dim3((P + PRE_BLOCK - 1) / PRE_BLOCK), dim3(PRE_BLOCK), 0, s, P, rec, bin,
                     block_offs, xy, conic_opacity, rgb, depth, tiles, point_offsets, rect, clamped);
}

}  // namespace gsr
