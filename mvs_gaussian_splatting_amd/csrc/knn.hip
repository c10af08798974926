// distCUDA2 replacement: mean squared distance of every point to its 3 nearest neighbours,
// the only other native dependency of the reference (simple_knn, an absent submodule; call site
// scene/gaussian_model.py:21,210 -- used once per scene to initialise the scales).
// Exact k-NN, robust to any density (SfM clouds are anything but uniform): the points are sorted along a Morton
// curve (30-bit codes, the rasterizer's stable radix sort), consecutive runs of 128 sorted points form boxes and 64
// boxes a super-box, each with its bounding box.  A query scans its own box, then walks the super-boxes, descends
// into those whose bounding box is closer than its current 3rd-best distance, and likewise into their boxes: a box
// that cannot hold a closer point is never opened, so the result is exact.  Boxes hold a fixed NUMBER of points, so
// dense regions get small boxes -- a uniform grid of cells degenerates to O(n^2) there (15 s for 6 M clustered points).
#include <float.h>

#include "gsr_common.h"
#include "gsr_launch.h"

namespace gsr {

constexpr int KNN_BOX = 128;        // points per box
constexpr int KNN_SUPER = 64;       // boxes per super-box

__device__ inline unsigned f2ord(float f) {   // order-preserving float -> uint
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float ord2f(unsigned o) {
  const unsigned u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
#if defined(__HIP_DEVICE_COMPILE__)
  return __uint_as_float(u);
#else
  float f; memcpy(&f, &u, 4); return f;
#endif
}

__global__ __launch_bounds__(256) void knn_bbox_kernel(const float* __restrict__ pts, int N, unsigned* __restrict__ mm) {
  unsigned lo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, hi[3] = {0u, 0u, 0u};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const unsigned o = f2ord(pts[3 * (size_t)i + a]);
      lo[a] = min(lo[a], o);
      hi[a] = max(hi[a], o);
    }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) {
      lo[a] = min(lo[a], (unsigned)__shfl_xor((int)lo[a], d, WAVE));
      hi[a] = max(hi[a], (unsigned)__shfl_xor((int)hi[a], d, WAVE));
    }
    if ((threadIdx.x & (WAVE - 1)) == 0) {
      atomicMin(&mm[a], lo[a]);
      atomicMax(&mm[3 + a], hi[a]);
    }
  }
}

__device__ inline void keep3(float d, float* best) {
  if (d < best[2]) {
    if (d < best[1]) {
      best[2] = best[1];
      if (d < best[0]) { best[1] = best[0]; best[0] = d; } else best[1] = d;
    } else {
      best[2] = d;
    }
  }
}

__device__ inline uint32_t spread10(uint32_t v) {   // 10 bits -> every third bit
  v &= 0x3ffu;
  v = (v | (v << 16)) & 0x030000ffu;
  v = (v | (v << 8)) & 0x0300f00fu;
  v = (v | (v << 4)) & 0x030c30c3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}

__global__ __launch_bounds__(256) void knn_morton_kernel(const float* __restrict__ pts, int N,
                                                         const unsigned* __restrict__ mm, uint32_t* __restrict__ keys,
                                                         uint32_t* __restrict__ vals) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  uint32_t code = 0;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float lo = ord2f(mm[a]), hi = ord2f(mm[3 + a]);
    const float t = (pts[3 * (size_t)i + a] - lo) / fmaxf(hi - lo, 1e-30f);
    const uint32_t q = (uint32_t)fminf(1023.0f, fmaxf(0.0f, t * 1023.0f));
    code |= spread10(q) << a;
  }
  keys[i] = code;
  vals[i] = (uint32_t)i;
}

// sorted points as (x, y, z, original index) + the bounding box of every run of KNN_BOX of them
__global__ __launch_bounds__(KNN_BOX) void knn_boxes_kernel(const float* __restrict__ pts, int N,
                                                            const uint32_t* __restrict__ sorted_idx,
                                                            float4* __restrict__ spts, float* __restrict__ box_lo,
                                                            float* __restrict__ box_hi) {
  __shared__ float red[6][KNN_BOX / WAVE];
  const int s = blockIdx.x * KNN_BOX + threadIdx.x;
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  if (s < N) {
    const uint32_t i = sorted_idx[s];
    const float x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
    spts[s] = make_float4(x, y, z, __uint_as_float(i));
    lo[0] = hi[0] = x; lo[1] = hi[1] = y; lo[2] = hi[2] = z;
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], d, WAVE));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], d, WAVE));
    }
    if ((threadIdx.x & (WAVE - 1)) == 0) { red[a][threadIdx.x / WAVE] = lo[a]; red[3 + a][threadIdx.x / WAVE] = hi[a]; }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    float l = red[threadIdx.x][0], h = red[3 + threadIdx.x][0];
    for (int w = 1; w < KNN_BOX / WAVE; ++w) { l = fminf(l, red[threadIdx.x][w]); h = fmaxf(h, red[3 + threadIdx.x][w]); }
    box_lo[3 * (size_t)blockIdx.x + threadIdx.x] = l;
    box_hi[3 * (size_t)blockIdx.x + threadIdx.x] = h;
  }
}

__global__ __launch_bounds__(KNN_SUPER) void knn_super_kernel(int nboxes, const float* __restrict__ box_lo,
                                                              const float* __restrict__ box_hi,
                                                              float* __restrict__ sup_lo, float* __restrict__ sup_hi) {
  const int b = blockIdx.x * KNN_SUPER + threadIdx.x;     // one wave per super-box
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  if (b < nboxes) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { lo[a] = box_lo[3 * (size_t)b + a]; hi[a] = box_hi[3 * (size_t)b + a]; }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], d, WAVE));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], d, WAVE));
    }
    if (threadIdx.x == 0) { sup_lo[3 * (size_t)blockIdx.x + a] = lo[a]; sup_hi[3 * (size_t)blockIdx.x + a] = hi[a]; }
  }
}

__device__ inline float box_dist2(const float* __restrict__ lo, const float* __restrict__ hi, float px, float py, float pz) {
  const float dx = fmaxf(fmaxf(lo[0] - px, px - hi[0]), 0.0f);
  const float dy = fmaxf(fmaxf(lo[1] - py, py - hi[1]), 0.0f);
  const float dz = fmaxf(fmaxf(lo[2] - pz, pz - hi[2]), 0.0f);
  return dx * dx + dy * dy + dz * dz;
}

__global__ __launch_bounds__(256) void knn_query_kernel(int N, int nboxes, int nsuper, const float4* __restrict__ spts,
                                                        const float* __restrict__ box_lo, const float* __restrict__ box_hi,
                                                        const float* __restrict__ sup_lo, const float* __restrict__ sup_hi,
                                                        float* __restrict__ mean_dist2) {
  const int s = blockIdx.x * 256 + threadIdx.x;   // queries in Morton order: neighbouring lanes open the same boxes
  if (s >= N) return;
  const float4 me = spts[s];
  const float px = me.x, py = me.y, pz = me.z;
  float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
  auto scan_box = [&](int b) {
    const int first = b * KNN_BOX, last = min(N, first + KNN_BOX);
    for (int j = first; j < last; ++j) {
      if (j == s) continue;
      const float4 q = spts[j];
      const float dx = q.x - px, dy = q.y - py, dz = q.z - pz;
      keep3(dx * dx + dy * dy + dz * dz, best);
    }
  };
  const int own = s / KNN_BOX;
  scan_box(own);                                   // a good bound before any pruning test
  for (int S = 0; S < nsuper; ++S) {
    if (!(box_dist2(sup_lo + 3 * (size_t)S, sup_hi + 3 * (size_t)S, px, py, pz) < best[2])) continue;
    const int b0 = S * KNN_SUPER, b1 = min(nboxes, b0 + KNN_SUPER);
    for (int b = b0; b < b1; ++b) {
      if (b == own) continue;
      if (box_dist2(box_lo + 3 * (size_t)b, box_hi + 3 * (size_t)b, px, py, pz) < best[2]) scan_box(b);
    }
  }
  mean_dist2[__float_as_uint(me.w)] = (best[0] + best[1] + best[2]) / 3.0f;
}

struct KnnLayout {
  size_t mm, ka, kb, va, vb, spts, box_lo, box_hi, sup_lo, sup_hi, sort, bytes;
  int nboxes, nsuper;
  explicit KnnLayout(int N) {
    const size_t n = N > 0 ? (size_t)N : 1;
    nboxes = (int)((n + KNN_BOX - 1) / KNN_BOX);
    nsuper = (nboxes + KNN_SUPER - 1) / KNN_SUPER;
    size_t o = 0;
    mm = o;     o = align_up(o + 64, 256);
    ka = o;     o = align_up(o + 4 * n, 256);
    kb = o;     o = align_up(o + 4 * n, 256);
    va = o;     o = align_up(o + 4 * n, 256);
    vb = o;     o = align_up(o + 4 * n, 256);
    spts = o;   o = align_up(o + 16 * n, 256);
    box_lo = o; o = align_up(o + 12 * (size_t)nboxes, 256);
    box_hi = o; o = align_up(o + 12 * (size_t)nboxes, 256);
    sup_lo = o; o = align_up(o + 12 * (size_t)nsuper, 256);
    sup_hi = o; o = align_up(o + 12 * (size_t)nsuper, 256);
    sort = o;   o = align_up(o + SortLayout((uint32_t)n).bytes, 256);
    bytes = o;
  }
};

size_t knn_workspace_bytes(int N) { return KnnLayout(N).bytes; }

void launch_knn3(const float* pts, int N, float* mean_dist2, void* ws, hipStream_t s) {
  if (N <= 0) return;
  const KnnLayout L(N);
  char* b = static_cast<char*>(ws);
  unsigned* mm = reinterpret_cast<unsigned*>(b + L.mm);
  uint32_t* ka = reinterpret_cast<uint32_t*>(b + L.ka);
  uint32_t* kb = reinterpret_cast<uint32_t*>(b + L.kb);
  uint32_t* va = reinterpret_cast<uint32_t*>(b + L.va);
  uint32_t* vb = reinterpret_cast<uint32_t*>(b + L.vb);
  float4* spts = reinterpret_cast<float4*>(b + L.spts);
  float* box_lo = reinterpret_cast<float*>(b + L.box_lo);
  float* box_hi = reinterpret_cast<float*>(b + L.box_hi);
  float* sup_lo = reinterpret_cast<float*>(b + L.sup_lo);
  float* sup_hi = reinterpret_cast<float*>(b + L.sup_hi);
  // min words start at 0xffffffff, max words at 0 (order-preserving uint encoding of the coordinates)
  (void)hipMemsetAsync(mm, 0xff, 12, s);
  (void)hipMemsetAsync(mm + 3, 0, 12, s);
  const int nb = (N + 255) / 256;
  hipLaunchKernelGGL(knn_bbox_kernel, dim3(nb < 1024 ? nb : 1024), dim3(256), 0, s, pts, N, mm);
  hipLaunchKernelGGL(knn_morton_kernel, dim3(nb), dim3(256), 0, s, pts, N, mm, ka, va);
  const bool in_b = launch_sort_pairs_u32(ka, va, kb, vb, (uint32_t)N, 30, b + L.sort, s);
  hipLaunchKernelGGL(knn_boxes_kernel, dim3(L.nboxes), dim3(KNN_BOX), 0, s, pts, N, in_b ? vb : va, spts, box_lo, box_hi);
  hipLaunchKernelGGL(knn_super_kernel, dim3(L.nsuper), dim3(KNN_SUPER), 0, s, L.nboxes, box_lo, box_hi, sup_lo, sup_hi);
  hipLaunchKernelGGL(knn_query_kernel, dim3(nb), dim3(256), 0, s, N, L.nboxes, L.nsuper, spts, box_lo, box_hi, sup_lo,
                     sup_hi, mean_dist2);
}

}  // namespace gsr
