// distCUDA2 replacement: mean squared distance of every point to its 3 nearest neighbours,
// the only other native dependency of the reference (simple_knn, an absent submodule; call site
// scene/gaussian_model.py:21,210 -- used once per scene to initialise the scales).
// Exact k-NN on a uniform grid: points are bucketed by cell with the stable radix sort of the rasterizer, and
// every query grows its search cube ring by ring until the 3rd best distance is provably final (all unsearched
// points lie at least r cells away); queries that would need more than KNN_MAX_RING rings (isolated outliers)
// finish with a brute-force sweep.
#include <float.h>

#include "gsr_common.h"
#include "gsr_launch.h"

namespace gsr {

constexpr int KNN_MAX_RING = 6;

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

struct KnnGrid {          // device-resident; written by knn_grid_kernel
  float minx, miny, minz, inv_cell, cell;
  int gx, gy, gz;
};

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

__global__ void knn_grid_kernel(const unsigned* __restrict__ mm, int N, int max_dim, KnnGrid* __restrict__ g) {
  const float lx = ord2f(mm[0]), ly = ord2f(mm[1]), lz = ord2f(mm[2]);
  const float ex = ord2f(mm[3]) - lx, ey = ord2f(mm[4]) - ly, ez = ord2f(mm[5]) - lz;
  const float ext = fmaxf(fmaxf(ex, ey), fmaxf(ez, 1e-20f));
  // ~4 points per occupied cell for a volume-filling cloud, capped so that cell ids fit the ranges array
  int res = (int)cbrtf((float)N * 0.25f);
  res = max(1, min(res, max_dim));
  const float cell = ext / (float)res * 1.0001f;
  g->minx = lx; g->miny = ly; g->minz = lz;
  g->cell = cell; g->inv_cell = 1.0f / cell;
  g->gx = min(max_dim, (int)(ex / cell) + 1);
  g->gy = min(max_dim, (int)(ey / cell) + 1);
  g->gz = min(max_dim, (int)(ez / cell) + 1);
}

__device__ inline int3 cell_of(const KnnGrid& g, float x, float y, float z) {
  int3 c;
  c.x = min(g.gx - 1, max(0, (int)((x - g.minx) * g.inv_cell)));
  c.y = min(g.gy - 1, max(0, (int)((y - g.miny) * g.inv_cell)));
  c.z = min(g.gz - 1, max(0, (int)((z - g.minz) * g.inv_cell)));
  return c;
}

__global__ __launch_bounds__(256) void knn_cell_keys_kernel(const float* __restrict__ pts, int N,
                                                            const KnnGrid* __restrict__ gp, uint32_t* __restrict__ keys,
                                                            uint32_t* __restrict__ vals) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const KnnGrid g = *gp;
  const int3 c = cell_of(g, pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2]);
  keys[i] = (uint32_t)((c.z * g.gy + c.y) * g.gx + c.x);
  vals[i] = (uint32_t)i;
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

__global__ __launch_bounds__(256) void knn_query_kernel(const float* __restrict__ pts, int N,
                                                        const KnnGrid* __restrict__ gp,
                                                        const uint32_t* __restrict__ sorted_idx,
                                                        const uint2* __restrict__ ranges,
                                                        float* __restrict__ mean_dist2) {
  const int s = blockIdx.x * 256 + threadIdx.x;   // query in cell order: neighbouring lanes walk the same cells
  if (s >= N) return;
  const KnnGrid g = *gp;
  const uint32_t self = sorted_idx[s];
  const float px = pts[3 * (size_t)self], py = pts[3 * (size_t)self + 1], pz = pts[3 * (size_t)self + 2];
  const int3 c = cell_of(g, px, py, pz);
  float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
  auto visit = [&](uint32_t j) {
    if (j == self) return;
    const float dx = pts[3 * (size_t)j] - px, dy = pts[3 * (size_t)j + 1] - py, dz = pts[3 * (size_t)j + 2] - pz;
    keep3(dx * dx + dy * dy + dz * dz, best);
  };
  bool done = false;
  const int rmax = max(g.gx, max(g.gy, g.gz));
  for (int r = 0; r <= KNN_MAX_RING && !done; ++r) {
    for (int dz = -r; dz <= r; ++dz) {
      const int z = c.z + dz;
      if (z < 0 || z >= g.gz) continue;
      for (int dy = -r; dy <= r; ++dy) {
        const int y = c.y + dy;
        if (y < 0 || y >= g.gy) continue;
        const bool face = (dz == -r || dz == r || dy == -r || dy == r);
        for (int dx = -r; dx <= r; dx += (face ? 1 : 2 * r > 0 ? 2 * r : 1)) {   // shell only
          const int x = c.x + dx;
          if (x < 0 || x >= g.gx) continue;
          const uint2 rg = ranges[(z * g.gy + y) * g.gx + x];
          for (uint32_t k = rg.x; k < rg.y; ++k) visit(sorted_idx[k]);
        }
      }
    }
    // every point outside the searched cube is at least r cells away (the query sits inside its own cell)
    const float bound = (float)r * g.cell;
    done = best[2] <= bound * bound || r >= rmax;
  }
  if (!done) {   // isolated point: exact fallback
    best[0] = best[1] = best[2] = FLT_MAX;
    for (int j = 0; j < N; ++j) visit((uint32_t)j);
  }
  mean_dist2[self] = (best[0] + best[1] + best[2]) / 3.0f;
}

constexpr int KNN_MAX_DIM = 160;   // 160^3 = 4.1 M cells

size_t knn_workspace_bytes(int N) {
  size_t n = N > 0 ? (size_t)N : 1, o = 0;
  o = align_up(o + 64, 256);                                 // min/max + grid
  o = align_up(o + 4 * n, 256) * 1;                          // keys a
  o += align_up(4 * n, 256) * 3;                             // keys b, vals a, vals b
  o += align_up(8 * (size_t)KNN_MAX_DIM * KNN_MAX_DIM * KNN_MAX_DIM, 256);   // cell ranges
  o += align_up(SortLayout((uint32_t)n).bytes, 256);
  return o;
}

void launch_knn3(const float* pts, int N, float* mean_dist2, void* ws, hipStream_t s) {
  if (N <= 0) return;
  char* b = static_cast<char*>(ws);
  const size_t n = (size_t)N, a4 = align_up(4 * n, 256);
  unsigned* mm = reinterpret_cast<unsigned*>(b);
  KnnGrid* grid = reinterpret_cast<KnnGrid*>(b + 32);
  size_t o = 256;
  uint32_t* ka = reinterpret_cast<uint32_t*>(b + o); o += a4;
  uint32_t* kb = reinterpret_cast<uint32_t*>(b + o); o += a4;
  uint32_t* va = reinterpret_cast<uint32_t*>(b + o); o += a4;
  uint32_t* vb = reinterpret_cast<uint32_t*>(b + o); o += a4;
  uint2* ranges = reinterpret_cast<uint2*>(b + o);
  const size_t cells = (size_t)KNN_MAX_DIM * KNN_MAX_DIM * KNN_MAX_DIM;
  o += align_up(8 * cells, 256);
  void* scratch = b + o;
  const unsigned init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
  (void)hipMemcpyAsync(mm, init, sizeof(init), hipMemcpyHostToDevice, s);
  (void)hipMemsetAsync(ranges, 0, 8 * cells, s);
  const int nb = (N + 255) / 256;
  hipLaunchKernelGGL(knn_bbox_kernel, dim3(nb < 1024 ? nb : 1024), dim3(256), 0, s, pts, N, mm);
  hipLaunchKernelGGL(knn_grid_kernel, dim3(1), dim3(1), 0, s, mm, N, KNN_MAX_DIM, grid);
  hipLaunchKernelGGL(knn_cell_keys_kernel, dim3(nb), dim3(256), 0, s, pts, N, grid, ka, va);
  int bits = 0;
  while ((1ull << bits) < cells) ++bits;
  const bool in_b = launch_sort_pairs_u32(ka, va, kb, vb, (uint32_t)N, bits, scratch, s);
  launch_identify_tile_ranges_u32((uint32_t)N, in_b ? kb : ka, ranges, s);
  hipLaunchKernelGGL(knn_query_kernel, dim3(nb), dim3(256), 0, s, pts, N, grid, in_b ? vb : va, ranges, mean_dist2);
}

}  // namespace gsr
