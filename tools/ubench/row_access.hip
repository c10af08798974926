// The memory behaviour of preprocess_fwd in isolation (no projection, no SH arithmetic): per Gaussian 44 bytes of
// position / scale / rotation / opacity always, and for the ~65 % that are visible 12 B f_dc + 180 B f_rest read and a
// 64-byte record written, plus 16 B BinInfo + 4 B radius for everyone.  Which ACCESS SHAPE streams those bytes fastest?
//   R0  row per lane, as the compiler emits it for `for (i < 45) f[i] = rr[i]` (4-byte aligned rows: dwordx4/x3/x2/x1 mixes)
//   R1  row per lane, 45 single dword loads (forced)
//   R2  wave-cooperative, ALL 64 rows (11520 contiguous bytes, 16 B per lane per access) through LDS -- reads the rows of
//       invisible Gaussians too
//   R3  wave-cooperative, VISIBLE rows only: the wave's visible rows are enumerated, 16-byte chunk c of the list goes to
//       lane c % 64 (12 chunks per row: adjacent lanes read adjacent bytes), through LDS
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/row_access.hip -o tools/ubench/bin/row_access
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

constexpr int ROW = 45;
struct __attribute__((aligned(64))) Rec { float v[16]; };

__device__ inline bool visible(uint32_t i, uint32_t pct) {
  uint32_t h = i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  return (h % 100u) < pct;
}

typedef float v4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float v4a __attribute__((ext_vector_type(4)));

// R4 / R5 / R7: row per lane with explicit 16-byte pieces.  NT_LOAD: nontemporal row loads; NT_STORE: nontemporal stores
// of the record / BinInfo / radius; PER: Gaussians per lane (the second one 256 further on: both rows' loads in flight)
template <bool NT_LOAD, bool NT_STORE, int PER>
__global__ __launch_bounds__(256) void k2(int P, const float* __restrict__ xyz, const float* __restrict__ scl,
                                          const float4* __restrict__ rot, const float* __restrict__ opa,
                                          const float* __restrict__ fdc, const float* __restrict__ frest,
                                          Rec* __restrict__ rec, float4* __restrict__ bin, int* __restrict__ radii,
                                          uint32_t pct) {
  float acc[PER];
  bool vis[PER];
  float f[PER][48];
  int idx[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    idx[u] = (blockIdx.x * PER + u) * 256 + threadIdx.x;
    acc[u] = 0.f; vis[u] = false;
    if (idx[u] < P) {
      const size_t i = idx[u];
      const float px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
      const float s0 = scl[3 * i], s1 = scl[3 * i + 1], s2 = scl[3 * i + 2];
      const float4 q = rot[i];
      const float o = opa[i];
      acc[u] = px + py + pz + s0 + s1 + s2 + q.x + q.y + q.z + q.w + o;
      vis[u] = visible((uint32_t)i, pct) && acc[u] != 12345.6789f;
    }
  }
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    if (vis[u]) {
      const float* __restrict__ rr = frest + (size_t)idx[u] * ROW;
#pragma unroll
      for (int c = 0; c < 11; ++c) {
        const v4u v = NT_LOAD ? __builtin_nontemporal_load(reinterpret_cast<const v4u*>(rr + 4 * c))
                              : *reinterpret_cast<const v4u*>(rr + 4 * c);
        f[u][3 + 4 * c] = v.x; f[u][4 + 4 * c] = v.y; f[u][5 + 4 * c] = v.z; f[u][6 + 4 * c] = v.w;
      }
      f[u][47] = NT_LOAD ? __builtin_nontemporal_load(rr + 44) : rr[44];
      f[u][0] = fdc[3 * (size_t)idx[u]]; f[u][1] = fdc[3 * (size_t)idx[u] + 1]; f[u][2] = fdc[3 * (size_t)idx[u] + 2];
    }
  }
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    if (vis[u]) {
      v4a* dst = reinterpret_cast<v4a*>(rec + idx[u]);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        v4a r;
        r.x = f[u][4 * c] + f[u][16 + 4 * c] + f[u][32 + 4 * c] + acc[u];
        r.y = f[u][4 * c + 1] + f[u][17 + 4 * c] + f[u][33 + 4 * c] + acc[u];
        r.z = f[u][4 * c + 2] + f[u][18 + 4 * c] + f[u][34 + 4 * c] + acc[u];
        r.w = f[u][4 * c + 3] + f[u][19 + 4 * c] + f[u][35 + 4 * c] + acc[u];
        if (NT_STORE) __builtin_nontemporal_store(r, dst + c); else dst[c] = r;
      }
    }
    if (idx[u] < P) {
      const int rad = vis[u] ? 3 : 0;
      v4a b = {vis[u] ? acc[u] : 0.f, vis[u] ? 1.f : 0.f, vis[u] ? 2.f : 0.f, vis[u] ? 3.f : 0.f};
      if (NT_STORE) {
        __builtin_nontemporal_store(rad, radii + idx[u]);
        __builtin_nontemporal_store(b, reinterpret_cast<v4a*>(bin + idx[u]));
      } else {
        radii[idx[u]] = rad;
        *reinterpret_cast<v4a*>(bin + idx[u]) = b;
      }
    }
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void k(int P, const float* __restrict__ xyz, const float* __restrict__ scl,
                                         const float4* __restrict__ rot, const float* __restrict__ opa,
                                         const float* __restrict__ fdc, const float* __restrict__ frest,
                                         Rec* __restrict__ rec, float4* __restrict__ bin, int* __restrict__ radii,
                                         uint32_t pct) {
  __shared__ float stage[MODE >= 2 ? 4 * 64 * ROW : 1];
  __shared__ int rowlist[256];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float acc = 0.f;
  bool vis = false;
  float f[48];
  if (idx < P) {
    const float px = xyz[3 * (size_t)idx], py = xyz[3 * (size_t)idx + 1], pz = xyz[3 * (size_t)idx + 2];
    const float s0 = scl[3 * (size_t)idx], s1 = scl[3 * (size_t)idx + 1], s2 = scl[3 * (size_t)idx + 2];
    const float4 q = rot[idx];
    const float o = opa[idx];
    acc = px + py + pz + s0 + s1 + s2 + q.x + q.y + q.z + q.w + o;
    vis = visible((uint32_t)idx, pct) && acc != 12345.6789f;
  }
  if (MODE == 0 || MODE == 1) {
    if (vis) {
      const float* __restrict__ rr = frest + (size_t)idx * ROW;
      if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < ROW; ++i) f[3 + i] = rr[i];
      } else {
#pragma unroll
        for (int i = 0; i < ROW; ++i) f[3 + i] = __builtin_nontemporal_load(rr + i) ;
      }
    }
  } else if (MODE == 2) {
    float* st = stage + wid * 64 * ROW;
    const int first = blockIdx.x * 256 + wid * 64;
    const int rows = min(64, P - first);
    const float* __restrict__ src = frest + (size_t)first * ROW;
    const int nfl = rows * ROW;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int e = (i * 64 + lane) * 4;
      if (e + 3 < nfl) {
        const float4 v = *reinterpret_cast<const float4*>(src + e);
        *reinterpret_cast<float4*>(st + e) = v;
      } else {
        for (int t = 0; t < 4; ++t) if (e + t < nfl) st[e + t] = src[e + t];
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (vis) {
#pragma unroll
      for (int i = 0; i < ROW; ++i) f[3 + i] = st[lane * ROW + i];
    }
  } else {
    // visible rows of the wave, enumerated; chunk c (16 B; 12 per row, the last one 4 B) -> lane c % 64
    float* st = stage + wid * 64 * ROW;
    const int first = blockIdx.x * 256 + wid * 64;
    const unsigned long long m = __ballot(vis);
    const int nvis = __popcll(m);
    const int my_slot = __popcll(m & ((1ull << lane) - 1ull));
    const int nchunks = nvis * 12;
    if (vis) rowlist[wid * 64 + my_slot] = lane;
    __builtin_amdgcn_wave_barrier();
    for (int c0 = 0; c0 < nchunks; c0 += 64) {
      const int c = c0 + lane;
      if (c < nchunks) {
        const int slot = c / 12, part = c - slot * 12;
        const int row_lane = rowlist[wid * 64 + slot];
        const float* __restrict__ src = frest + (size_t)(first + row_lane) * ROW + part * 4;
        float* d = st + slot * ROW + part * 4;
        if (part < 11) {
          const float4 v = *reinterpret_cast<const float4*>(src);
          d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        } else {
          d[0] = src[0];
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (vis) {
#pragma unroll
      for (int i = 0; i < ROW; ++i) f[3 + i] = st[my_slot * ROW + i];
    }
  }
  if (vis) {
    f[0] = fdc[3 * (size_t)idx]; f[1] = fdc[3 * (size_t)idx + 1]; f[2] = fdc[3 * (size_t)idx + 2];
    Rec r;
#pragma unroll
    for (int i = 0; i < 16; ++i) r.v[i] = f[i] + f[16 + i] + f[32 + i] + acc;
    rec[idx] = r;
  }
  if (idx < P) {
    radii[idx] = vis ? 3 : 0;
    bin[idx] = vis ? make_float4(acc, 1.f, 2.f, 3.f) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

template <int MODE>
static void run(const char* name, int P, const float* xyz, const float* scl, const float4* rot, const float* opa,
                const float* fdc, const float* frest, Rec* rec, float4* bin, int* radii, uint32_t pct) {
  const int grid = (P + 255) / 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, P, xyz, scl, rot, opa, fdc, frest, rec, bin, radii, pct);
  hipEventRecord(e0);
  const int reps = 20;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, P, xyz, scl, rot, opa, fdc, frest, rec, bin, radii, pct);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double vis = P * (pct / 100.0);
  const double bytes = 44.0 * P + (192.0 + 64.0) * vis + 20.0 * P;
  printf("%-58s visible %3u %%: %7.4f ms  %7.1f GB/s of the bytes the frame needs (%.3f GB)\n", name, pct, ms,
         bytes / (ms * 1e-3) / 1e9, bytes / 1e9);
}

template <bool NT_LOAD, bool NT_STORE, int PER>
static void run2(const char* name, int P, const float* xyz, const float* scl, const float4* rot, const float* opa,
                 const float* fdc, const float* frest, Rec* rec, float4* bin, int* radii, uint32_t pct) {
  const int grid = (P + 256 * PER - 1) / (256 * PER);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k2<NT_LOAD, NT_STORE, PER>), dim3(grid), dim3(256), 0, 0, P, xyz, scl, rot, opa, fdc, frest, rec, bin, radii, pct);
  hipEventRecord(e0);
  const int reps = 20;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k2<NT_LOAD, NT_STORE, PER>), dim3(grid), dim3(256), 0, 0, P, xyz, scl, rot, opa, fdc, frest, rec, bin, radii, pct);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double vis = P * (pct / 100.0);
  const double bytes = 44.0 * P + (192.0 + 64.0) * vis + 20.0 * P;
  printf("%-58s visible %3u %%: %7.4f ms  %7.1f GB/s of the bytes the frame needs (%.3f GB)\n", name, pct, ms,
         bytes / (ms * 1e-3) / 1e9, bytes / 1e9);
}

int main(int argc, char** argv) {
  const int P = argc > 1 ? atoi(argv[1]) : 6000000;
  float *xyz, *scl, *opa, *fdc, *frest; float4 *rot, *bin; Rec* rec; int* radii;
  hipMalloc(&xyz, 12 * (size_t)P); hipMalloc(&scl, 12 * (size_t)P); hipMalloc(&rot, 16 * (size_t)P);
  hipMalloc(&opa, 4 * (size_t)P); hipMalloc(&fdc, 12 * (size_t)P); hipMalloc(&frest, 180 * (size_t)P + 64);
  hipMalloc(&rec, 64 * (size_t)P); hipMalloc(&bin, 16 * (size_t)P); hipMalloc(&radii, 4 * (size_t)P);
  hipMemset(xyz, 0, 12 * (size_t)P); hipMemset(scl, 0, 12 * (size_t)P); hipMemset(rot, 0, 16 * (size_t)P);
  hipMemset(opa, 0, 4 * (size_t)P); hipMemset(fdc, 0, 12 * (size_t)P); hipMemset(frest, 0, 180 * (size_t)P + 64);
  for (uint32_t pct : {65u, 100u}) {
    run<0>("R0 row per lane (compiler's choice of widths)", P, xyz, scl, rot, opa, fdc, frest, rec, bin, radii, pct);
    run<1>("R1 row per lane, 45 nontemporal dword loads", P, xyz, scl, rot, opa, fdc, frest, rec, bin, radii, pct);
    run<2>("R2 wave-cooperative, all 64 rows, through LDS", P, xyz, scl, rot, opa, fdc, frest, rec, bin, radii, pct);
    run<3>("R3 wave-cooperative, visible rows only, through LDS", P, xyz, scl, rot, opa, fdc, frest, rec, bin, radii, pct);
    run2<false, false, 1>("R4 row per lane, explicit 16-byte pieces", P, xyz, scl, rot, opa, fdc, frest, rec, bin, radii, pct);
    run2<true, false, 1>("R5 R4 + nontemporal row loads", P, xyz, scl, rot, opa, fdc, frest, rec, bin, radii, pct);
    run2<false, true, 1>("R6 R4 + nontemporal stores", P, xyz, scl, rot, opa, fdc, frest, rec, bin, radii, pct);
    run2<true, true, 1>("R7 R4 + nontemporal loads and stores", P, xyz, scl, rot, opa, fdc, frest, rec, bin, radii, pct);
    run2<false, false, 2>("R8 R4, two Gaussians per lane", P, xyz, scl, rot, opa, fdc, frest, rec, bin, radii, pct);
    run2<true, true, 2>("R9 R7, two Gaussians per lane", P, xyz, scl, rot, opa, fdc, frest, rec, bin, radii, pct);
  }
  return 0;
}
