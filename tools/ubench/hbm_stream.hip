// What the device streams at: read-only, write-only and copy kernels over 1 GiB buffers, 16 bytes per lane per access,
// with plain and nontemporal accesses and several grid sizes.  The streaming stages of the frame (preprocess_fwd,
// preprocess_bwd) are judged against these rates.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/hbm_stream.hip -o gpurun_out/hbm_stream && gpurun_out/hbm_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef float v4 __attribute__((ext_vector_type(4)));

template <int MODE>      // 0 copy, 1 copy nt store, 2 copy nt load + nt store, 3 write only, 4 write only nt, 5 read only
__global__ __launch_bounds__(256) void k(const v4* __restrict__ a, v4* __restrict__ b, size_t n, float* sink) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  v4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    if (MODE == 0) b[i] = a[i];
    if (MODE == 1) __builtin_nontemporal_store(a[i], &b[i]);
    if (MODE == 2) __builtin_nontemporal_store(__builtin_nontemporal_load(&a[i]), &b[i]);
    if (MODE == 3) b[i] = v4{1.f, 2.f, 3.f, (float)i};
    if (MODE == 4) __builtin_nontemporal_store(v4{1.f, 2.f, 3.f, (float)i}, &b[i]);
    if (MODE == 5) { const v4 x = a[i]; acc += x; }
  }
  if (MODE == 5 && acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
}

template <int MODE>
static void run(const char* name, const v4* a, v4* b, size_t n, float* sink, double bytes_per_elem) {
  const int grids[] = {256 * 4, 256 * 8, 256 * 16, 256 * 32, 0};
  for (int gi = 0; gi < 5; ++gi) {
    const int grid = grids[gi] ? grids[gi] : (int)((n + 255) / 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, a, b, n, sink);
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, a, b, n, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s grid %8d: %7.1f GB/s\n", name, grid, 10 * bytes_per_elem * n / (ms * 1e-3) / 1e9);
  }
}

int main() {
  const size_t n = (size_t)1 << 26;      // 64 Mi x 16 bytes = 1 GiB
  v4 *a, *b;
  float* sink;
  hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&sink, 4);
  hipMemset(a, 0, n * 16); hipMemset(b, 0, n * 16);
  run<0>("copy (read + write)", a, b, n, sink, 32);
  run<1>("copy, nontemporal store", a, b, n, sink, 32);
  run<2>("copy, nontemporal load + store", a, b, n, sink, 32);
  run<3>("write only", a, b, n, sink, 16);
  run<4>("write only, nontemporal", a, b, n, sink, 16);
  run<5>("read only", a, b, n, sink, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int r = 0; r < 10; ++r) hipMemcpyAsync(b, a, n * 16, hipMemcpyDeviceToDevice, 0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  printf("%-34s               : %7.1f GB/s\n", "hipMemcpyAsync device to device", 10 * 32.0 * n / (ms * 1e-3) / 1e9);
  return 0;
}
