// VALU issue-rate microbenchmark: plain v_fma_f32 vs v_pk_fma_f32 vs a compare/cndmask mix, at 1..8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float* out, int iters, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  float2v p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
  float2v pa = {a, a}, pb = {b, b};
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {   // 8 independent scalar fma
      x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
      x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
    } else if (MODE == 1) {   // 4 independent packed fma (same flops as 8 scalar)
      p0 = __builtin_elementwise_fma(p0, pa, pb); p1 = __builtin_elementwise_fma(p1, pa, pb);
      p2 = __builtin_elementwise_fma(p2, pa, pb); p3 = __builtin_elementwise_fma(p3, pa, pb);
    } else {   // dependent chain of 8 scalar fma
      x0 = __builtin_fmaf(x0, a, b); x0 = __builtin_fmaf(x0, a, b); x0 = __builtin_fmaf(x0, a, b); x0 = __builtin_fmaf(x0, a, b);
      x0 = __builtin_fmaf(x0, a, b); x0 = __builtin_fmaf(x0, a, b); x0 = __builtin_fmaf(x0, a, b); x0 = __builtin_fmaf(x0, a, b);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int MODE>
void run(const char* name, int waves_per_simd) {
  const int cus = 256, iters = 20000;
  const int blocks = cus * waves_per_simd;      // 256-thread blocks: 4 waves = 1 per SIMD
  float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(out, 100, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(out, iters, 1.0001f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_wave = (double)iters * (MODE == 1 ? 4 : 8);
  const double waves_per_simd_total = waves_per_simd;   // each SIMD hosts this many waves
  // wall cycles per instruction issued on one SIMD (assume 2.4 GHz)
  const double cyc = ms * 1e-3 * 2.4e9 / (instr_per_wave * waves_per_simd_total);
  printf("%-12s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz), %.1f TFLOP/s\n", name,
         waves_per_simd, ms, cyc, (double)blocks * 256 * iters * 16 / (ms * 1e-3) / 1e12);
  hipFree(out);
}
int main() {
  for (int w : {1, 2, 4, 8}) { run<0>("fma x8", w); run<1>("pk_fma x4", w); run<2>("fma chain", w); }
  return 0;
}
