// What a per-lane select costs on gfx950, in the shapes the compositing kernels' inner loops use (follow-up of
// valu_rate.hip, whose "v_cndmask_b32 vcc x16" line reads 23 cycles against 4.7 with an SGPR-pair mask):
//   the mask in VCC or in an SGPR pair, written by a VALU compare or by a SALU s_and_b64 right in front of the select
//   (the backward's `ok = pos <= last && alpha >= 1/255`), an EXEC-masked region instead of a select, and a few
//   candidates for folding sums over the four lanes of a DPP bank.
// (every asm block that holds an s_and_b64 / s_and_saveexec_b64 declares "scc" clobbered: the loop branch lives there)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/select_rate.hip -o tools/ubench/bin/select_rate && tools/ubench/bin/select_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)

enum { SEL_VCC_VALU = 0, SEL_VCC_ONCE, SEL_VCC_SALU, SEL_SGPR_SALU, SEL_SGPR_VALU, EXEC_REGION, CMP2_AND_SEL_VCC, CMP2_AND_SEL_SGPR,
       MUL_MASK, DPP_FOLD9, DPP_MOV_ADD, FMA_REF, NMODES };
static const char* names[NMODES] = {
    "v_cmp->vcc ; v_cndmask vcc            (8 pairs)",
    "v_cmp->vcc once ; 16 x v_cndmask vcc",
    "s_and_b64 vcc ; v_cndmask vcc         (8 pairs)",
    "s_and_b64 s[24:25] ; v_cndmask_e64    (8 pairs)",
    "v_cmp->s[24:25] ; v_cndmask_e64       (8 pairs)",
    "v_cmp ; s_and_saveexec ; 4 fma ; s_mov exec (4x)",
    "2 v_cmp->sgpr ; s_and vcc ; cndmask vcc (4x: bwd shape)",
    "2 v_cmp->sgpr ; s_and sgpr ; cndmask e64 (4x)",
    "v_cmp ; v_cndmask 1.0/0 ... as v_mul by mask (8 pairs)",
    "quad_fold9: 18 v_add_f32_dpp",
    "9 x (v_mov_dpp + v_add) x 2",
    "v_fma_f32 x16 (reference)"};
static const int insts[NMODES] = {16, 17, 8, 8, 16, 24, 12, 12, 16, 18, 36, 16};   // VALU instructions per iteration

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* clk, int iters, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7, x8 = x0 + 8;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    if (MODE == SEL_VCC_VALU) {
      asm volatile(
          "v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %9, vcc\n v_cmp_lt_f32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %9, vcc\n"
          "v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %9, vcc\n v_cmp_lt_f32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %9, vcc\n"
          "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %4, %4, %9, vcc\n v_cmp_lt_f32 vcc, %5, %8\n v_cndmask_b32 %5, %5, %9, vcc\n"
          "v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %6, %6, %9, vcc\n v_cmp_lt_f32 vcc, %7, %8\n v_cndmask_b32 %7, %7, %9, vcc\n"
          : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
    } else if (MODE == SEL_VCC_ONCE) {
      asm volatile(
          "v_cmp_lt_f32 vcc, %0, %8\n"
          "v_cndmask_b32 %0, %0, %9, vcc\n v_cndmask_b32 %1, %1, %9, vcc\n v_cndmask_b32 %2, %2, %9, vcc\n v_cndmask_b32 %3, %3, %9, vcc\n"
          "v_cndmask_b32 %4, %4, %9, vcc\n v_cndmask_b32 %5, %5, %9, vcc\n v_cndmask_b32 %6, %6, %9, vcc\n v_cndmask_b32 %7, %7, %9, vcc\n"
          "v_cndmask_b32 %0, %0, %9, vcc\n v_cndmask_b32 %1, %1, %9, vcc\n v_cndmask_b32 %2, %2, %9, vcc\n v_cndmask_b32 %3, %3, %9, vcc\n"
          "v_cndmask_b32 %4, %4, %9, vcc\n v_cndmask_b32 %5, %5, %9, vcc\n v_cndmask_b32 %6, %6, %9, vcc\n v_cndmask_b32 %7, %7, %9, vcc\n"
          : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
    } else if (MODE == SEL_VCC_SALU) {
      asm volatile(
          "v_cmp_lt_f32 s[20:21], %0, %8\n v_cmp_lt_f32 s[22:23], %1, %8\n"
          "s_and_b64 vcc, s[20:21], s[22:23]\n v_cndmask_b32 %0, %0, %9, vcc\n s_and_b64 vcc, s[20:21], s[22:23]\n v_cndmask_b32 %1, %1, %9, vcc\n"
          "s_and_b64 vcc, s[20:21], s[22:23]\n v_cndmask_b32 %2, %2, %9, vcc\n s_and_b64 vcc, s[20:21], s[22:23]\n v_cndmask_b32 %3, %3, %9, vcc\n"
          "s_and_b64 vcc, s[20:21], s[22:23]\n v_cndmask_b32 %4, %4, %9, vcc\n s_and_b64 vcc, s[20:21], s[22:23]\n v_cndmask_b32 %5, %5, %9, vcc\n"
          : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b)
          : "vcc", "scc", "s20", "s21", "s22", "s23");
    } else if (MODE == SEL_SGPR_SALU) {
      asm volatile(
          "v_cmp_lt_f32 s[20:21], %0, %8\n v_cmp_lt_f32 s[22:23], %1, %8\n"
          "s_and_b64 s[24:25], s[20:21], s[22:23]\n v_cndmask_b32_e64 %0, %0, %9, s[24:25]\n s_and_b64 s[26:27], s[20:21], s[22:23]\n v_cndmask_b32_e64 %1, %1, %9, s[26:27]\n"
          "s_and_b64 s[24:25], s[20:21], s[22:23]\n v_cndmask_b32_e64 %2, %2, %9, s[24:25]\n s_and_b64 s[26:27], s[20:21], s[22:23]\n v_cndmask_b32_e64 %3, %3, %9, s[26:27]\n"
          "s_and_b64 s[24:25], s[20:21], s[22:23]\n v_cndmask_b32_e64 %4, %4, %9, s[24:25]\n s_and_b64 s[26:27], s[20:21], s[22:23]\n v_cndmask_b32_e64 %5, %5, %9, s[26:27]\n"
          : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b)
          : "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
    } else if (MODE == SEL_SGPR_VALU) {
      asm volatile(
          "v_cmp_lt_f32 s[20:21], %0, %8\n v_cndmask_b32_e64 %0, %0, %9, s[20:21]\n v_cmp_lt_f32 s[22:23], %1, %8\n v_cndmask_b32_e64 %1, %1, %9, s[22:23]\n"
          "v_cmp_lt_f32 s[24:25], %2, %8\n v_cndmask_b32_e64 %2, %2, %9, s[24:25]\n v_cmp_lt_f32 s[26:27], %3, %8\n v_cndmask_b32_e64 %3, %3, %9, s[26:27]\n"
          "v_cmp_lt_f32 s[20:21], %4, %8\n v_cndmask_b32_e64 %4, %4, %9, s[20:21]\n v_cmp_lt_f32 s[22:23], %5, %8\n v_cndmask_b32_e64 %5, %5, %9, s[22:23]\n"
          "v_cmp_lt_f32 s[24:25], %6, %8\n v_cndmask_b32_e64 %6, %6, %9, s[24:25]\n v_cmp_lt_f32 s[26:27], %7, %8\n v_cndmask_b32_e64 %7, %7, %9, s[26:27]\n"
          : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b)
          : "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
    } else if (MODE == EXEC_REGION) {
      asm volatile(REP4(
          "v_cmp_lt_f32 vcc, %8, %0\n s_and_saveexec_b64 s[20:21], vcc\n"
          "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n"
          "s_mov_b64 exec, s[20:21]\n")
          : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc", "scc", "s20", "s21");
    } else if (MODE == CMP2_AND_SEL_VCC) {
      asm volatile(REP4(
          "v_cmp_le_u32 s[20:21], %0, %8\n v_cmp_le_f32 s[22:23], %9, %1\n s_and_b64 vcc, s[20:21], s[22:23]\n v_cndmask_b32 %1, 0, %1, vcc\n")
          : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b)
          : "vcc", "scc", "s20", "s21", "s22", "s23");
    } else if (MODE == CMP2_AND_SEL_SGPR) {
      asm volatile(REP4(
          "v_cmp_le_u32 s[20:21], %0, %8\n v_cmp_le_f32 s[22:23], %9, %1\n s_and_b64 s[24:25], s[20:21], s[22:23]\n v_cndmask_b32_e64 %1, 0, %1, s[24:25]\n")
          : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b)
          : "scc", "s20", "s21", "s22", "s23", "s24", "s25");
    } else if (MODE == MUL_MASK) {
      asm volatile(
          "v_cmp_lt_f32 vcc, %0, %8\n v_mul_f32 %0, %0, %9\n v_cmp_lt_f32 vcc, %1, %8\n v_mul_f32 %1, %1, %9\n"
          "v_cmp_lt_f32 vcc, %2, %8\n v_mul_f32 %2, %2, %9\n v_cmp_lt_f32 vcc, %3, %8\n v_mul_f32 %3, %3, %9\n"
          "v_cmp_lt_f32 vcc, %4, %8\n v_mul_f32 %4, %4, %9\n v_cmp_lt_f32 vcc, %5, %8\n v_mul_f32 %5, %5, %9\n"
          "v_cmp_lt_f32 vcc, %6, %8\n v_mul_f32 %6, %6, %9\n v_cmp_lt_f32 vcc, %7, %8\n v_mul_f32 %7, %7, %9\n"
          : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
    } else if (MODE == DPP_FOLD9) {
      asm volatile(
          "s_nop 1\n"
          "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
          "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
          "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
          "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
          "v_add_f32_dpp %8, %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
          "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
          "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
          "v_add_f32_dpp %4, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
          "v_add_f32_dpp %6, %6, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
          "v_add_f32_dpp %8, %8, %8 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
          : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+v"(x8));
    } else if (MODE == DPP_MOV_ADD) {
      float t;
      asm volatile(REP4(
          "v_mov_b32_dpp %9, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32 %0, %0, %9\n"
          "v_mov_b32_dpp %9, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32 %1, %1, %9\n"
          "v_mov_b32_dpp %9, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_add_f32 %2, %2, %9\n"
          "v_mov_b32_dpp %9, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_add_f32 %3, %3, %9\n")
          "v_mov_b32_dpp %9, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32 %4, %4, %9\n"
          "v_mov_b32_dpp %9, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_add_f32 %5, %5, %9\n"
          : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+v"(x8), "=&v"(t));
    } else if (MODE == FMA_REF) {
      asm volatile(
          "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
          "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
          "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
          "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
          : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + x8;
}

template <int MODE>
void run(int waves_per_simd) {
  const int cus = 256, iters = 20000;
  const int blocks = cus * waves_per_simd;
  float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
  unsigned long long* clk; hipMalloc(&clk, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) k<MODE><<<blocks, 256>>>(out, clk, iters, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(out, clk, iters, 1.0001f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double ghz = (double)h[0] / ((double)h[1] * 10.0);
  const double n = (double)iters * insts[MODE] * waves_per_simd;
  printf("%-58s waves/SIMD=%d  %8.3f ms  clock %.2f GHz  %6.2f cycles per VALU instruction per SIMD  (%.1f per iteration)\n",
         names[MODE], waves_per_simd, ms, ghz, ms * 1e-3 * ghz * 1e9 / n, ms * 1e-3 * ghz * 1e9 / ((double)iters * waves_per_simd));
  hipFree(out); hipFree(clk);
}
int main() {
  for (int w : {1, 3, 8}) {
    run<FMA_REF>(w); run<SEL_VCC_VALU>(w); run<SEL_VCC_ONCE>(w); run<SEL_VCC_SALU>(w); run<SEL_SGPR_SALU>(w); run<SEL_SGPR_VALU>(w);
    run<EXEC_REGION>(w); run<CMP2_AND_SEL_VCC>(w); run<CMP2_AND_SEL_SGPR>(w); run<MUL_MASK>(w); run<DPP_FOLD9>(w); run<DPP_MOV_ADD>(w);
  }
  return 0;
}
