// Issue-rate microbenchmark for the compositing kernels' instruction mix on gfx950, at 1..8 waves per SIMD:
// v_fma_f32, v_pk_fma_f32, a dependent fma chain, v_exp_f32, v_cmp+v_cndmask, SALU, ds_read_b128 (wave-uniform
// address = broadcast) and a "visit" mix shaped like one sub-block evaluation of render_fwd.
// Reports wall cycles per wave-instruction per SIMD at the clock measured in-kernel (s_memtime / s_memrealtime).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o gpurun_out/valu_rate && gpurun_out/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

enum { FMA8 = 0, PKFMA4, CHAIN, EXP8, CMPSEL, SALU, LDSB128, VISIT, FMA_SALU, CMP32, CMP64, SEL32, SEL64, MINMAX, FMA_HALF,
       EXP_HALF, READLANE, SAVEEXEC, VMOV, LDS_FMA, CMPX, DPPADD, DPPROW, PERM32, PERM16, RCP, RDFIRST, MED3, BPERM, NMODES };
static const char* names[NMODES] = {"v_fma_f32 x16 (indep)", "v_pk_fma_f32 x16 (indep)", "v_fma_f32 chain x16",
                                    "v_exp_f32 x16 (indep)", "v_cmp+v_cndmask x8 pairs", "s_add_u32 x16 (SALU)",
                                    "ds_read_b128 broadcast x16", "visit mix (12 valu+1 exp)", "8 fma + 8 salu interleaved",
                                    "v_cmp_lt_f32 vcc x16", "v_cmp_lt_f32 -> sgpr pair x16", "v_cndmask_b32 vcc x16",
                                    "v_cndmask_b32 sgpr mask x16", "v_min/v_max x16", "v_fma_f32 x16, EXEC = low 32 lanes",
                                    "v_exp_f32 x16, EXEC = low 32 lanes", "v_readlane_b32 x16", "s_and_saveexec + s_or x8 pairs",
                                    "v_mov_b32 x16", "4 ds_read_b128 bcast + 12 fma", "v_cmpx_lt + restore x8 pairs",
                                    "v_add_f32_dpp quad_perm x16", "v_add_f32_dpp row_mirror / row_ror x16", "v_permlane32_swap x16",
                                    "v_permlane16_swap x16", "v_rcp_f32 x16", "v_readfirstlane_b32 x16", "v_med3_f32 x16",
                                    "ds_bpermute_b32 x16"};
static const int insts[NMODES] = {16, 16, 16, 16, 16, 16, 16, 13, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16};

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* clk, int iters, float a, float b) {
  __shared__ float4 lds[256];
  lds[threadIdx.x] = make_float4(a, b, a, b);
  __syncthreads();
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  float y0 = x0 * 2, y1 = x1 * 2, y2 = x2 * 2, y3 = x3 * 2, y4 = x4 * 2, y5 = x5 * 2, y6 = x6 * 2, y7 = x7 * 2;
  uint32_t s0 = blockIdx.x, s1 = 1;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    if (MODE == FMA8) {
      asm volatile(
          "v_fma_f32 %0, %0, %16, %17\n v_fma_f32 %1, %1, %16, %17\n v_fma_f32 %2, %2, %16, %17\n v_fma_f32 %3, %3, %16, %17\n"
          "v_fma_f32 %4, %4, %16, %17\n v_fma_f32 %5, %5, %16, %17\n v_fma_f32 %6, %6, %16, %17\n v_fma_f32 %7, %7, %16, %17\n"
          "v_fma_f32 %8, %8, %16, %17\n v_fma_f32 %9, %9, %16, %17\n v_fma_f32 %10, %10, %16, %17\n v_fma_f32 %11, %11, %16, %17\n"
          "v_fma_f32 %12, %12, %16, %17\n v_fma_f32 %13, %13, %16, %17\n v_fma_f32 %14, %14, %16, %17\n v_fma_f32 %15, %15, %16, %17\n"
          : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+v"(y0), "+v"(y1), "+v"(y2),
            "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7)
          : "v"(a), "v"(b));
    } else if (MODE == PKFMA4) {
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {y0, y1}, p5 = {y2, y3}, p6 = {y4, y5}, p7 = {y6, y7};
      f2 pa = {a, a}, pb = {b, b};
      asm volatile(
          "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
          "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
          "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
          "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
          : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
          : "v"(pa), "v"(pb));
      x0 = p0.x; x1 = p0.y; x2 = p1.x; x3 = p1.y; x4 = p2.x; x5 = p2.y; x6 = p3.x; x7 = p3.y;
      y0 = p4.x; y1 = p4.y; y2 = p5.x; y3 = p5.y; y4 = p6.x; y5 = p6.y; y6 = p7.x; y7 = p7.y;
    } else if (MODE == CHAIN) {
      asm volatile(REP16("v_fma_f32 %0, %0, %1, %2\n") : "+v"(x0) : "v"(a), "v"(b));
    } else if (MODE == EXP8) {
      asm volatile(
          "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n"
          "v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n v_exp_f32 %8, %8\n v_exp_f32 %9, %9\n v_exp_f32 %10, %10\n v_exp_f32 %11, %11\n"
          "v_exp_f32 %12, %12\n v_exp_f32 %13, %13\n v_exp_f32 %14, %14\n v_exp_f32 %15, %15\n"
          : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+v"(y0), "+v"(y1), "+v"(y2),
            "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7));
    } else if (MODE == CMPSEL) {
      asm volatile(
          "v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %9, vcc\n v_cmp_lt_f32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %9, vcc\n"
          "v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %9, vcc\n v_cmp_lt_f32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %9, vcc\n"
          "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %4, %4, %9, vcc\n v_cmp_lt_f32 vcc, %5, %8\n v_cndmask_b32 %5, %5, %9, vcc\n"
          "v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %6, %6, %9, vcc\n v_cmp_lt_f32 vcc, %7, %8\n v_cndmask_b32 %7, %7, %9, vcc\n"
          : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
          : "v"(a), "v"(b)
          : "vcc");
    } else if (MODE == SALU) {
      asm volatile(REP16("s_add_u32 %0, %0, %1\n") : "+s"(s0) : "s"(s1) : "scc");
    } else if (MODE == LDSB128) {
      float4 r0v, r1v, r2v, r3v;
      const uint32_t addr = (uint32_t)((i & 15) * 16);
      asm volatile(
          "ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4 offset:48\n"
          "ds_read_b128 %0, %4 offset:64\n ds_read_b128 %1, %4 offset:80\n ds_read_b128 %2, %4 offset:96\n ds_read_b128 %3, %4 offset:112\n"
          "ds_read_b128 %0, %4 offset:128\n ds_read_b128 %1, %4 offset:144\n ds_read_b128 %2, %4 offset:160\n ds_read_b128 %3, %4 offset:176\n"
          "ds_read_b128 %0, %4 offset:192\n ds_read_b128 %1, %4 offset:208\n ds_read_b128 %2, %4 offset:224\n ds_read_b128 %3, %4 offset:240\n"
          "s_waitcnt lgkmcnt(0)\n"
          : "=&v"(r0v), "=&v"(r1v), "=&v"(r2v), "=&v"(r3v)
          : "v"(addr));
      x0 += r0v.x + r1v.y + r2v.z + r3v.w;
    } else if (MODE == VISIT) {
      // dx,dy -> quadratic form -> exp2 -> alpha -> tests -> blend (13 instructions, shape of one sub-block evaluation)
      asm volatile(
          "v_sub_f32 %4, %8, %0\n"
          "v_mul_f32 %5, %9, %4\n"
          "v_fma_f32 %5, %8, %1, %5\n"
          "v_mul_f32 %6, %9, %1\n"
          "v_mul_f32 %6, %6, %1\n"
          "v_fma_f32 %5, %4, %5, %6\n"
          "v_exp_f32 %6, %5\n"
          "v_mul_f32 %6, %9, %6\n"
          "v_min_f32 %6, %6, %8\n"
          "v_fma_f32 %7, %2, %6, %2\n"
          "v_cmp_lt_f32 vcc, %7, %8\n"
          "v_fma_f32 %3, %6, %2, %3\n"
          "v_cndmask_b32 %2, %2, %7, vcc\n"
          : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
          : "v"(a), "v"(b)
          : "vcc");
    } else if (MODE == FMA_SALU) {
      asm volatile(
          "v_fma_f32 %0, %0, %9, %10\n s_add_u32 %8, %8, 1\n v_fma_f32 %1, %1, %9, %10\n s_add_u32 %8, %8, 1\n"
          "v_fma_f32 %2, %2, %9, %10\n s_add_u32 %8, %8, 1\n v_fma_f32 %3, %3, %9, %10\n s_add_u32 %8, %8, 1\n"
          "v_fma_f32 %4, %4, %9, %10\n s_add_u32 %8, %8, 1\n v_fma_f32 %5, %5, %9, %10\n s_add_u32 %8, %8, 1\n"
          "v_fma_f32 %6, %6, %9, %10\n s_add_u32 %8, %8, 1\n v_fma_f32 %7, %7, %9, %10\n s_add_u32 %8, %8, 1\n"
          : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+s"(s0)
          : "v"(a), "v"(b)
          : "scc");
    } else if (MODE == CMP32) {
      asm volatile(REP16("v_cmp_lt_f32 vcc, %0, %1\n") : : "v"(x0), "v"(a) : "vcc");
    } else if (MODE == CMP64) {
      asm volatile(REP4("v_cmp_lt_f32 s[20:21], %0, %1\n v_cmp_lt_f32 s[22:23], %0, %1\n v_cmp_lt_f32 s[24:25], %0, %1\n v_cmp_lt_f32 s[26:27], %0, %1\n")
                   : : "v"(x0), "v"(a) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
    } else if (MODE == SEL32) {
      asm volatile(
          "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
          "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
          "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
          "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
          : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a) : "vcc");
    } else if (MODE == SEL64) {
      asm volatile(
          "v_cndmask_b32 %0, %0, %8, s[20:21]\n v_cndmask_b32 %1, %1, %8, s[20:21]\n v_cndmask_b32 %2, %2, %8, s[20:21]\n v_cndmask_b32 %3, %3, %8, s[20:21]\n"
          "v_cndmask_b32 %4, %4, %8, s[20:21]\n v_cndmask_b32 %5, %5, %8, s[20:21]\n v_cndmask_b32 %6, %6, %8, s[20:21]\n v_cndmask_b32 %7, %7, %8, s[20:21]\n"
          "v_cndmask_b32 %0, %0, %8, s[20:21]\n v_cndmask_b32 %1, %1, %8, s[20:21]\n v_cndmask_b32 %2, %2, %8, s[20:21]\n v_cndmask_b32 %3, %3, %8, s[20:21]\n"
          "v_cndmask_b32 %4, %4, %8, s[20:21]\n v_cndmask_b32 %5, %5, %8, s[20:21]\n v_cndmask_b32 %6, %6, %8, s[20:21]\n v_cndmask_b32 %7, %7, %8, s[20:21]\n"
          : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a) : "s20", "s21");
    } else if (MODE == MINMAX) {
      asm volatile(
          "v_min_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n"
          "v_min_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_max_f32 %7, %7, %8\n"
          "v_min_f32 %0, %0, %9\n v_max_f32 %1, %1, %9\n v_min_f32 %2, %2, %9\n v_max_f32 %3, %3, %9\n"
          "v_min_f32 %4, %4, %9\n v_max_f32 %5, %5, %9\n v_min_f32 %6, %6, %9\n v_max_f32 %7, %7, %9\n"
          : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    } else if (MODE == FMA_HALF || MODE == EXP_HALF) {
      if (MODE == FMA_HALF)
        asm volatile(
            "s_mov_b64 s[20:21], exec\n s_mov_b64 exec, 0xffffffff\n"
            "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
            "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
            "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
            "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
            "s_mov_b64 exec, s[20:21]\n"
            : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "s20", "s21");
      else
        asm volatile(
            "s_mov_b64 s[20:21], exec\n s_mov_b64 exec, 0xffffffff\n"
            "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
            "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
            "s_mov_b64 exec, s[20:21]\n"
            : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : : "s20", "s21");
    } else if (MODE == READLANE) {
      asm volatile(REP4("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %0, 5\n v_readlane_b32 s22, %0, 7\n v_readlane_b32 s23, %0, 9\n")
                   : : "v"(x0) : "s20", "s21", "s22", "s23");
    } else if (MODE == SAVEEXEC) {
      asm volatile(REP4("s_and_saveexec_b64 s[20:21], vcc\n s_or_b64 exec, exec, s[20:21]\n s_and_saveexec_b64 s[22:23], vcc\n s_or_b64 exec, exec, s[22:23]\n")
                   : : : "s20", "s21", "s22", "s23");
    } else if (MODE == VMOV) {
      asm volatile(
          "v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n"
          "v_mov_b32 %0, %9\n v_mov_b32 %1, %9\n v_mov_b32 %2, %9\n v_mov_b32 %3, %9\n v_mov_b32 %4, %9\n v_mov_b32 %5, %9\n v_mov_b32 %6, %9\n v_mov_b32 %7, %9\n"
          : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    } else if (MODE == LDS_FMA) {
      float4 r0v, r1v, r2v, r3v;
      const uint32_t addr = (uint32_t)((i & 15) * 16);
      asm volatile(
          "ds_read_b128 %0, %12\n v_fma_f32 %4, %4, %13, %14\n v_fma_f32 %5, %5, %13, %14\n v_fma_f32 %6, %6, %13, %14\n"
          "ds_read_b128 %1, %12 offset:16\n v_fma_f32 %7, %7, %13, %14\n v_fma_f32 %8, %8, %13, %14\n v_fma_f32 %9, %9, %13, %14\n"
          "ds_read_b128 %2, %12 offset:32\n v_fma_f32 %10, %10, %13, %14\n v_fma_f32 %11, %11, %13, %14\n v_fma_f32 %4, %4, %13, %14\n"
          "ds_read_b128 %3, %12 offset:48\n v_fma_f32 %5, %5, %13, %14\n v_fma_f32 %6, %6, %13, %14\n v_fma_f32 %7, %7, %13, %14\n"
          "s_waitcnt lgkmcnt(0)\n"
          : "=&v"(r0v), "=&v"(r1v), "=&v"(r2v), "=&v"(r3v), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
          : "v"(addr), "v"(a), "v"(b));
      y0 += r0v.x + r1v.y + r2v.z + r3v.w;
    } else if (MODE == CMPX) {
      asm volatile(REP4("v_cmpx_lt_f32 vcc, %0, %1\n s_mov_b64 exec, -1\n v_cmpx_lt_f32 vcc, %0, %1\n s_mov_b64 exec, -1\n"
                        "v_cmpx_lt_f32 vcc, %0, %1\n s_mov_b64 exec, -1\n v_cmpx_lt_f32 vcc, %0, %1\n s_mov_b64 exec, -1\n")
                   : : "v"(a), "v"(b) : "vcc");
    } else if (MODE == DPPADD) {
      asm volatile(
          "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_add_f32_dpp %4, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %5, %5, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_add_f32_dpp %6, %6, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %7, %7, %7 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_add_f32_dpp %4, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %5, %5, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_add_f32_dpp %6, %6, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %7, %7, %7 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
    } else if (MODE == DPPROW) {
      asm volatile(
          "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %3, %3 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_add_f32_dpp %4, %4, %4 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %5, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_add_f32_dpp %6, %6, %6 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %7, %7, %7 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %3, %3 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_add_f32_dpp %4, %4, %4 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %5, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_add_f32_dpp %6, %6, %6 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %7, %7, %7 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
    } else if (MODE == PERM32) {
      asm volatile(REP4("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
    } else if (MODE == PERM16) {
      asm volatile(REP4("v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
    } else if (MODE == RCP) {
      asm volatile(
          "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
          "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
          : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
    } else if (MODE == RDFIRST) {
      asm volatile(REP4("v_readfirstlane_b32 s20, %0\n v_readfirstlane_b32 s21, %1\n v_readfirstlane_b32 s22, %2\n v_readfirstlane_b32 s23, %3\n")
                   : : "v"(x0), "v"(x1), "v"(x2), "v"(x3) : "s20", "s21", "s22", "s23");
    } else if (MODE == MED3) {
      asm volatile(
          "v_med3_f32 %0, %0, %8, %9\n v_med3_f32 %1, %1, %8, %9\n v_med3_f32 %2, %2, %8, %9\n v_med3_f32 %3, %3, %8, %9\n"
          "v_med3_f32 %4, %4, %8, %9\n v_med3_f32 %5, %5, %8, %9\n v_med3_f32 %6, %6, %8, %9\n v_med3_f32 %7, %7, %8, %9\n"
          "v_med3_f32 %0, %0, %8, %9\n v_med3_f32 %1, %1, %8, %9\n v_med3_f32 %2, %2, %8, %9\n v_med3_f32 %3, %3, %8, %9\n"
          "v_med3_f32 %4, %4, %8, %9\n v_med3_f32 %5, %5, %8, %9\n v_med3_f32 %6, %6, %8, %9\n v_med3_f32 %7, %7, %8, %9\n"
          : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    } else if (MODE == BPERM) {
      const uint32_t addr = (uint32_t)(((threadIdx.x + 1) & 63) * 4);
      asm volatile(
          REP4("ds_bpermute_b32 %0, %4, %0\n ds_bpermute_b32 %1, %4, %1\n ds_bpermute_b32 %2, %4, %2\n ds_bpermute_b32 %3, %4, %3\n s_waitcnt lgkmcnt(0)\n")
          : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(addr));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + y0 + y1 + y2 + y3 + y4 + y5 + y6 + y7 + (float)s0;
}

template <int MODE>
void run(int waves_per_simd) {
  const int cus = 256, iters = 20000;
  const int blocks = cus * waves_per_simd;      // 256-thread blocks: 4 waves = 1 per SIMD
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
  const double ghz = (double)h[0] / ((double)h[1] * 10.0);       // s_memrealtime ticks at 100 MHz
  const double n = (double)iters * insts[MODE] * waves_per_simd;  // wave-instructions issued on one SIMD
  printf("%-30s waves/SIMD=%d  %8.3f ms  clock %.2f GHz  %.2f cycles per wave-instruction per SIMD\n", names[MODE],
         waves_per_simd, ms, ghz, ms * 1e-3 * ghz * 1e9 / n);
  hipFree(out); hipFree(clk);
}
int main() {
  for (int w : {1, 2, 4, 6, 8}) {
    run<FMA8>(w); run<PKFMA4>(w); run<CHAIN>(w); run<EXP8>(w); run<CMPSEL>(w); run<SALU>(w); run<LDSB128>(w); run<VISIT>(w);
    run<FMA_SALU>(w); run<CMP32>(w); run<CMP64>(w); run<SEL32>(w); run<SEL64>(w); run<MINMAX>(w); run<FMA_HALF>(w);
    run<EXP_HALF>(w); run<READLANE>(w); run<SAVEEXEC>(w); run<VMOV>(w); run<LDS_FMA>(w); run<CMPX>(w);
    run<DPPADD>(w); run<DPPROW>(w); run<PERM32>(w); run<PERM16>(w); run<RCP>(w); run<RDFIRST>(w); run<MED3>(w); run<BPERM>(w);
  }
  return 0;
}
