// Densification bookkeeping: scene/gaussian_model.py:750-772 densify_and_prune with its clone
// (:580-610), split (:506-578), postfix (:466-504) and prune (:401-449) steps as one plan + row-gather passes.
//
// The reference builds the new model through ~60 boolean-mask indexing / cat / repeat launches (each mask index is a
// nonzero + gather with a host sync) and touches every parameter and Adam moment three times.  Here one kernel
// classifies every Gaussian, four block scans place the survivors, and each tensor is read once and written once:
//
//   out = [ kept originals | kept clones | kept first children | kept second children ]      (the reference's order)
//
//   plan      : per Gaussian  grad = accum / denom (NaN -> 0);  sel = grad >= thr;  big = max(exp(scaling)) > pd*extent
//               clone = sel & !big; split = sel & big;  prune = sigmoid(opacity) < min_opacity | max scale > 0.1*extent
//               (the screen-size test of :762 always reads the zeros that densification_postfix has just stored, :503)
//   positions : exclusive scans (block totals -> block offsets -> in-block) of the four flag classes
//   gather    : dst row <- src row for kept / clone / child rows (moments: zeros for the new rows, :458-459)
//   children  : xyz = R(rotation) (exp(scaling) * noise) + xyz ;  scaling = log(exp(scaling) / (0.8 * 2))
#include "gsr_common.h"
#include "gsr_launch.h"

namespace gsr {

constexpr int DN_BLOCK = PRE_BLOCK;
enum : uint8_t { DN_KEEP = 1, DN_CLONE = 2, DN_CHILD = 4, DN_SPLIT_SEL = 8 };

__global__ __launch_bounds__(DN_BLOCK) void densify_plan_kernel(int P, const float* __restrict__ accum,
                                                                 const float* __restrict__ denom,
                                                                 const float* __restrict__ scaling,
                                                                 const float* __restrict__ opacity, float thr, float pde,
                                                                 float min_opacity, float ws_limit, int use_ws,
                                                                 uint8_t* __restrict__ flags,
                                                                 uint32_t* __restrict__ block_counts, int nblocks) {
  __shared__ uint32_t cnt[4];
  if (threadIdx.x < 4) cnt[threadIdx.x] = 0;
  __syncthreads();
  const int i = blockIdx.x * DN_BLOCK + threadIdx.x;
  uint8_t f = 0;
  if (i < P) {
    float g = accum[i] / denom[i];
    if (g != g) g = 0.0f;
    const float s0 = expf(scaling[3 * i]), s1 = expf(scaling[3 * i + 1]), s2 = expf(scaling[3 * i + 2]);
    const float mx = fmaxf(fmaxf(s0, s1), s2);
    const bool sel = g >= thr;
    const bool clone = sel && mx <= pde, split = sel && mx > pde;
    const float op = 1.0f / (1.0f + expf(-opacity[i]));
    const bool low = op < min_opacity;
    const bool prune_o = low || (use_ws && mx > ws_limit);
    // the children carry log(scale / 1.6); the prune test sees exp() of that (:548, :760-763)
    const float c0 = expf(logf(s0 / 1.6f)), c1 = expf(logf(s1 / 1.6f)), c2 = expf(logf(s2 / 1.6f));
    const bool prune_c = low || (use_ws && fmaxf(fmaxf(c0, c1), c2) > ws_limit);
    if (!split && !prune_o) f |= DN_KEEP;
    if (clone && !prune_o) f |= DN_CLONE;
    if (split && !prune_c) f |= DN_CHILD;
    if (split) f |= DN_SPLIT_SEL;
    flags[i] = f;
  }
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const uint64_t m = __builtin_amdgcn_ballot_w64((f >> b) & 1);
    if ((threadIdx.x & (WAVE - 1)) == 0 && m) atomicAdd(&cnt[b], (uint32_t)__builtin_popcountll(m));
  }
  __syncthreads();
  if (threadIdx.x < 4) block_counts[threadIdx.x * (nblocks + 1) + blockIdx.x] = cnt[threadIdx.x];
}

// pos[c][i] = output row of Gaussian i in class c (keep / clone / child), or -1; sel_rank[i] = rank among the
// split-selected (the row of its first noise sample), or -1.
__global__ __launch_bounds__(DN_BLOCK) void densify_positions_kernel(int P, const uint8_t* __restrict__ flags,
                                                                      const uint32_t* __restrict__ block_offs, int nblocks,
                                                                      int32_t* __restrict__ pos) {
  __shared__ uint32_t wave_tot[4][DN_BLOCK / WAVE];
  const int i = blockIdx.x * DN_BLOCK + threadIdx.x;
  const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
  const uint8_t f = i < P ? flags[i] : 0;
  uint32_t rank[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const uint64_t m = __builtin_amdgcn_ballot_w64((f >> b) & 1);
    rank[b] = (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[b][wid] = (uint32_t)__builtin_popcountll(m);
  }
  __syncthreads();
  if (i >= P) return;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    uint32_t base = block_offs[b * (nblocks + 1) + blockIdx.x];
    for (int w = 0; w < wid; ++w) base += wave_tot[b][w];
    pos[(size_t)b * P + i] = ((f >> b) & 1) ? (int32_t)(base + rank[b]) : -1;
  }
}

// One element of one row per thread; rows are `w` floats wide.  zero_new: the appended rows receive zeros (Adam
// moments of new points, :458-459) instead of copies.
__global__ void densify_gather_rows_kernel(size_t total, int w, int P, const float* __restrict__ src,
                                           const int32_t* __restrict__ pos, uint32_t n_keep, uint32_t n_clone,
                                           uint32_t n_child, int zero_new, float* __restrict__ dst) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const uint32_t i = (uint32_t)(e / (size_t)w), c = (uint32_t)(e - (size_t)i * w);
  const int32_t pk = pos[i], pc = pos[(size_t)P + i], ph = pos[2 * (size_t)P + i];
  if (pk < 0 && pc < 0 && ph < 0) return;
  const float v = src[e];
  const float nv = zero_new ? 0.0f : v;
  if (pk >= 0) dst[(size_t)pk * w + c] = v;
  if (pc >= 0) dst[((size_t)n_keep + pc) * w + c] = nv;
  if (ph >= 0) {
    dst[((size_t)n_keep + n_clone + ph) * w + c] = nv;
    dst[((size_t)n_keep + n_clone + n_child + ph) * w + c] = nv;
  }
}

__global__ void densify_split_children_kernel(int P, const float* __restrict__ xyz, const float* __restrict__ scaling,
                                              const float* __restrict__ rotation, const float* __restrict__ noise,
                                              const int32_t* __restrict__ pos, uint32_t n_keep, uint32_t n_clone,
                                              uint32_t n_child, uint32_t n_sel, float* __restrict__ dst_xyz,
                                              float* __restrict__ dst_scaling) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const int32_t ph = pos[2 * (size_t)P + i];
  if (ph < 0) return;
  const uint32_t r = (uint32_t)pos[3 * (size_t)P + i];
  const float qr = rotation[4 * i], qx = rotation[4 * i + 1], qy = rotation[4 * i + 2], qz = rotation[4 * i + 3];
  const float norm = sqrtf(qr * qr + qx * qx + qy * qy + qz * qz);          // utils/general_utils.py:78-99
  const float w = qr / norm, x = qx / norm, y = qy / norm, z = qz / norm;
  const float R00 = 1.f - 2.f * (y * y + z * z), R01 = 2.f * (x * y - w * z), R02 = 2.f * (x * z + w * y);
  const float R10 = 2.f * (x * y + w * z), R11 = 1.f - 2.f * (x * x + z * z), R12 = 2.f * (y * z - w * x);
  const float R20 = 2.f * (x * z - w * y), R21 = 2.f * (y * z + w * x), R22 = 1.f - 2.f * (x * x + y * y);
  const float s0 = expf(scaling[3 * i]), s1 = expf(scaling[3 * i + 1]), s2 = expf(scaling[3 * i + 2]);
  const float px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
  const float l0 = logf(s0 / 1.6f), l1 = logf(s1 / 1.6f), l2 = logf(s2 / 1.6f);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float* z3 = noise + 3 * ((size_t)k * n_sel + r);
    const float a = s0 * z3[0], b = s1 * z3[1], c = s2 * z3[2];
    const size_t o = (size_t)n_keep + n_clone + (size_t)k * n_child + ph;
    dst_xyz[3 * o] = (R00 * a + R01 * b + R02 * c) + px;
    dst_xyz[3 * o + 1] = (R10 * a + R11 * b + R12 * c) + py;
    dst_xyz[3 * o + 2] = (R20 * a + R21 * b + R22 * c) + pz;
    dst_scaling[3 * o] = l0; dst_scaling[3 * o + 1] = l1; dst_scaling[3 * o + 2] = l2;
  }
}

// ---- host side ------------------------------------------------------------------------------------------------
DensifyLayout::DensifyLayout(int P) {
  nblocks = (P + DN_BLOCK - 1) / DN_BLOCK;
  if (nblocks < 1) nblocks = 1;
  size_t o = 0;
  flags = o;        o = align_up(o + (size_t)(P > 0 ? P : 1), 256);
  block_counts = o; o = align_up(o + 4 * 4 * (size_t)(nblocks + 1), 256);
  block_offs = o;   o = align_up(o + 4 * 4 * (size_t)(nblocks + 1), 256);
  totals = o;       o = align_up(o + 64, 256);
  pos = o;          o = align_up(o + 4 * 4 * (size_t)(P > 0 ? P : 1), 256);
  bytes = o;
}

void launch_densify_plan(int P, const float* accum, const float* denom, const float* scaling, const float* opacity,
                         float thr, float pde, float min_opacity, float ws_limit, int use_ws, void* ws, hipStream_t s) {
  const DensifyLayout L(P);
  char* base = static_cast<char*>(ws);
  uint8_t* flags = reinterpret_cast<uint8_t*>(base + L.flags);
  uint32_t* counts = reinterpret_cast<uint32_t*>(base + L.block_counts);
  uint32_t* offs = reinterpret_cast<uint32_t*>(base + L.block_offs);
  uint32_t* totals = reinterpret_cast<uint32_t*>(base + L.totals);
  int32_t* pos = reinterpret_cast<int32_t*>(base + L.pos);
  const int nb = L.nblocks, stride = nb + 1;
  hipLaunchKernelGGL(densify_plan_kernel, dim3(nb), dim3(DN_BLOCK), 0, s, P, accum, denom, scaling, opacity, thr, pde,
                     min_opacity, ws_limit, use_ws, flags, counts, nb);
  launch_scan_block_sums(counts, offs, totals, counts + stride, offs + stride, totals + 1, nb, s);
  launch_scan_block_sums(counts + 2 * stride, offs + 2 * stride, totals + 2, counts + 3 * stride, offs + 3 * stride,
                         totals + 3, nb, s);
  hipLaunchKernelGGL(densify_positions_kernel, dim3(nb), dim3(DN_BLOCK), 0, s, P, flags, offs, nb, pos);
}

void launch_densify_gather_rows(int P, int w, const float* src, const void* ws, const uint32_t counts[4], int zero_new,
                                float* dst, hipStream_t s) {
  const DensifyLayout L(P);
  const int32_t* pos = reinterpret_cast<const int32_t*>(static_cast<const char*>(ws) + L.pos);
  const size_t total = (size_t)P * w;
  if (total == 0) return;
  hipLaunchKernelGGL(densify_gather_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, total, w, P,
                     src, pos, counts[0], counts[1], counts[2], zero_new, dst);
}

void launch_densify_split_children(int P, const float* xyz, const float* scaling, const float* rotation,
                                   const float* noise, const void* ws, const uint32_t counts[4], float* dst_xyz,
                                   float* dst_scaling, hipStream_t s) {
  const DensifyLayout L(P);
  const int32_t* pos = reinterpret_cast<const int32_t*>(static_cast<const char*>(ws) + L.pos);
  if (P == 0 || counts[2] == 0) return;
  hipLaunchKernelGGL(densify_split_children_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, xyz, scaling, rotation,
                     noise, pos, counts[0], counts[1], counts[2], counts[3], dst_xyz, dst_scaling);
}

}  // namespace gsr
