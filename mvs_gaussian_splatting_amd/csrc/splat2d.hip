// BASELINE config 1: the dense, additive 2D Gaussian image of
// 2D-Gaussian-Splatting-main/2d_gaussian_splatting.py:44-123, forward and backward.
//
// The reference tabulates exp(-x^T S^-1 x / 2) on a K x K grid over [-5,5]^2, max-normalises the table, zero-pads
// it to the image and translates it with affine_grid + grid_sample(bilinear, align_corners=True, zeros), multiplies
// by the colour and sums over the Gaussians, then clamps to [0,1].  Bilinear resampling of a table whose entries are
// an analytic function of the grid abscissae is evaluated here directly: every (pixel, Gaussian) pair evaluates its
// four taps, no N x 3 x H x W intermediate exists (the reference's is 197 MB at N = 1000, 128 x 128).
//
//   prepare : one wave per Gaussian -- covariance, inverse, positive-definiteness flag, table maximum (the
//             normaliser and its grid position, which carries the gradient of the max)
//   forward : 16 x 16 pixel tiles x Gaussian chunks -> partial images (deterministic), then a chunk sum + clamp
//   backward: one 256-lane block per Gaussian over its support rectangle, block-reduced -- no atomics
#include "gsr_common.h"
#include "gsr_launch.h"

namespace gsr {

struct __attribute__((aligned(16))) Splat2dRec {
  float i00, i01x2, i11, scale;     // inverse covariance (xx, 2*xy, yy), 1 / (norm * table max)
  float offx, offy, xs, ys;         // translation in pixels (coords * (size-1)/2), abscissae of the table maximum
  float r, g, b, det;
};
static_assert(sizeof(Splat2dRec) == 48, "Splat2dRec");

constexpr int S2_MAX_K = 2048;      // abscissa table held in LDS
constexpr int S2_TILE = 16;
constexpr int S2_BATCH = 64;        // Gaussians staged per round
constexpr float S2_TWO_PI = 6.283185307179586f;

__device__ inline float s2_z(float i00, float i01x2, float i11, float x, float y) {
  // -0.5 * (i00*x*x + 2*i01*x*y + i11*y*y), left to right as the reference evaluates it
  return -0.5f * (((i00 * x) * x + (i01x2 * x) * y) + (i11 * y) * y);
}

__global__ __launch_bounds__(256) void splat2d_prepare_kernel(int N, int K, int H, int W, const float* __restrict__ sx,
                                                               const float* __restrict__ sy,
                                                               const float* __restrict__ rho,
                                                               const float* __restrict__ coords,
                                                               const float* __restrict__ colours,
                                                               const float* __restrict__ ax, Splat2dRec* __restrict__ rec,
                                                               int* __restrict__ bad) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int b = blockIdx.x * (256 / WAVE) + threadIdx.x / WAVE;
  if (b >= N) return;
  const float a = sx[b], c = sy[b], r = rho[b];
  const float c00 = a * a, c01 = (r * a) * c, c11 = c * c;
  const float det = c00 * c11 - c01 * c01;
  const bool ok = det > 0.0f;
  const float i00 = c11 / det, i01 = -c01 / det, i11 = c00 / det;
  const float i01x2 = 2.0f * i01;
  const float norm = S2_TWO_PI * sqrtf(det);
  // table maximum (first index wins among equal values, like a row-major max)
  float best = -1.0f;
  int best_i = 0;
  for (int i = lane; i < K * K; i += WAVE) {
    const int rr = i / K, cc = i - rr * K;
    const float v = __expf(s2_z(i00, i01x2, i11, ax[rr], ax[cc])) / norm;
    if (v > best) { best = v; best_i = i; }
  }
#pragma unroll
  for (int d = WAVE / 2; d > 0; d >>= 1) {
    const float ov = __shfl_xor(best, d, WAVE);
    const int oi = __shfl_xor(best_i, d, WAVE);
    if (ov > best || (ov == best && oi < best_i)) { best = ov; best_i = oi; }
  }
  if (lane == 0) {
    Splat2dRec o;
    o.i00 = i00; o.i01x2 = i01x2; o.i11 = i11;
    o.scale = 1.0f / (norm * best);
    o.offx = coords[2 * b] * ((float)(W - 1) * 0.5f);
    o.offy = coords[2 * b + 1] * ((float)(H - 1) * 0.5f);
    const int rr = best_i / K;
    o.xs = ax[rr]; o.ys = ax[best_i - rr * K];
    o.r = colours[3 * b]; o.g = colours[3 * b + 1]; o.b = colours[3 * b + 2];
    o.det = det;
    rec[b] = o;
    if (!ok || !(best > 0.0f)) atomicOr(bad, 1);
  }
}

// The four bilinear taps of one (pixel, Gaussian) pair.  u, v: table column / row coordinate of the pixel.
struct S2Taps {
  float k00, k01, k10, k11, fu, fv;
  float x0, x1, y0, y1;
};

__device__ inline S2Taps s2_taps(const Splat2dRec& g, const float* __restrict__ ax, int K, float jf, float if_, float left,
                                 float top) {
  S2Taps t;
  const float u = (jf + g.offx) - left, v = (if_ + g.offy) - top;
  const float u0 = floorf(u), v0 = floorf(v);
  t.fu = u - u0; t.fv = v - v0;
  const int c0 = (int)u0, r0 = (int)v0;
  const bool rv0 = r0 >= 0 && r0 < K, rv1 = r0 + 1 >= 0 && r0 + 1 < K;
  const bool cv0 = c0 >= 0 && c0 < K, cv1 = c0 + 1 >= 0 && c0 + 1 < K;
  t.x0 = ax[min(max(r0, 0), K - 1)]; t.x1 = ax[min(max(r0 + 1, 0), K - 1)];
  t.y0 = ax[min(max(c0, 0), K - 1)]; t.y1 = ax[min(max(c0 + 1, 0), K - 1)];
  t.k00 = (rv0 && cv0) ? __expf(s2_z(g.i00, g.i01x2, g.i11, t.x0, t.y0)) * g.scale : 0.0f;
  t.k01 = (rv0 && cv1) ? __expf(s2_z(g.i00, g.i01x2, g.i11, t.x0, t.y1)) * g.scale : 0.0f;
  t.k10 = (rv1 && cv0) ? __expf(s2_z(g.i00, g.i01x2, g.i11, t.x1, t.y0)) * g.scale : 0.0f;
  t.k11 = (rv1 && cv1) ? __expf(s2_z(g.i00, g.i01x2, g.i11, t.x1, t.y1)) * g.scale : 0.0f;
  return t;
}

// block-uniform test: can any pixel of rows [i_lo, i_hi] x columns [j_lo, j_hi] see a tap of this Gaussian?
__device__ inline bool s2_touches(const Splat2dRec& g, int K, float left, float top, int i_lo, int i_hi, int j_lo,
                                  int j_hi) {
  const float v_lo = floorf(((float)i_lo + g.offy) - top), v_hi = floorf(((float)i_hi + g.offy) - top);
  const float u_lo = floorf(((float)j_lo + g.offx) - left), u_hi = floorf(((float)j_hi + g.offx) - left);
  return v_hi >= -1.0f && v_lo <= (float)(K - 1) && u_hi >= -1.0f && u_lo <= (float)(K - 1);
}

__global__ __launch_bounds__(S2_TILE * S2_TILE) void splat2d_fwd_kernel(int N, int K, int H, int W, int per_chunk,
                                                                        const Splat2dRec* __restrict__ rec,
                                                                        const float* __restrict__ ax_g,
                                                                        float* __restrict__ partial) {
  __shared__ float ax[S2_MAX_K];
  __shared__ Splat2dRec st[S2_BATCH];
  const int tx = threadIdx.x & (S2_TILE - 1), ty = threadIdx.x / S2_TILE;
  const int j0 = blockIdx.x * S2_TILE, i0 = blockIdx.y * S2_TILE;
  const int j = j0 + tx, i = i0 + ty;
  const int first = blockIdx.z * per_chunk, last = min(N, first + per_chunk);
  const float left = (float)((W - K) / 2), top = (float)((H - K) / 2);
  for (int k = threadIdx.x; k < K; k += blockDim.x) ax[k] = ax_g[k];
  const int i_hi = min(i0 + S2_TILE, H) - 1, j_hi = min(j0 + S2_TILE, W) - 1;
  float ar = 0.f, ag = 0.f, ab = 0.f;
  for (int base = first; base < last; base += S2_BATCH) {
    __syncthreads();
    const int n = min(S2_BATCH, last - base);
    for (int k = threadIdx.x; k < n * 12; k += blockDim.x)
      reinterpret_cast<float*>(st)[k] = reinterpret_cast<const float*>(rec + base)[k];
    __syncthreads();
    for (int k = 0; k < n; ++k) {
      const Splat2dRec g = st[k];
      if (!s2_touches(g, K, left, top, i0, i_hi, j0, j_hi)) continue;
      const S2Taps t = s2_taps(g, ax, K, (float)j, (float)i, left, top);
      const float wv0 = 1.0f - t.fv, wu0 = 1.0f - t.fu;
      const float s = (((t.k00 * wv0) * wu0 + (t.k01 * wv0) * t.fu) + (t.k10 * t.fv) * wu0) + (t.k11 * t.fv) * t.fu;
      ar += g.r * s; ag += g.g * s; ab += g.b * s;
    }
  }
  if (i < H && j < W) {
    const size_t plane = (size_t)H * W, o = (size_t)i * W + j;
    float* __restrict__ dst = partial + 3 * plane * blockIdx.z;
    dst[o] = ar; dst[plane + o] = ag; dst[2 * plane + o] = ab;
  }
}

__global__ void splat2d_reduce_kernel(size_t n, int chunks, const float* __restrict__ partial, float* __restrict__ pre,
                                      float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int c = 0; c < chunks; ++c) s += partial[(size_t)c * n + i];
  pre[i] = s;
  out[i] = fminf(fmaxf(s, 0.0f), 1.0f);
}

// clamp(x, 0, 1) passes the gradient where 0 <= x <= 1 (torch.clamp's backward)
__global__ void splat2d_mask_grad_kernel(size_t n, const float* __restrict__ pre, const float* __restrict__ dL_dout,
                                         float* __restrict__ gm) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float p = pre[i];
  gm[i] = (p >= 0.0f && p <= 1.0f) ? dL_dout[i] : 0.0f;
}

__global__ __launch_bounds__(256) void splat2d_bwd_kernel(int N, int K, int H, int W, const Splat2dRec* __restrict__ rec,
                                                           const float* __restrict__ ax_g,
                                                           const float* __restrict__ sx, const float* __restrict__ sy,
                                                           const float* __restrict__ rho, const float* __restrict__ gm,
                                                           float* __restrict__ d_sx, float* __restrict__ d_sy,
                                                           float* __restrict__ d_rho, float* __restrict__ d_coords,
                                                           float* __restrict__ d_colours) {
  __shared__ float ax[S2_MAX_K];
  __shared__ float red[8][256 / WAVE];
  const int b = blockIdx.x;
  for (int k = threadIdx.x; k < K; k += blockDim.x) ax[k] = ax_g[k];
  __syncthreads();
  const Splat2dRec g = rec[b];
  const float left = (float)((W - K) / 2), top = (float)((H - K) / 2);
  // support rectangle: pixels whose floor(v) lies in [-1, K-1] (monotone in the pixel index -> scan the ends)
  int i_lo = 0, i_hi = H - 1, j_lo = 0, j_hi = W - 1;
  while (i_lo <= i_hi && floorf(((float)i_lo + g.offy) - top) < -1.0f) ++i_lo;
  while (i_hi >= i_lo && floorf(((float)i_hi + g.offy) - top) > (float)(K - 1)) --i_hi;
  while (j_lo <= j_hi && floorf(((float)j_lo + g.offx) - left) < -1.0f) ++j_lo;
  while (j_hi >= j_lo && floorf(((float)j_hi + g.offx) - left) > (float)(K - 1)) --j_hi;
  const int rw = j_hi - j_lo + 1, rh = i_hi - i_lo + 1;
  const size_t plane = (size_t)H * W;
  float a00 = 0.f, a01 = 0.f, a11 = 0.f, dfu = 0.f, dfv = 0.f, cr = 0.f, cg = 0.f, cb = 0.f;
  const float xs2 = g.xs * g.xs, xys = g.xs * g.ys, ys2 = g.ys * g.ys;
  if (rw > 0 && rh > 0) {
    for (int p = threadIdx.x; p < rw * rh; p += blockDim.x) {
      const int pi = p / rw, i = i_lo + pi, j = j_lo + (p - pi * rw);
      const size_t o = (size_t)i * W + j;
      const float g0 = gm[o], g1 = gm[plane + o], g2 = gm[2 * plane + o];
      const S2Taps t = s2_taps(g, ax, K, (float)j, (float)i, left, top);
      const float wv0 = 1.0f - t.fv, wu0 = 1.0f - t.fu;
      const float w00 = wv0 * wu0 * t.k00, w01 = wv0 * t.fu * t.k01, w10 = t.fv * wu0 * t.k10, w11 = t.fv * t.fu * t.k11;
      const float s = (w00 + w01) + (w10 + w11);
      cr += g0 * s; cg += g1 * s; cb += g2 * s;
      const float G = g0 * g.r + g1 * g.g + g2 * g.b;
      const float x0s = t.x0 * t.x0, x1s = t.x1 * t.x1, y0s = t.y0 * t.y0, y1s = t.y1 * t.y1;
      // d k / d i00 = k * -(x^2 - xs^2)/2 ; d k / d i01 = k * -(x y - xs ys) ; d k / d i11 = k * -(y^2 - ys^2)/2
      a00 += G * (-0.5f) * ((w00 + w01) * (x0s - xs2) + (w10 + w11) * (x1s - xs2));
      a11 += G * (-0.5f) * ((w00 + w10) * (y0s - ys2) + (w01 + w11) * (y1s - ys2));
      a01 -= G * (w00 * (t.x0 * t.y0 - xys) + w01 * (t.x0 * t.y1 - xys) + w10 * (t.x1 * t.y0 - xys) +
                  w11 * (t.x1 * t.y1 - xys));
      dfu += G * (wv0 * (t.k01 - t.k00) + t.fv * (t.k11 - t.k10));
      dfv += G * (wu0 * (t.k10 - t.k00) + t.fu * (t.k11 - t.k01));
    }
  }
  float vals[8] = {a00, a01, a11, dfu, dfv, cr, cg, cb};
  const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const float v = wave_reduce_add_f32(vals[q]);
    if (lane == 0) red[q][wid] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) t[q] = (red[q][0] + red[q][1]) + (red[q][2] + red[q][3]);
    const float a = sx[b], c = sy[b], r = rho[b];
    const float c00 = a * a, c01 = r * a * c, c11 = c * c, D = g.det;
    // i00 = c11/D, i01 = -c01/D, i11 = c00/D
    const float Q = t[0] * c11 - t[1] * c01 + t[2] * c00;
    const float dD = -Q / (D * D);
    const float dc00 = t[2] / D + dD * c11;
    const float dc11 = t[0] / D + dD * c00;
    const float dc01 = -t[1] / D - 2.0f * dD * c01;
    d_sx[b] = 2.0f * a * dc00 + r * c * dc01;
    d_sy[b] = 2.0f * c * dc11 + r * a * dc01;
    d_rho[b] = a * c * dc01;
    d_coords[2 * b] = t[3] * ((float)(W - 1) * 0.5f);
    d_coords[2 * b + 1] = t[4] * ((float)(H - 1) * 0.5f);
    d_colours[3 * b] = t[5]; d_colours[3 * b + 1] = t[6]; d_colours[3 * b + 2] = t[7];
  }
}

// ---- host side ------------------------------------------------------------------------------------------------
Splat2dLayout::Splat2dLayout(int N, int H, int W) {
  const size_t plane3 = 3 * (size_t)H * W;
  const int pix_blocks = ((W + S2_TILE - 1) / S2_TILE) * ((H + S2_TILE - 1) / S2_TILE);
  int c = (2048 + pix_blocks - 1) / pix_blocks;         // enough blocks to fill 256 CUs several times over
  const int max_c = (N + S2_BATCH - 1) / S2_BATCH;
  if (c > max_c) c = max_c;
  if (c > 64) c = 64;
  if (c < 1) c = 1;
  chunks = c;
  per_chunk = N > 0 ? (N + chunks - 1) / chunks : 1;
  size_t o = 0;
  rec = o;     o = align_up(o + sizeof(Splat2dRec) * (size_t)(N > 0 ? N : 1), 256);
  flag = o;    o = align_up(o + 64, 256);
  pre = o;     o = align_up(o + 4 * plane3, 256);
  gmask = o;   o = align_up(o + 4 * plane3, 256);
  partial = o; o = align_up(o + 4 * plane3 * (size_t)chunks, 256);
  bytes = o;
}

void launch_splat2d_fwd(int N, int K, int H, int W, const float* sx, const float* sy, const float* rho,
                        const float* coords, const float* colours, const float* ax, void* ws, float* out,
                        hipStream_t s) {
  const Splat2dLayout L(N, H, W);
  char* base = static_cast<char*>(ws);
  Splat2dRec* rec = reinterpret_cast<Splat2dRec*>(base + L.rec);
  int* flag = reinterpret_cast<int*>(base + L.flag);
  float* pre = reinterpret_cast<float*>(base + L.pre);
  float* partial = reinterpret_cast<float*>(base + L.partial);
  (void)hipMemsetAsync(flag, 0, 4, s);
  if (N > 0)
    hipLaunchKernelGGL(splat2d_prepare_kernel, dim3((N + 3) / 4), dim3(256), 0, s, N, K, H, W, sx, sy, rho, coords,
                       colours, ax, rec, flag);
  const dim3 grid((W + S2_TILE - 1) / S2_TILE, (H + S2_TILE - 1) / S2_TILE, L.chunks);
  hipLaunchKernelGGL(splat2d_fwd_kernel, grid, dim3(S2_TILE * S2_TILE), 0, s, N, K, H, W, L.per_chunk, rec, ax, partial);
  const size_t n = 3 * (size_t)H * W;
  hipLaunchKernelGGL(splat2d_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, L.chunks, partial,
                     pre, out);
}

void launch_splat2d_bwd(int N, int K, int H, int W, const float* sx, const float* sy, const float* rho, const float* ax,
                        void* ws, const float* dL_dout, float* d_sx, float* d_sy, float* d_rho, float* d_coords,
                        float* d_colours, hipStream_t s) {
  const Splat2dLayout L(N, H, W);
  char* base = static_cast<char*>(ws);
  const Splat2dRec* rec = reinterpret_cast<const Splat2dRec*>(base + L.rec);
  const float* pre = reinterpret_cast<const float*>(base + L.pre);
  float* gm = reinterpret_cast<float*>(base + L.gmask);
  const size_t n = 3 * (size_t)H * W;
  hipLaunchKernelGGL(splat2d_mask_grad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, pre, dL_dout, gm);
  if (N > 0)
    hipLaunchKernelGGL(splat2d_bwd_kernel, dim3(N), dim3(256), 0, s, N, K, H, W, rec, ax, sx, sy, rho, gm, d_sx, d_sy,
                       d_rho, d_coords, d_colours);
}

int splat2d_max_kernel_size() { return S2_MAX_K; }

}  // namespace gsr
