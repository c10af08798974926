// Fused training loss of the reference's step (train.py:99-101): (1-lambda) * L1 + lambda * (1 - SSIM), forward and
// gradient.  dssim_mode 1 is the 2D script's variant (2d_gaussian_splatting.py:196-202):
// (1-lambda) * L1 + lambda * mean(clamp((1 - SSIM)/2, 0, 1)).  SSIM as utils/loss_utils.py:23-63: 11x11 Gaussian window (sigma 1.5), depthwise,
// zero padding, C1 = 0.01^2, C2 = 0.03^2, mean over all pixels and channels.  The window is separable
// (create_window builds it as an outer product), so every 16x16 output tile stages a 26x26 halo in LDS and
// runs a horizontal then a vertical 11-tap pass.
//   forward : per pixel the five windowed moments -> SSIM, its sum, and the three derivative maps
//             A = dL/dmu1, B = dL/dE[x^2], Cm = dL/dE[xy]  (the loss depends on x only through them)
//   backward: dL/dx = w * A + 2 x (w * B) + y (w * Cm) + (1-lambda) sign(x-y)/n      (w symmetric)
#include "gsr_common.h"
#include "gsr_launch.h"

namespace gsr {

constexpr int SS_T = 16;                 // output tile edge
constexpr int SS_R = 5;                  // window radius (11 taps)
constexpr int SS_H = SS_T + 2 * SS_R;    // halo edge (26)

struct Win11 { float w[11]; };

__device__ inline float halo_load(const float* __restrict__ img, int W, int H, int x, int y) {
  return (x >= 0 && x < W && y >= 0 && y < H) ? img[(size_t)y * W + x] : 0.0f;
}

__global__ __launch_bounds__(SS_T * SS_T) void ssim_fwd_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                                int W, int H, Win11 win, float lambda, float inv_n,
                                                                int dssim_mode, float* __restrict__ partials,
                                                                float* __restrict__ maps) {
  __shared__ float sx[SS_H][SS_H + 1];
  __shared__ float sy[SS_H][SS_H + 1];
  __shared__ float hz[5][SS_H][SS_T + 1];   // horizontally filtered x, y, xx, yy, xy
  __shared__ float red[2][SS_T * SS_T / WAVE];
  const int tx = threadIdx.x & (SS_T - 1), ty = threadIdx.x / SS_T;
  const int x0 = blockIdx.x * SS_T, y0 = blockIdx.y * SS_T;
  const size_t plane = (size_t)W * H;
  const float* __restrict__ xc = X + plane * blockIdx.z;
  const float* __restrict__ yc = Y + plane * blockIdx.z;
  for (int i = threadIdx.x; i < SS_H * SS_H; i += SS_T * SS_T) {
    const int hy = i / SS_H, hx = i - hy * SS_H;
    sx[hy][hx] = halo_load(xc, W, H, x0 + hx - SS_R, y0 + hy - SS_R);
    sy[hy][hx] = halo_load(yc, W, H, x0 + hx - SS_R, y0 + hy - SS_R);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < SS_H * SS_T; i += SS_T * SS_T) {
    const int hy = i / SS_T, ox = i - hy * SS_T;
    float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const float xv = sx[hy][ox + k], yv = sy[hy][ox + k], wk = win.w[k];
      a += wk * xv; b += wk * yv; aa += wk * xv * xv; bb += wk * yv * yv; ab += wk * xv * yv;
    }
    hz[0][hy][ox] = a; hz[1][hy][ox] = b; hz[2][hy][ox] = aa; hz[3][hy][ox] = bb; hz[4][hy][ox] = ab;
  }
  __syncthreads();
  float m1 = 0.f, m2 = 0.f, exx = 0.f, eyy = 0.f, exy = 0.f;
#pragma unroll
  for (int k = 0; k < 11; ++k) {
    const float wk = win.w[k];
    m1 += wk * hz[0][ty + k][tx]; m2 += wk * hz[1][ty + k][tx]; exx += wk * hz[2][ty + k][tx];
    eyy += wk * hz[3][ty + k][tx]; exy += wk * hz[4][ty + k][tx];
  }
  const int px = x0 + tx, py = y0 + ty;
  const bool in = px < W && py < H;
  float s = 0.f, s_acc = 0.f, l1 = 0.f;
  if (in) {
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    const float mu1_sq = m1 * m1, mu2_sq = m2 * m2, mu12 = m1 * m2;
    const float s1 = exx - mu1_sq, s2 = eyy - mu2_sq, s12 = exy - mu12;
    const float N1 = 2.0f * mu12 + C1, N2 = 2.0f * s12 + C2, D1 = mu1_sq + mu2_sq + C1, D2 = s1 + s2 + C2;
    const float inv = 1.0f / (D1 * D2);
    s = N1 * N2 * inv;
    float dLds = -lambda * inv_n;                                     // d/ds of lambda * (1 - mean s)
    if (dssim_mode == GSR_DSSIM_CLAMPED_HALF) {                       // lambda * mean(clamp((1 - s)/2, 0, 1))
      const float d = (1.0f - s) * 0.5f;
      dLds = (d >= 0.0f && d <= 1.0f) ? 0.5f * dLds : 0.0f;
      s_acc = fminf(fmaxf(d, 0.0f), 1.0f);
    } else {
      s_acc = s;
    }
    const float ds_dm1 = 2.0f * m2 * (N2 - N1) * inv - s * 2.0f * m1 * (D2 - D1) * inv;
    const size_t o = plane * blockIdx.z + (size_t)py * W + px;
    const size_t total = plane * gridDim.z;
    maps[o] = dLds * ds_dm1;
    maps[total + o] = dLds * (-s / D2);
    maps[2 * total + o] = dLds * (2.0f * N1 * inv);
    l1 = fabsf(sx[ty + SS_R][tx + SS_R] - sy[ty + SS_R][tx + SS_R]);
  }
  s = wave_reduce_add_f32(s_acc);
  l1 = wave_reduce_add_f32(l1);
  const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
  if (lane == 0) { red[0][wid] = l1; red[1][wid] = s; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int w = 0; w < SS_T * SS_T / WAVE; ++w) { a += red[0][w]; b += red[1][w]; }
    // one partial pair per block, summed in block order by ssim_sum_kernel: tens of thousands of atomics on two
    // addresses serialise (0.6 ms at 1080p) and would make the loss value depend on the arrival order
    const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    partials[2 * blk] = a;
    partials[2 * blk + 1] = b;
  }
}

// sums[0] += sum of the blocks' L1 partials, sums[1] += sum of their SSIM partials (fixed order: deterministic)
__global__ __launch_bounds__(1024) void ssim_sum_kernel(size_t nblocks, const float* __restrict__ partials,
                                                        float* __restrict__ sums) {
  __shared__ float red[2][1024 / WAVE];
  float a = 0.f, b = 0.f;
  for (size_t i = threadIdx.x; i < nblocks; i += 1024) { a += partials[2 * i]; b += partials[2 * i + 1]; }
  a = wave_reduce_add_f32(a);
  b = wave_reduce_add_f32(b);
  const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
  if (lane == 0) { red[0][wid] = a; red[1][wid] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float sa = 0.f, sb = 0.f;
    for (int w = 0; w < 1024 / WAVE; ++w) { sa += red[0][w]; sb += red[1][w]; }
    sums[0] += sa;
    sums[1] += sb;
  }
}

__global__ __launch_bounds__(SS_T * SS_T) void ssim_bwd_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                                int W, int H, Win11 win, float l1_scale,
                                                                const float* __restrict__ maps,
                                                                float* __restrict__ dL_dx) {
  __shared__ float sm[3][SS_H][SS_H + 1];
  __shared__ float hz[3][SS_H][SS_T + 1];
  const int tx = threadIdx.x & (SS_T - 1), ty = threadIdx.x / SS_T;
  const int x0 = blockIdx.x * SS_T, y0 = blockIdx.y * SS_T;
  const size_t plane = (size_t)W * H, total = plane * gridDim.z;
  for (int i = threadIdx.x; i < SS_H * SS_H; i += SS_T * SS_T) {
    const int hy = i / SS_H, hx = i - hy * SS_H;
#pragma unroll
    for (int m = 0; m < 3; ++m)
      sm[m][hy][hx] = halo_load(maps + m * total + plane * blockIdx.z, W, H, x0 + hx - SS_R, y0 + hy - SS_R);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < SS_H * SS_T; i += SS_T * SS_T) {
    const int hy = i / SS_T, ox = i - hy * SS_T;
    float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const float wk = win.w[k];
      a += wk * sm[0][hy][ox + k]; b += wk * sm[1][hy][ox + k]; c += wk * sm[2][hy][ox + k];
    }
    hz[0][hy][ox] = a; hz[1][hy][ox] = b; hz[2][hy][ox] = c;
  }
  __syncthreads();
  const int px = x0 + tx, py = y0 + ty;
  if (px < W && py < H) {
    float ga = 0.f, gb = 0.f, gc = 0.f;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const float wk = win.w[k];
      ga += wk * hz[0][ty + k][tx]; gb += wk * hz[1][ty + k][tx]; gc += wk * hz[2][ty + k][tx];
    }
    const size_t o = plane * blockIdx.z + (size_t)py * W + px;
    const float xv = X[o], yv = Y[o], d = xv - yv;
    const float sg = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
    dL_dx[o] = ga + 2.0f * xv * gb + yv * gc + l1_scale * sg;
  }
}

void launch_l1_dssim(const float* x, const float* gt, int C, int H, int W, float lambda, int dssim_mode, float* sums,
                     float* dL_dx, float* maps, hipStream_t s) {
  Win11 win;
  double g[11], tot = 0.0;
  for (int i = 0; i < 11; ++i) { g[i] = exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)); tot += g[i]; }
  // utils/loss_utils.py:23-25 builds the window in float32: exp() per tap, then a float32 normalisation
  float gf[11], totf = 0.f;
  for (int i = 0; i < 11; ++i) { gf[i] = (float)g[i]; totf += gf[i]; }
  for (int i = 0; i < 11; ++i) win.w[i] = gf[i] / totf;
  (void)tot;
  const dim3 grid((W + SS_T - 1) / SS_T, (H + SS_T - 1) / SS_T, C);
  const float inv_n = 1.0f / ((float)C * (float)H * (float)W);
  float* partials = maps + 3 * (size_t)C * H * W;      // behind the three derivative maps
  const size_t nblocks = (size_t)grid.x * grid.y * grid.z;
  hipLaunchKernelGGL(ssim_fwd_kernel, grid, dim3(SS_T * SS_T), 0, s, x, gt, W, H, win, lambda, inv_n, dssim_mode,
                     partials, maps);
  hipLaunchKernelGGL(ssim_sum_kernel, dim3(1), dim3(1024), 0, s, nblocks, partials, sums);
  hipLaunchKernelGGL(ssim_bwd_kernel, grid, dim3(SS_T * SS_T), 0, s, x, gt, W, H, win, (1.0f - lambda) * inv_n, maps, dL_dx);
}

}  // namespace gsr
