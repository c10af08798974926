"""CPU restatement of BASELINE config 1 (plumbing): the dense, additive 2D Gaussian image fit of
``2D-Gaussian-Splatting-main/2d_gaussian_splatting.py`` -- TEST INFRASTRUCTURE ONLY.

Restated functions (reference file:line):
* ``generate_2D_gaussian_splatting``  :44-123   -> :func:`splat2d_ref`
* ``create_window`` / ``ssim`` / ``d_ssim_loss`` / ``combined_loss``  :147-202 -> :func:`combined_loss_ref`

The reference pads each K x K kernel to the image size and translates it with
``affine_grid`` + ``grid_sample(align_corners=True)``; here the same bilinear resampling is written out
with explicit gathers (zero outside the kernel), which is what ``grid_sample`` with zero padding computes.
Pinned by ``tests/golden/splat2d.npz`` (generated from the reference's own function bodies).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def splat2d_ref(kernel_size: int, sigma_x, sigma_y, rho, coords, colours, image_size=(256, 256, 3), ax=None):
    """-> image ``[H, W, 3]`` in [0, 1].  ``coords`` are the normalised translations of the reference
    (``give_required_data``, :242-254).  ``ax`` overrides the kernel-grid abscissae (default: ``-5 + 10 *
    linspace(0, 1, K)`` in the working dtype, :66-71) -- the float64 truth of a float32 run uses the float32 table,
    which is an *input* of the C ABI (``gsr_splat2d_forward``)."""
    B = colours.shape[0]
    K = kernel_size
    H, W = int(image_size[0]), int(image_size[1])
    dt = colours.dtype
    sx, sy, r = sigma_x.view(B, 1, 1), sigma_y.view(B, 1, 1), rho.view(B, 1, 1)
    c00, c01, c11 = sx ** 2, r * sx * sy, sy ** 2
    det = c00 * c11 - c01 * c01
    if (det <= 0).any():
        raise ValueError("Covariance matrix must be positive semi-definite")
    i00, i01, i11 = c11 / det, -c01 / det, c00 / det
    if ax is None:
        ax = -5.0 + 10.0 * torch.linspace(0, 1, steps=K, dtype=dt)
    ax = ax.to(dt)
    xx = ax.view(1, K, 1)          # first coordinate runs along kernel rows (reference meshgrid, :72-76)
    yy = ax.view(1, 1, K)
    z = -0.5 * (i00 * xx * xx + 2.0 * i01 * xx * yy + i11 * yy * yy)
    kernel = torch.exp(z) / (2 * math.pi * torch.sqrt(det))
    kernel = kernel / kernel.amax(dim=(-1, -2), keepdim=True)                  # [B, K, K]

    pad_h, pad_w = H - K, W - K
    if pad_h < 0 or pad_w < 0:
        raise ValueError("Kernel size should be smaller or equal to the image size.")
    left, top = pad_w // 2, pad_h // 2
    # output pixel (i, j) samples the padded kernel at (j + tx*(W-1)/2, i + ty*(H-1)/2)   (align_corners=True)
    jj = torch.arange(W, dtype=dt).view(1, 1, W)
    ii = torch.arange(H, dtype=dt).view(1, H, 1)
    u = jj + coords[:, 0].view(B, 1, 1) * ((W - 1) / 2.0) - left             # kernel column coordinate
    v = ii + coords[:, 1].view(B, 1, 1) * ((H - 1) / 2.0) - top              # kernel row coordinate
    u0, v0 = torch.floor(u), torch.floor(v)
    fu, fv = u - u0, v - v0
    u0, v0 = u0.long(), v0.long()

    def tap(vi, ui):
        ok = (vi >= 0) & (vi < K) & (ui >= 0) & (ui < K)
        flat = (vi.clamp(0, K - 1) * K + ui.clamp(0, K - 1)).expand(B, H, W)
        val = torch.gather(kernel.reshape(B, K * K), 1, flat.reshape(B, H * W)).reshape(B, H, W)
        return torch.where(ok.expand(B, H, W), val, torch.zeros_like(val))

    s = (tap(v0, u0) * (1 - fv) * (1 - fu) + tap(v0, u0 + 1) * (1 - fv) * fu
         + tap(v0 + 1, u0) * fv * (1 - fu) + tap(v0 + 1, u0 + 1) * fv * fu)   # [B, H, W]
    img = torch.einsum("bc,bhw->chw", colours, s)
    return torch.clamp(img, 0, 1).permute(1, 2, 0)


def _window(window_size: int, channel: int, dtype):
    g = torch.exp(torch.tensor([-(x - window_size // 2) ** 2 / float(2 * 1.5 ** 2) for x in range(window_size)]))
    g = (g / g.sum()).unsqueeze(1)
    return g.mm(g.t()).float().unsqueeze(0).unsqueeze(0).expand(channel, 1, window_size, window_size).contiguous().to(dtype)


def dssim_ref(img1, img2, window_size: int = 11):
    """mean(clamp((1 - SSIM)/2, 0, 1)) on HWC images (:160-197)."""
    ch = img1.shape[2]
    a = img1.unsqueeze(0).permute(0, 3, 1, 2)
    b = img2.unsqueeze(0).permute(0, 3, 1, 2)
    w = _window(window_size, ch, a.dtype)
    pad = window_size // 2
    mu1, mu2 = F.conv2d(a, w, padding=pad, groups=ch), F.conv2d(b, w, padding=pad, groups=ch)
    s1 = F.conv2d(a * a, w, padding=pad, groups=ch) - mu1.pow(2)
    s2 = F.conv2d(b * b, w, padding=pad, groups=ch) - mu2.pow(2)
    s12 = F.conv2d(a * b, w, padding=pad, groups=ch) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1.pow(2) + mu2.pow(2) + C1) * (s1 + s2 + C2))
    return torch.clamp((1 - ssim) / 2, 0, 1).mean()


def combined_loss_ref(pred, target, lambda_param: float = 0.5):
    """(1 - lambda) * L1 + lambda * D-SSIM  (:200-202)."""
    return (1 - lambda_param) * (pred - target).abs().mean() + lambda_param * dssim_ref(pred, target)
