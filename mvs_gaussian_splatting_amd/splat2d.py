"""BASELINE config 1: the 2D script's image generator and loss on the HIP path.

Mirrors ``2D-Gaussian-Splatting-main/2d_gaussian_splatting.py``:

* :func:`generate_2D_gaussian_splatting`  (``:44-123``) -- same name, argument order and error behaviour
  (``ValueError`` for a covariance that is not positive definite, ``:59-61``, and for a kernel larger than the
  image, ``:93-94``); the ``N x 3 x H x W`` intermediate of the reference never exists here.
* :func:`combined_loss`  (``:200-202``; ``d_ssim_loss`` ``:196-197``, ``ssim`` ``:160-194``) -- value and gradient
  from the fused L1 + D-SSIM kernels (``GSR_DSSIM_CLAMPED_HALF``).

There is no CPU path: CPU tensors raise ``GsrError``.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib

_AX_CACHE = {}


def _abscissae(K: int, device: torch.device) -> torch.Tensor:
    """``-5 + 10 * linspace(0, 1, K)`` evaluated on the host exactly like the reference (``:66-71``)."""
    key = (K, str(device))
    ax = _AX_CACHE.get(key)
    if ax is None:
        start, end = torch.tensor([-5.0]), torch.tensor([5.0])
        ax = (start + (end - start) * torch.linspace(0, 1, steps=K)).to(torch.float32).to(device).contiguous()
        _AX_CACHE[key] = ax
    return ax


class _Splat2D(torch.autograd.Function):
    @staticmethod
    def forward(ctx, K, sigma_x, sigma_y, rho, coords, colours, H, W):
        lib = _lib.load()
        dev = colours.device
        ins = [t.detach().to(torch.float32).contiguous() for t in (sigma_x, sigma_y, rho, coords, colours)]
        N = int(ins[4].shape[0])
        if ins[0].numel() != N or ins[1].numel() != N or ins[2].numel() != N or tuple(ins[3].shape) != (N, 2) \
                or tuple(ins[4].shape) != (N, 3):
            raise ValueError("expected sigma_x, sigma_y, rho [N], coords [N,2], colours [N,3]")
        ax = _abscissae(K, dev)
        nbytes = lib.gsr_splat2d_workspace_bytes(N, H, W)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        out = torch.empty(3, H, W, dtype=torch.float32, device=dev)
        bad = C.c_int32(0)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(lib.gsr_splat2d_forward(N, K, H, W, *(t.data_ptr() for t in ins), ax.data_ptr(), ws.data_ptr(),
                                               nbytes, out.data_ptr(), C.byref(bad), stream), "gsr_splat2d_forward")
        if bad.value:
            raise ValueError("Covariance matrix must be positive semi-definite")
        ctx.save_for_backward(*ins, ax, ws)
        ctx.dims = (N, K, H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        sx, sy, rho, coords, colours, ax, ws = ctx.saved_tensors
        N, K, H, W = ctx.dims
        g = g.to(torch.float32).contiguous()
        d = [torch.empty_like(t) for t in (sx, sy, rho, coords, colours)]
        with torch.cuda.device(g.device):
            stream = torch.cuda.current_stream(g.device).cuda_stream
            _lib.check(lib.gsr_splat2d_backward(N, K, H, W, sx.data_ptr(), sy.data_ptr(), rho.data_ptr(), ax.data_ptr(),
                                                ws.data_ptr(), ws.numel(), g.data_ptr(), *(t.data_ptr() for t in d),
                                                stream), "gsr_splat2d_backward")
        return (None, *d, None, None)


def generate_2D_gaussian_splatting(kernel_size, sigma_x, sigma_y, rho, coords, colours, image_size=(256, 256, 3),
                                   device=None):
    """-> ``[H, W, 3]`` image in [0, 1] (a channel-last view of a ``[3, H, W]`` buffer, as in the reference)."""
    if not colours.is_cuda:
        raise _lib.GsrError("generate_2D_gaussian_splatting needs ROCm GPU tensors (no CPU path)")
    H, W = int(image_size[0]), int(image_size[1])
    K = int(kernel_size)
    if H - K < 0 or W - K < 0:
        raise ValueError("Kernel size should be smaller or equal to the image size.")
    out = _Splat2D.apply(K, sigma_x.reshape(-1), sigma_y.reshape(-1), rho.reshape(-1), coords, colours, H, W)
    return out.permute(1, 2, 0)


class _CombinedLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred_chw, target_chw, lam):
        lib = _lib.load()
        Cn, H, W = (int(v) for v in pred_chw.shape)
        n = pred_chw.numel()
        dev = pred_chw.device
        sums = torch.zeros(2, dtype=torch.float32, device=dev)
        grad = torch.empty_like(pred_chw)
        ws = torch.empty(lib.gsr_l1_dssim_workspace_bytes(Cn, H, W), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(lib.gsr_l1_dssim_loss_fwd_bwd(pred_chw.data_ptr(), target_chw.data_ptr(), Cn, H, W, float(lam),
                                                     _lib.DSSIM_CLAMPED_HALF, sums.data_ptr(), grad.data_ptr(),
                                                     ws.data_ptr(), stream), "gsr_l1_dssim_loss_fwd_bwd")
        ctx.save_for_backward(grad)
        return ((1.0 - lam) * sums[0] / n + lam * sums[1] / n).reshape(())

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None


def combined_loss(pred, target, lambda_param=0.5):
    """``(1 - lambda) * L1 + lambda * mean(clamp((1 - SSIM)/2, 0, 1))`` on ``[H, W, C]`` images."""
    if not pred.is_cuda:
        raise _lib.GsrError("combined_loss needs ROCm GPU tensors (no CPU path)")
    if pred.shape != target.shape or pred.dim() != 3:
        raise TypeError("combined_loss expects two [H, W, C] tensors of the same shape")
    p = pred.permute(2, 0, 1).to(torch.float32).contiguous()       # no copy for generate_2D_gaussian_splatting's output
    t = target.detach().permute(2, 0, 1).to(torch.float32).contiguous()
    return _CombinedLoss.apply(p, t, float(lambda_param))
