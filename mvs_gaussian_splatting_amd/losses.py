"""Caller-side loss / statistics steps that sit next to the rasterizer in the train loop, fused into
single HIP kernels behind the C ABI: L1 (``utils/loss_utils.py:17-18``, ``train.py:99``) and the
densification statistics (``scene/gaussian_model.py:775-777``, ``train.py:130``)."""
from __future__ import annotations

import torch

from . import _lib


class _L1Loss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: torch.Tensor, gt: torch.Tensor):
        lib = _lib.load()
        if not x.is_cuda:
            raise _lib.GsrError("l1_loss needs ROCm GPU tensors (no CPU path)")
        xc, gc = x.contiguous(), gt.contiguous()
        if xc.dtype != torch.float32 or gc.dtype != torch.float32 or xc.shape != gc.shape:
            raise TypeError("l1_loss expects two float32 tensors of the same shape")
        n = xc.numel()
        loss_sum = torch.empty(1, dtype=torch.float32, device=x.device)
        ws = torch.empty(lib.gsr_l1_loss_workspace_bytes(), dtype=torch.uint8, device=x.device)   # block partials
        grad = torch.empty_like(xc)
        with torch.cuda.device(x.device):
            stream = torch.cuda.current_stream(x.device).cuda_stream
            _lib.check(lib.gsr_l1_loss_fwd_bwd(xc.data_ptr(), gc.data_ptr(), n, 1.0 / n, loss_sum.data_ptr(),
                                               grad.data_ptr(), ws.data_ptr(), stream), "gsr_l1_loss_fwd_bwd")
        ctx.save_for_backward(grad)
        return (loss_sum / n).reshape(())

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None


def l1_loss(network_output: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    """mean(|x - gt|); the gradient sign(x - gt)/n is produced by the same kernel pass."""
    return _L1Loss.apply(network_output, gt)


class _L1DssimLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: torch.Tensor, gt: torch.Tensor, lambda_dssim: float):
        lib = _lib.load()
        if not x.is_cuda:
            raise _lib.GsrError("l1_dssim_loss needs ROCm GPU tensors (no CPU path)")
        xc, gc = x.contiguous(), gt.contiguous()
        if xc.dtype != torch.float32 or gc.dtype != torch.float32 or xc.shape != gc.shape or xc.dim() != 3:
            raise TypeError("l1_dssim_loss expects two float32 [C,H,W] tensors of the same shape")
        Cn, H, W = (int(v) for v in xc.shape)
        n = xc.numel()
        sums = torch.zeros(2, dtype=torch.float32, device=x.device)
        grad = torch.empty_like(xc)
        ws = torch.empty(lib.gsr_l1_dssim_workspace_bytes(Cn, H, W), dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            stream = torch.cuda.current_stream(x.device).cuda_stream
            _lib.check(lib.gsr_l1_dssim_loss_fwd_bwd(xc.data_ptr(), gc.data_ptr(), Cn, H, W, float(lambda_dssim),
                                                     _lib.DSSIM_ONE_MINUS_MEAN, sums.data_ptr(), grad.data_ptr(), ws.data_ptr(), stream),
                       "gsr_l1_dssim_loss_fwd_bwd")
        ctx.save_for_backward(grad)
        return ((1.0 - lambda_dssim) * sums[0] / n + lambda_dssim * (1.0 - sums[1] / n)).reshape(())

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None


def l1_dssim_loss(network_output: torch.Tensor, gt: torch.Tensor, lambda_dssim: float = 0.2) -> torch.Tensor:
    """The reference's training loss ``(1 - lambda) * l1_loss + lambda * (1 - ssim)`` (``train.py:99-101``,
    ``utils/loss_utils.py:17-63``, ``arguments/__init__.py:96`` lambda_dssim = 0.2) as two fused kernels that
    produce the value and the gradient together."""
    return _L1DssimLoss.apply(network_output, gt, lambda_dssim)


@torch.no_grad()
def add_densification_stats(model, viewspace_point_tensor: torch.Tensor, radii: torch.Tensor) -> None:
    """``train.py:130-131`` in one kernel: for radii > 0 accumulate ||grad.xy||, count, track max radius.
    A frame rendered with ``pipe.fuse_densify_stats`` has had them taken by its backward already."""
    lib = _lib.load()
    grad = viewspace_point_tensor.grad
    if getattr(viewspace_point_tensor, "_gsr_stats_fused", False):
        if grad is None:
            raise ValueError("viewspace_points has no .grad (call backward first)")
        return
    if grad is None:
        raise ValueError("viewspace_points has no .grad (call backward first)")
    if not grad.is_cuda:
        raise _lib.GsrError("add_densification_stats needs ROCm GPU tensors (no CPU path)")
    grad = grad.contiguous()
    P = int(grad.shape[0])
    for t in (model.xyz_gradient_accum, model.denom, model.max_radii2D):
        if not (t.is_contiguous() and t.dtype == torch.float32 and t.numel() == P):
            raise TypeError("accumulators must be contiguous float32 with one element per Gaussian")
    with torch.cuda.device(grad.device):
        stream = torch.cuda.current_stream(grad.device).cuda_stream
        _lib.check(lib.gsr_densify_stats(P, grad.data_ptr(), radii.contiguous().data_ptr(),
                                         model.xyz_gradient_accum.data_ptr(), model.denom.data_ptr(),
                                         model.max_radii2D.data_ptr(), stream), "gsr_densify_stats")
