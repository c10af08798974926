"""Drop-in for ``simple_knn._C.distCUDA2`` (``scene/gaussian_model.py:21,210``): mean squared distance of every
point to its three nearest other points, exact, on the GPU (``csrc/knn.hip``)."""
from __future__ import annotations

import torch

from . import _lib


@torch.no_grad()
def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    if not points.is_cuda:
        raise _lib.GsrError("distCUDA2 needs a ROCm GPU tensor (no CPU path)")
    pts = points.detach().float().contiguous()
    if pts.dim() != 2 or pts.shape[1] != 3:
        raise ValueError("points must be [N, 3]")
    N = int(pts.shape[0])
    out = torch.empty(N, dtype=torch.float32, device=pts.device)
    nbytes = lib.gsr_knn3_workspace_bytes(N)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=pts.device)
    with torch.cuda.device(pts.device):
        stream = torch.cuda.current_stream(pts.device).cuda_stream
        _lib.check(lib.gsr_dist2_knn3(pts.data_ptr(), N, out.data_ptr(), ws.data_ptr(), nbytes, stream), "gsr_dist2_knn3")
    return out
