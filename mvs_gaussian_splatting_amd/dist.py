"""View-sharded multi-GPU step.  The path shards by independent views: every rank holds the
full (replicated, read-only) Gaussian parameters and renders views {rank, rank + N, ...}; there is no
Gaussian- or pixel-level exchange.  The only collective is one ``all_reduce(SUM)`` per step on a 4-float
vector ``[loss_sum, l1_sum, n_views, 0]`` (RCCL over xGMI with backend "nccl"; 16 bytes, latency-bound).

One process per GPU, launched with ``python -m torch.distributed.run`` (RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* from the environment).  ``render_fn`` / ``loss_fn`` are injectable so that the driver logic can be
exercised on CPU ranks with the gloo backend (the tests inject the CPU oracle there; the product path
always uses the HIP operator).
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Sequence

import torch
import torch.distributed as dist


def views_of_rank(n_views: int, rank: int, world: int) -> List[int]:
    """Static round-robin partition of the views: rank r renders r, r + N, r + 2N, ..."""
    return list(range(rank, n_views, world))


def init_process_group(backend: Optional[str] = None, device: Optional[torch.device] = None) -> None:
    """Initialise the default group from the torchrun environment (no-op for a single process)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 or dist.is_initialized():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend=backend, **kw)


def sharded_train_step(model, cameras: Sequence, targets: Sequence[torch.Tensor], bg: torch.Tensor, pipe,
                       render_fn: Optional[Callable] = None, loss_fn: Optional[Callable] = None,
                       stats_fn: Optional[Callable] = None, group=None) -> dict:
    """One step over all views: each rank runs forward + loss + backward (+ densification statistics) on its
    share, then the loss vector is all-reduced.  Gradients stay local (the reference has no gradient
    exchange).  Returns the global mean loss and bookkeeping."""
    if render_fn is None or loss_fn is None:
        from . import render as _render, l1_loss as _l1, add_densification_stats as _stats
        render_fn = render_fn or _render
        loss_fn = loss_fn or _l1
        stats_fn = stats_fn or _stats
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = views_of_rank(len(cameras), rank, world)
    dev = bg.device
    vec = torch.zeros(4, dtype=torch.float32, device=dev)
    pixels = 0
    for v in mine:
        pkg = render_fn(cameras[v], model, pipe, bg)
        loss = loss_fn(pkg["render"], targets[v])
        loss.backward()
        if stats_fn is not None:
            stats_fn(model, pkg["viewspace_points"], pkg["radii"])
        vec[0] += loss.detach()
        vec[1] += loss.detach()
        vec[2] += 1.0
        pixels += int(pkg["render"].shape[-1] * pkg["render"].shape[-2])
    if world > 1:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=group)
    n = max(float(vec[2].item()), 1.0)
    return {"loss": float(vec[0].item()) / n, "views": int(vec[2].item()), "local_views": mine,
            "local_pixels": pixels, "world": world, "rank": rank}
