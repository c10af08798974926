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


class CollectivePlan:
    """Outcome of :func:`negotiate_collectives`, identical on every rank."""
    __slots__ = ("cpu_collectives", "rccl_ranks", "rccl_failed", "backend_note", "why")

    def __init__(self, cpu_collectives, rccl_ranks, rccl_failed, backend_note, why=""):
        self.cpu_collectives, self.rccl_ranks, self.rccl_failed = cpu_collectives, rccl_ranks, rccl_failed
        self.backend_note, self.why = backend_note, why


def negotiate_collectives(world: int, backend: str, probe_device_allreduce: Callable[[], float],
                          agree_min: Callable[[int], int]) -> CollectivePlan:
    """Decide -- THE SAME WAY ON EVERY RANK -- whether the path's one collective (the 16-byte loss all-reduce) runs on
    device tensors over RCCL or has to fall back to host tensors over gloo.

    RCCL communicators are created lazily, at the first collective, and may come up on some ranks and fail on others
    (a missing peer device, a duplicate device in a one-card rehearsal).  If each rank acted on its own outcome the
    ranks would issue different collectives afterwards and hang.  So: every rank probes with an all-reduce of ones
    (``probe_device_allreduce`` returns the reduced value or raises), turns the outcome into a 0/1 flag, and the flags
    are reduced with MIN over a channel that is known to work (``agree_min``: the gloo half of the process group).
    Only a unanimous 1 keeps RCCL.  Pure function of its two callables: unit-tested with fakes
    (tests/test_bench_protocol.py)."""
    if world <= 1:
        return CollectivePlan(False, None, False, backend)
    if backend != "nccl":
        return CollectivePlan(True, None, False, backend)
    ok, why = 1, ""
    try:
        got = float(probe_device_allreduce())
        if int(round(got)) != world:
            ok, why = 0, f"all-reduce of ones returned {got}, expected {world}"
    except Exception as ex:  # noqa: BLE001 -- any failure of the probe means "no RCCL on this rank"
        ok, why = 0, str(ex)[:300]
    agreed = int(agree_min(int(ok)))          # every rank reaches this line, whatever its own outcome
    if agreed == 1:
        return CollectivePlan(False, world, False, "nccl", "")
    return CollectivePlan(True, None, True, "gloo (nccl failed to initialise on at least one rank)",
                          why or "another rank reported a failed probe")


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
