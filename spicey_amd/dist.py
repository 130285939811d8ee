"""Instance sharding across the GPUs of one node.

The transient of one instance is a serial recurrence, so the only parallel axis is INSTANCES: they
are block-partitioned over ranks (one process per GPU, `torch.distributed`; backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in CPU tests) and every rank runs its shard with no data-path collective.
Collectives are used only around the run: broadcast of what all ranks share (source table, topology
header) from rank 0, and gathers of per-rank results / checksums / timings.
"""
from __future__ import annotations

import os
from typing import List, Optional

import numpy as np
import torch
import torch.distributed as dist

from .launch import free_port, spawn_local_ranks  # noqa: F401  (re-exported; stdlib-only module, usable before any GPU call)


def world() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def init(backend: str, device: Optional[torch.device] = None) -> None:
    """Join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun contract)."""
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1 or dist.is_initialized():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, **kw)


def shard_range(n_total: int, r: Optional[int] = None, w: Optional[int] = None) -> range:
    """Block partition: instance i belongs to rank floor(i * world / n_total)."""
    r = rank() if r is None else r
    w = world() if w is None else w
    lo = -(-r * n_total // w)
    hi = -(-(r + 1) * n_total // w)
    return range(lo, hi)


def broadcast_f64(arr: Optional[np.ndarray], shape_hint=None, device: torch.device = torch.device("cpu"), src: int = 0) -> torch.Tensor:
    """Broadcast a float64 array from `src` (other ranks pass None); returns a tensor on `device`."""
    w = world()
    if w == 1:
        return torch.as_tensor(np.ascontiguousarray(arr, dtype=np.float64), device=device)
    hdr = torch.zeros(4, dtype=torch.int64, device=device)
    if rank() == src:
        a = np.ascontiguousarray(arr, dtype=np.float64)
        hdr[0] = a.ndim
        for i, s in enumerate(a.shape):
            hdr[1 + i] = s
    dist.broadcast(hdr, src)
    shape = tuple(int(hdr[1 + i].item()) for i in range(int(hdr[0].item())))
    t = torch.as_tensor(a, device=device) if rank() == src else torch.empty(shape, dtype=torch.float64, device=device)
    dist.broadcast(t, src)
    return t


def max_over_ranks(x: float, device: torch.device = torch.device("cpu")) -> float:
    t = torch.tensor([x], dtype=torch.float64, device=device)
    if world() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(x: float, device: torch.device = torch.device("cpu")) -> float:
    t = torch.tensor([x], dtype=torch.float64, device=device)
    if world() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_to_all(t: torch.Tensor) -> List[torch.Tensor]:
    """all_gather of equally-shaped per-rank tensors (per-rank checksums, probe results, ...)."""
    if world() == 1:
        return [t]
    out = [torch.empty_like(t) for _ in range(world())]
    dist.all_gather(out, t.contiguous())
    return out


def barrier() -> None:
    if world() > 1:
        dist.barrier()


def gather_rows_to_root(t: torch.Tensor, n_total: int, dst: int = 0) -> Optional[torch.Tensor]:
    """Result gather of the sharded batch: every rank holds the rows (instances) of its block of
    `shard_range(n_total)`, `dst` receives all of them in instance order, [n_total, ...]; other ranks get None.
    Each peer's block travels over its own link to the root (RCCL gather = grouped send/recv; per-link bound on xGMI).
    Blocks of a block partition differ by at most one row: they are padded to the largest for the collective."""
    w = world()
    if w == 1:
        return t
    sizes = [len(shard_range(n_total, r, w)) for r in range(w)]
    mx = max(sizes)
    t = t.contiguous()
    if t.shape[0] != sizes[rank()]:
        raise ValueError(f"rank {rank()} holds {t.shape[0]} rows, its shard has {sizes[rank()]}")
    if t.shape[0] < mx:
        pad = torch.zeros((mx - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        t = torch.cat([t, pad], 0)
    if rank() == dst:
        parts = [torch.empty_like(t) for _ in range(w)]
        dist.gather(t, parts, dst=dst)
        return torch.cat([p[: sizes[r]] for r, p in enumerate(parts)], 0)
    dist.gather(t, None, dst=dst)
    return None
