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


# ---- what rank 0 hands out before a run: the shared topology to everybody, each rank's block of the
# ---- per-instance parameters and state to that rank --------------------------------------------------------------------
def _bcast_i32(arr: Optional[np.ndarray], device: torch.device, src: int) -> np.ndarray:
    hdr = torch.zeros(1, dtype=torch.int64, device=device)
    if rank() == src:
        a = np.ascontiguousarray(arr, dtype=np.int32).reshape(-1)
        hdr[0] = a.size
    dist.broadcast(hdr, src)
    n = int(hdr.item())
    t = torch.as_tensor(a, device=device) if rank() == src else torch.empty(n, dtype=torch.int32, device=device)
    if n:
        dist.broadcast(t, src)
    return t.cpu().numpy()


def broadcast_topology(flat, device: torch.device = torch.device("cpu"), src: int = 0):
    """The circuit structure all instances share — node ids of every element class, node count, recorded columns: one int32
    message of 4 x (2 nR + 2 nC + 2 nL + 2 nV + 4 nS + 2 nD) + header bytes (~32 KB on the 1000-node chain) — from rank
    `src` (which passes its FlatCircuit; the others pass None) to every rank.  Returns a FlatCircuit with that topology and
    ZERO instances' worth of values: `scatter_params_from_root` fills them in."""
    from . import abi
    if world() == 1:
        return flat
    keys = abi.FlatCircuit.TOPO
    if rank() == src:
        lens = [len(getattr(flat, k)) for k in keys]
        has_out = flat.out_nodes is not None and len(flat.out_nodes) > 0
        msg = np.concatenate([np.array([flat.n_nodes, int(has_out), len(flat.out_nodes) if has_out else 0] + lens, np.int32)]
                             + [getattr(flat, k) for k in keys] + ([flat.out_nodes] if has_out else []))
    else:
        msg = None
    msg = _bcast_i32(msg, device, src)
    nk = len(keys)
    n_nodes, has_out, n_out = int(msg[0]), int(msg[1]), int(msg[2])
    lens = [int(x) for x in msg[3:3 + nk]]
    off = 3 + nk
    kw = {}
    for k, ln in zip(keys, lens):
        kw[k] = msg[off:off + ln].copy()
        off += ln
    if has_out:
        kw["out_nodes"] = msg[off:off + n_out].copy()
    return abi.FlatCircuit(n_nodes, 1, **kw)


def scatter_params_from_root(topo, full, n_total: int, device: torch.device = torch.device("cpu"), src: int = 0):
    """Per-instance parameters and state of a batch that rank `src` holds (`full`: FlatCircuit with n_total instances; None on
    the other ranks) to the ranks that run them: rank r receives the rows of `shard_range(n_total, r)` of every value array
    (R, C, vPrev, L, iPrev, switch and diode parameters and state: 8 B x (nR + 2 nC + 2 nL + 4 nS + 3 nD) + 4 B x nS per
    instance, 32 KB on the 1000-node chain) in ONE float64 message per rank (`dist.scatter`: the root sends each peer its
    block over that peer's own link).  `topo` is the FlatCircuit of `broadcast_topology`.  Returns this rank's FlatCircuit."""
    from . import abi
    w = world()
    if w == 1:
        return full
    keys = abi.FlatCircuit.VALS + ("S_ison",)
    widths = [getattr(topo, abi.FlatCircuit._KIND[k[0]]) for k in keys]
    row = sum(widths)
    sizes = [len(shard_range(n_total, r, w)) for r in range(w)]
    mx = max(sizes)
    mine = torch.empty((mx, max(row, 1)), dtype=torch.float64, device=device)
    if rank() == src:
        if full.n_inst != n_total:
            raise ValueError(f"the root holds {full.n_inst} instances, n_total says {n_total}")
        rows = np.concatenate([np.asarray(getattr(full, k), np.float64).reshape(n_total, -1) for k in keys], axis=1) if row else np.zeros((n_total, 1))
        parts = []
        for r in range(w):
            blk = np.zeros((mx, max(row, 1)))
            sr = shard_range(n_total, r, w)
            blk[: len(sr)] = rows[sr.start: sr.stop]
            parts.append(torch.as_tensor(blk, device=device))
        dist.scatter(mine, parts, src=src)
    else:
        dist.scatter(mine, None, src=src)
    got = mine.cpu().numpy()[: sizes[rank()]]
    kw = {k: getattr(topo, k) for k in abi.FlatCircuit.TOPO}
    off = 0
    for k, wd in zip(keys, widths):
        blk = got[:, off:off + wd]
        kw[k] = blk.astype(np.int32) if k == "S_ison" else blk
        off += wd
    kw["out_nodes"] = topo.out_nodes
    return abi.FlatCircuit(topo.n_nodes, sizes[rank()], **kw)


def assert_distinct_devices(ident: int, device: torch.device = torch.device("cpu")) -> List[int]:
    """Every rank contributes one integer that identifies the GPU it runs on (bench.py: PCI bus id); all of them must differ —
    N ranks that landed on fewer than N devices would still print a number, of the wrong experiment.  Returns the list."""
    t = torch.tensor([int(ident)], dtype=torch.int64, device=device)
    ids = [int(x.item()) for x in gather_to_all(t)]
    if len(set(ids)) != len(ids):
        raise RuntimeError(f"{len(ids)} ranks share {len(set(ids))} device(s): identifiers {ids}")
    return ids
