"""Multi-GPU: batch sharding + the one collective the path has (a broadcast of the packed weight tables).

One process per GPU (``torch.distributed``; backend "nccl" is RCCL on ROCm, over xGMI inside a node).  Images are
independent units, so rank r takes a contiguous slice of dim 0 and there is no halo, no reduction and no
data-path collective.  The only exchange is rank 0's packed weight tables (≈18.6 KB for
[*,3,906,438]→[320,196]): a single latency-bound message per (shape, filter) key, after which every rank has the
table in its cache and runs exactly the single-GPU path.  The reference has no distributed layer at all.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Tuple

import torch
import torch.distributed as dist

from . import tables
from .tables import META_LEN, WeightTable


def _collectives_at_world_one() -> bool:
    """A 1-rank process group normally skips its (no-op) collectives.  bench.py --force-launcher sets AA_BENCH_FORCE_DIST=1 so
    that they run anyway: the only way to push the RCCL code path through real hardware on a one-GPU box."""
    import os

    return os.environ.get("AA_BENCH_FORCE_DIST") == "1"


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [start, stop) of ``total`` units for ``rank``; remainders go to the lowest ranks; a rank may
    get an empty range when total < world."""
    if world <= 0 or not (0 <= rank < world) or total < 0:
        raise ValueError("bad shard arguments")
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    stop = start + base + (1 if rank < rem else 0)
    return start, stop


def shard_batch(x: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """This rank's slice of a batch along dim 0 (a view, no copy)."""
    a, b = shard_range(x.shape[0], rank, world)
    return x[a:b]


def broadcast_table(table: Optional[WeightTable], src: int = 0, device: Optional[torch.device] = None,
                    group=None) -> WeightTable:
    """Broadcast one packed table from ``src``.  Two messages: a fixed-length int64 descriptor (so receivers can
    allocate), then the payload bytes.  Works with RCCL (GPU buffers) and gloo (CPU buffers, used by the tests)."""
    rank = dist.get_rank(group)
    backend = dist.get_backend(group)
    comm_dev = torch.device(device) if backend == "nccl" else torch.device("cpu")
    if rank == src:
        if table is None:
            raise ValueError("source rank must pass the table")
        meta = table.meta().to(comm_dev)
    else:
        meta = torch.zeros(META_LEN, dtype=torch.int64, device=comm_dev)
    dist.broadcast(meta, src=src, group=group)
    nbytes = int(meta[9].item())
    if rank == src:
        payload = table.buf.to(comm_dev) if table.buf.device != comm_dev else table.buf
    else:
        payload = torch.empty(nbytes, dtype=torch.uint8, device=comm_dev)
    dist.broadcast(payload, src=src, group=group)
    if rank == src:
        return table
    target = torch.device(device) if device is not None else comm_dev
    return WeightTable.from_meta(meta.cpu(), payload.to(target))


def prepare_tables(filter_id: int, kind: int, in_hw: Tuple[int, int], out_hw: Tuple[int, int], align_corners: bool,
                   device: torch.device, src: int = 0, group=None) -> List[WeightTable]:
    """Rank ``src`` builds the H and W tables on its GPU; everyone else receives them and installs them in the local
    cache, so the following forward calls on every rank hit the cache.  Returns [table_h, table_w]."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    out: List[WeightTable] = []
    for n_in, n_out in ((in_hw[0], out_hw[0]), (in_hw[1], out_hw[1])):
        t = tables.get_table(filter_id, kind, n_in, n_out, align_corners, 0.0, device) if rank == src else None
        if world > 1 or (dist.is_initialized() and _collectives_at_world_one()):
            t = broadcast_table(t, src=src, device=device, group=group)
            if rank != src:
                tables.put_table(t)
        out.append(t)
    return out


def reduce_max_seconds(seconds: float, device: Optional[torch.device] = None, group=None) -> float:
    """MAX over ranks of a timing (bench.py contract)."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not _collectives_at_world_one()):
        return seconds
    backend = dist.get_backend(group)
    dev = torch.device(device) if backend == "nccl" else torch.device("cpu")
    t = torch.tensor([seconds], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def reduce_sum_int(value: int, device: Optional[torch.device] = None, group=None) -> int:
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not _collectives_at_world_one()):
        return value
    backend = dist.get_backend(group)
    dev = torch.device(device) if backend == "nccl" else torch.device("cpu")
    t = torch.tensor([value], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return int(t.item())
