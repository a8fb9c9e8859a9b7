"""Batch-row sharding of BK-Core across the GPUs of a node.

Every batch row is an independent tridiagonal system (the reference batches with
``vmap(in_dims=(0, 0, 0, None))``, src/models/bk_core.py:214-216), so the operator shards by rows with
no collective in the forward or the backward: rank r of P owns a contiguous block of rows and runs the
ordinary single-GPU operator on it.  The helpers here are the host-side logic of that partition; they
work with any torch.distributed backend (RCCL on the GPUs, gloo in the CPU tests).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch


def row_shard(n_rows: int, world_size: int, rank: int) -> Tuple[int, int]:
    """[start, stop) of the rows rank `rank` owns: contiguous blocks, sizes differing by at most one."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank/world_size {rank}/{world_size}")
    base, rem = divmod(n_rows, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_rows(x: torch.Tensor, world_size: int, rank: int) -> torch.Tensor:
    """The rank's rows of a (B, ...) operand.  A broadcast operand (stride 0 over the batch, as produced
    by expand()) stays a broadcast view."""
    start, stop = row_shard(x.shape[0], world_size, rank)
    return x[start:stop]


def gather_rows(local: torch.Tensor, n_rows: int, group=None) -> torch.Tensor:
    """Concatenate the ranks' row blocks back into (n_rows, ...) on every rank (test / harness helper; the
    operator itself never communicates)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = [row_shard(n_rows, world, r) for r in range(world)]
    biggest = max(b - a for a, b in sizes)
    pad = torch.zeros((biggest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([buf[: b - a] for buf, (a, b) in zip(bufs, sizes)], dim=0)
