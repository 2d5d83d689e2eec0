"""Data-parallel plumbing of the training step (torch.distributed; backend "nccl" == RCCL over xGMI on
the GPU box, "gloo" in the CPU tests).  Pure tensor-in / tensor-out helpers so the exchange pattern is
testable without a GPU (tests/test_dist_cpu.py drives them with the oracle as the compute).

Pattern per step:
  * images: rank r takes rows [lo, hi) of the global batch; its CE gradient is pre-scaled by
    B_local / B_global (the reference's loss is a mean, lora_train_vlp.py:997);
  * text tower, class-sharded: rank r encodes classes [c_lo, c_hi); the [C, d] class features are
    assembled on every rank (all-gather, realised as an all-reduce of a zero-padded buffer: 825 KB at
    C = 403, latency-bound either way and it also runs under gloo); their gradient is summed over ranks
    and each rank back-propagates its own rows;
  * ONE all-reduce of the flat LoRA + prompt gradient buffer (1.5 MB), then the fused AdamW.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def world_info(group=None) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Balanced contiguous partition of range(n): sizes differ by at most one."""
    return (n * rank) // world, (n * (rank + 1)) // world


def allgather_rows(local: Optional[torch.Tensor], lo: int, hi: int, total: int, width: int, like: torch.Tensor,
                   group=None) -> torch.Tensor:
    """Every rank contributes rows [lo, hi) of a [total, width] matrix; returns the full matrix."""
    full = torch.zeros(total, width, device=like.device, dtype=like.dtype)
    if hi > lo:
        full[lo:hi].copy_(local)
    _, world = world_info(group)
    if world > 1:
        dist.all_reduce(full, group=group)
    return full


def allreduce_sum_(t: torch.Tensor, group=None) -> torch.Tensor:
    _, world = world_info(group)
    if world > 1:
        dist.all_reduce(t, group=group)
    return t


def allgather_int_rows(local: Optional[torch.Tensor], lo: int, hi: int, total: int, width: int, device,
                       group=None) -> torch.Tensor:
    """Integer results (top-5 labels, base/new flags) of rows [lo, hi) from every rank -> the full [total, width]
    int32 table on every rank (the only exchange of the TTA / OOD path).  Same
    zero-padded all-reduce as ``allgather_rows``: exact for integers, runs under RCCL and gloo."""
    full = torch.zeros(total, width, device=device, dtype=torch.int32)
    if hi > lo:
        full[lo:hi].copy_(local.to(torch.int32).reshape(hi - lo, width))
    _, world = world_info(group)
    if world > 1:
        dist.all_reduce(full, group=group)
    return full
