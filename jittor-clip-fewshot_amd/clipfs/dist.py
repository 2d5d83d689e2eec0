"""Data-parallel plumbing of the training step (torch.distributed; backend "nccl" == RCCL over xGMI on
the GPU box, "gloo" in the CPU tests -- the SAME collectives run under both).  Pure tensor-in / tensor-out
helpers so the exchange pattern is testable without a GPU (tests/test_dist_cpu.py drives them with the
oracle as the compute).

Pattern per step, bytes at cfg-3 (C = 403 classes, d = 512, N = 8 ranks, S = 51):
  * images: rank r takes rows [lo, hi) of the global batch; its CE gradient is pre-scaled by
    B_local / B_global (the reference's loss is a mean, lora_train_vlp.py:997);
  * text tower, class-sharded in blocks of S = ceil(C / N) classes: rank r encodes classes
    [r S, min((r+1) S, C));
      forward  : ONE ``all_gather_into_tensor`` of the [S, d] class-feature block  (send 104 KB / rank,
                 result [N S, d] = 835 KB, rows >= C are padding);
      backward : ONE ``reduce_scatter_tensor`` (sum) of the [N S, d] class-feature gradient -> the [S, d]
                 block of the rank's own classes (835 KB in, 104 KB out);
  * ONE ``all_reduce`` of the flat LoRA + prompt gradient buffer (373 760 floats = 1.5 MB), then the
    fused AdamW.
All three are latency-bound on xGMI (tens of microseconds); nothing is zero-padded to full size any more.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


# Test hook: issue the collectives even in a one-rank group, so that the RCCL entry points themselves (argument
# checks, views as send / receive buffers, stream ordering) are exercised on a one-GPU box.
FORCE_COLLECTIVES = False


def _live(world: int) -> bool:
    return world > 1 or (FORCE_COLLECTIVES and dist.is_available() and dist.is_initialized())


def world_info(group=None) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def backend_name(group=None) -> str:
    if dist.is_available() and dist.is_initialized():
        return str(dist.get_backend(group))
    return "none"


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Balanced contiguous partition of range(n): sizes differ by at most one (image shards)."""
    return (n * rank) // world, (n * (rank + 1)) // world


def block_rows(n: int, world: int) -> int:
    """Height S of the fixed-size blocks of the class partition (the last blocks may be short or empty)."""
    return -(-n // world)


def block_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Block partition of range(n) in blocks of S = ceil(n / world): the layout ``all_gather_into_tensor`` /
    ``reduce_scatter_tensor`` produce and consume without any repacking."""
    s = block_rows(n, world)
    return min(rank * s, n), min((rank + 1) * s, n)


def allgather_blocks(block: torch.Tensor, out: Optional[torch.Tensor] = None, group=None) -> torch.Tensor:
    """``block`` [S, w] (this rank's rows, padding rows zero) -> [world * S, w] on every rank."""
    _, world = world_info(group)
    if out is None:
        out = torch.empty(world * block.shape[0], block.shape[1], device=block.device, dtype=block.dtype)
    # the collectives take flat buffers: a wrong shape or a strided view would be exchanged silently
    if block.dim() != 2 or out.shape != (world * block.shape[0], block.shape[1]) or out.dtype != block.dtype:
        raise ValueError(f"allgather_blocks: block {tuple(block.shape)} x world {world} does not fill out {tuple(out.shape)}")
    if not (block.is_contiguous() and out.is_contiguous()):
        raise ValueError("allgather_blocks: send block and result must be contiguous")
    if _live(world):
        dist.all_gather_into_tensor(out, block, group=group)
    else:
        out.copy_(block)
    return out


def reduce_scatter_blocks(full: torch.Tensor, out: Optional[torch.Tensor] = None, group=None) -> torch.Tensor:
    """``full`` [world * S, w] (every rank's partial sums for ALL rows) -> this rank's [S, w] block of the total."""
    rank, world = world_info(group)
    if full.dim() != 2 or full.shape[0] % world != 0:
        raise ValueError(f"reduce_scatter_blocks: {tuple(full.shape)} is not world {world} x [S, w]")
    s = full.shape[0] // world
    if out is None:
        out = torch.empty(s, full.shape[1], device=full.device, dtype=full.dtype)
    if out.shape != (s, full.shape[1]) or out.dtype != full.dtype:
        raise ValueError(f"reduce_scatter_blocks: out {tuple(out.shape)} is not the [S = {s}, {full.shape[1]}] block")
    if not (full.is_contiguous() and out.is_contiguous()):
        raise ValueError("reduce_scatter_blocks: table and result must be contiguous")
    if _live(world):
        dist.reduce_scatter_tensor(out, full, op=dist.ReduceOp.SUM, group=group)
    else:
        out.copy_(full)
    return out


def allreduce_sum_(t: torch.Tensor, group=None) -> torch.Tensor:
    _, world = world_info(group)
    if _live(world):
        dist.all_reduce(t, group=group)
    return t


def allgather_int_rows(local: Optional[torch.Tensor], lo: int, hi: int, total: int, width: int, device,
                       group=None) -> torch.Tensor:
    """Integer results (top-5 labels, base/new flags) of rows [lo, hi) = ``shard_bounds(total, rank, world)`` from
    every rank -> the full [total, width] int32 table on every rank (the only exchange of
    the TTA / OOD path).  One ``all_gather_into_tensor`` of ceil(total / world)-row slots, then the short slots are
    squeezed out on the receiver."""
    rank, world = world_info(group)
    if world == 1:
        return local.to(torch.int32).reshape(total, width).to(device)
    s = block_rows(total, world)
    slot = torch.zeros(s, width, device=device, dtype=torch.int32)
    if hi > lo:
        slot[:hi - lo].copy_(local.to(torch.int32).reshape(hi - lo, width))
    table = torch.empty(world * s, width, device=device, dtype=torch.int32)
    dist.all_gather_into_tensor(table, slot, group=group)
    parts: List[torch.Tensor] = []
    for r in range(world):
        a, b = shard_bounds(total, r, world)
        parts.append(table[r * s:r * s + (b - a)])
    return torch.cat(parts, 0)
