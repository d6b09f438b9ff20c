"""Clip-level driver pieces: static clip sharding across ranks and the one exchange step
(gather of per-clip masks).  The reference distributes clips through an mp.Queue work queue
and funnels statistics through a second queue (tools/eval.py:137-143,
managers/evaluator.py:276-295,589-613); clips are independent, so here the shard is static
(clip i -> rank i mod world) and the masks are collected with one all-gather
(RCCL on GPUs, gloo in the CPU tests)."""
from __future__ import annotations

from typing import List

import torch


def shard_clips(n_clips: int, world: int, rank: int) -> List[int]:
    """Clip ids owned by `rank` (round-robin, as balanced as the reference's queue for
    equal-length clips)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return list(range(rank, n_clips, world))


def gather_masks(local_masks: torch.Tensor, world: int, group=None) -> torch.Tensor:
    """all-gather uint8 masks [clips_per_rank, F, H, W] -> [world*clips_per_rank, F, H, W]
    ordered by rank.  Every rank must contribute the same shape (pad short shards)."""
    if world == 1:
        return local_masks
    import torch.distributed as dist
    if local_masks.dtype != torch.uint8:
        raise TypeError("masks are exchanged as uint8 label maps")
    out = torch.empty((world * local_masks.shape[0],) + tuple(local_masks.shape[1:]),
                      dtype=torch.uint8, device=local_masks.device)
    dist.all_gather_into_tensor(out, local_masks.contiguous(), group=group)
    return out


def unshard_order(n_clips: int, world: int) -> List[int]:
    """Position in the gathered tensor -> clip id (inverse of shard_clips + rank-major gather),
    assuming n_clips % world == 0."""
    per = n_clips // world
    return [r + world * j for r in range(world) for j in range(per)]
