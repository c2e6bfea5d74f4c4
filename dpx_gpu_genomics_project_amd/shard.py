"""Multi-GPU plumbing: one process per GPU, pairs sharded, one RCCL gather of the scores.

Alignment pairs are independent, so the path shards with NO data-path collective: rank r owns the contiguous
pair range shard_range(num_pairs, r, world) (or, for weak-scaling benchmarks, its own sub-batch), runs the fill on
its GPU, and the int32 scores are gathered to rank 0 with a single collective (torch.distributed: backend
"nccl" is RCCL over xGMI on ROCm; "gloo" in the CPU tests).  4 bytes per pair per rank -- latency-bound.
"""
from __future__ import annotations

from typing import List, Optional, Tuple


def shard_range(num_pairs: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) of ceil(N/G)-sized shards (the last ranks may be short or empty)."""
    if world < 1 or not (0 <= rank < world) or num_pairs < 0:
        raise ValueError("bad shard request")
    per = -(-num_pairs // world)
    lo = min(rank * per, num_pairs)
    return lo, min(lo + per, num_pairs)


def gather_scores(local, rank: int, world: int, shard_sizes: Optional[List[int]] = None, dst: int = 0):
    """Gather every rank's int32 score tensor on `dst`; returns the concatenation there, None elsewhere.

    Equal shard sizes use one dist.gather; ragged shards are padded to the largest shard first (one collective
    either way)."""
    import torch
    import torch.distributed as dist

    if world == 1:
        return local
    sizes = shard_sizes or [local.numel()] * world
    width = max(sizes)
    send = local
    if local.numel() != width:
        send = torch.zeros(width, dtype=local.dtype, device=local.device)
        send[: local.numel()] = local
    bufs = [torch.empty(width, dtype=local.dtype, device=local.device) for _ in range(world)] if rank == dst else None
    dist.gather(send, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)])
