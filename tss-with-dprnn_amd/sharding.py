"""Utterance sharding for one-process-per-GPU inference.

The forward path has no cross-utterance dependency (GroupNorm(1,C)/gLN are per sample, dprnn.py:76-77;
LSTM state starts at zero per sequence, dprnn.py:36), so ranks simply own disjoint utterance ranges and the only
communication is a gather of per-utterance scalars (the mean the reference forms in inferencer.py:84-89) and the
max-over-ranks of a wall-clock for benchmarking.  Works on any torch.distributed backend: "nccl" (= RCCL over
xGMI on MI355X) in production, "gloo" in the CPU tests.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [start, stop) of `n_items` for `rank` (sizes differ by at most one)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_by_length(lengths: Sequence[int], world: int) -> List[List[int]]:
    """Greedy longest-first assignment of utterance indices to ranks so that the summed lengths balance
    (full-length test utterances range 3 s .. 14 s: SURVEY.md section 8d)."""
    order = sorted(range(len(lengths)), key=lambda i: -lengths[i])
    load = [0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += lengths[i]
    return [sorted(v) for v in out]


def _dist_ready() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def max_over_ranks(value: float, device=None) -> float:
    """MAX all-reduce of a scalar (bench.py: the slowest rank defines the step time)."""
    if not _dist_ready():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_metric(local: torch.Tensor, n_total: int) -> torch.Tensor:
    """All-gather per-utterance scalars held as contiguous shard_range slices; returns the full [n_total]
    vector on every rank (ranks may hold different counts: padded to the largest shard)."""
    if not _dist_ready():
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    cap = max(b - a for a, b in sizes)
    pad = torch.zeros(cap, dtype=local.dtype, device=local.device)
    pad[: local.numel()] = local.reshape(-1)
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([bufs[r][: sizes[r][1] - sizes[r][0]] for r in range(world)])
