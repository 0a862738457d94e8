"""Contig sharding across ranks and the final gather.

Contig tasks are independent (reference: one CallTask per contig, no shared state, parallel.py:264), so ranks
never exchange data on the data path.  The only collective is the gather of the per-rank call records, run by
torch.distributed on whatever backend the process group has (RCCL on GPUs: backend "nccl"; gloo in CPU tests).
"""
from __future__ import annotations

import numpy as np

from . import abi


def shard_lpt(weights, n_ranks: int) -> list:
    """Longest-processing-time-first assignment of items (by weight) to ranks; deterministic on every rank.
    Returns per rank the list of item indices, each in descending-weight order."""
    order = sorted(range(len(weights)), key=lambda i: (-weights[i], i))
    load = [0] * n_ranks
    out = [[] for _ in range(n_ranks)]
    for i in order:
        r = min(range(n_ranks), key=lambda k: (load[k], k))
        load[r] += weights[i]
        out[r].append(i)
    return out


def gather_calls(calls_tensor, n_calls: int, cap_calls: int, world: int):
    """All-gather fixed-capacity record buffers + counts; returns (counts[world], gathered uint8 tensor).
    `calls_tensor`: uint8 tensor of cap_calls * sizeof(snf_call_t) bytes on the group's device."""
    import torch
    import torch.distributed as dist
    count = torch.tensor([n_calls], dtype=torch.int64, device=calls_tensor.device)
    counts = torch.zeros(world, dtype=torch.int64, device=calls_tensor.device)
    gathered = torch.empty(world * calls_tensor.numel(), dtype=torch.uint8, device=calls_tensor.device)
    dist.all_gather_into_tensor(counts, count)
    dist.all_gather_into_tensor(gathered, calls_tensor)
    return counts, gathered


def unpack_gathered(counts, gathered, cap_calls: int) -> list:
    """Per rank: numpy structured array of its call records."""
    rec = abi.CALL_DTYPE.itemsize
    g = gathered.cpu().numpy()
    out = []
    for r, n in enumerate(counts.cpu().tolist()):
        lo = r * cap_calls * rec
        out.append(g[lo:lo + int(n) * rec].view(abi.CALL_DTYPE).copy())
    return out
