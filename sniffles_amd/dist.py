"""Contig sharding across ranks and the final gather.

Contig tasks are independent (reference: one CallTask per contig, no shared state, parallel.py:264), so ranks
never exchange data on the data path.  The only collective is the gather of the per-rank call records, run by
torch.distributed on whatever backend the process group has (RCCL on GPUs: backend "nccl"; gloo in CPU tests).
"""
from __future__ import annotations


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


class TaskQueue:
    """A shared queue of task indices, heaviest first, over the process group's key-value store (no data-path
    collective): every rank claims the next index with one atomic add, so a rank that got short contigs simply comes
    back earlier - the self-balancing alternative to `shard_lpt` when the cost of a contig is not known up front
    (reference: the parent's task deque that idle workers pull from, parallel.py:652-680).

        q = TaskQueue([t.n_leads for t in tasks])
        mine = [i for i in q]           # indices this rank processed, in claim order

    Every queue counts on its own store key: `key` plus a per-process generation number, so a second queue in the same
    process group (another pass, another sample, a retry) starts at zero again.  All ranks must therefore construct
    their queues in the same order - which they do anyway, a queue being a collective object.  `claim` is thread-safe
    (the store serialises the adds), so several host threads of a rank may drain one queue."""

    _generation = {}

    def __init__(self, weights, store=None, key: str = "snf_task_queue", barrier: bool = True):
        import torch.distributed as dist
        self.order = sorted(range(len(weights)), key=lambda i: (-weights[i], i))
        if store is None:
            from torch.distributed import distributed_c10d
            store = distributed_c10d._get_default_store()
        gen = TaskQueue._generation.get(key, 0)
        TaskQueue._generation[key] = gen + 1
        self.store, self.key = store, f"{key}/{gen}"
        self.claimed = []
        if barrier:
            dist.barrier()      # every rank has built the same order before the first claim

    def claim(self):
        k = int(self.store.add(self.key, 1)) - 1
        if k >= len(self.order):
            return None
        self.claimed.append(self.order[k])
        return self.order[k]

    def __iter__(self):
        while True:
            i = self.claim()
            if i is None:
                return
            yield i


class LocalQueue:
    """The same interface for a single process (one rank, several host threads): a counter under a lock."""

    def __init__(self, weights):
        import threading
        self.order = sorted(range(len(weights)), key=lambda i: (-weights[i], i))
        self._lock, self._next = threading.Lock(), 0
        self.claimed = []

    def claim(self):
        with self._lock:
            k = self._next
            self._next += 1
        if k >= len(self.order):
            return None
        self.claimed.append(self.order[k])
        return self.order[k]

    __iter__ = TaskQueue.__iter__


def gather_claims(claimed, world: int) -> list:
    """Which rank processed which tasks (per rank, in processing order): needed to map the rank-local task_index of the
    gathered records back to the task."""
    import torch.distributed as dist
    out = [None] * world
    dist.all_gather_object(out, list(claimed))
    return out
