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


LAYOUT_FIELDS = ("n_calls", "rnames_len", "alt_pool_len", "off_rnames", "off_alt", "bytes")


class GatheredResult:
    """What rank `dst` holds after `gather_results`: every rank's result block in host memory plus the index that makes ONE
    record table of them - `calls` ordered by task id, and inside a task in the order the ranks produced (position order for
    SNF_OUT_EXECUTE results - the order the reference's parent writes, `sniffles:544` over the per-task
    `sorted(svcalls, key=pos)` of `parallel.py:270-271`).  `calls["task_index"]` holds the GLOBAL task id; `alt_off` /
    `rn_off` index `alt_pool` / `rnames` of this object (same interface as `abi.Result`, so
    `vcf.VCF.write_records(res, ti, order)` takes it as it is).  `names`: optional {task id: {qname id: str}} of the
    supporting reads (gathered on request: `--output-rnames`).

    The table is assembled on first use (`calls` / `alt_pool` / `rnames`): the gather itself only lands the blocks and builds
    the per-task index (`n_calls`, `task_ids`, `task_call_off` cost nothing) - what the reference's parent holds right after
    `recv()`.  A result whose blocks live in a landing buffer of `gather_results` is valid until the gather after next
    reuses that buffer; `detach()` copies it out."""

    def __init__(self, calls=None, alt_pool=None, rnames=None, task_ids=(), names=None, blocks=None, task_ids_per_rank=None):
        import numpy as np
        self.names = names
        self._blocks, self._ids = blocks, task_ids_per_rank
        self._calls, self._alt, self._rn = calls, alt_pool, rnames
        self.task_ids = np.asarray(sorted(set(int(i) for i in task_ids)), np.int64)   # (a task id listed by several ranks: held by one)
        if blocks is None:
            t = calls["task_index"].astype(np.int64)
            self.task_call_off = np.searchsorted(t, np.concatenate([self.task_ids, [np.iinfo(np.int64).max]]), side="left")
            self.task_call_off[-1] = len(calls)
            self._runs = None
            return
        # per global task: (rank, first row, rows) in its rank's block - from the local task index column of the records
        # (a strided 4-byte read per record; the records themselves are not touched until `calls` is asked for)
        runs = {}
        for r, ((lay, blob), ids) in enumerate(zip(blocks, task_ids_per_rank)):
            n = int(lay["n_calls"])
            if not n:
                continue
            loc = np.frombuffer(blob, abi.CALL_DTYPE, count=n)["task_index"]
            cut = np.flatnonzero(np.diff(loc)) + 1                       # a task's calls are contiguous in its rank's block
            starts = np.concatenate([[0], cut]); ends = np.concatenate([cut, [n]])
            for s0, e0 in zip(starts.tolist(), ends.tolist()):
                runs[int(ids[int(loc[s0])])] = (r, s0, e0 - s0)
        self._runs = [runs.get(int(t), (0, 0, 0)) for t in self.task_ids]
        self.task_call_off = np.concatenate([[0], np.cumsum([k[2] for k in self._runs])]).astype(np.int64)

    @property
    def n_calls(self) -> int:
        return int(self.task_call_off[-1])

    def _assemble(self):
        import numpy as np
        if self._calls is not None:
            return
        blocks = self._blocks
        alt_base, rn_base, a0, r0 = [], [], 0, 0
        for lay, _ in blocks:
            alt_base.append(a0); rn_base.append(r0)
            a0 += int(lay["alt_pool_len"]); r0 += int(lay["rnames_len"])
        calls = np.empty(self.n_calls, abi.CALL_DTYPE)
        tabs = [np.frombuffer(blob, abi.CALL_DTYPE, count=int(lay["n_calls"])) for lay, blob in blocks]
        for t, (r, s0, n), o in zip(self.task_ids.tolist(), self._runs, self.task_call_off.tolist()):
            if not n:
                continue
            c = calls[o:o + n]
            c[:] = tabs[r][s0:s0 + n]                                    # one contiguous copy per task
            c["task_index"] = t
            if alt_base[r]:
                c["alt_off"] += alt_base[r]
            if rn_base[r]:
                c["rn_off"] += rn_base[r]
        def part(lay, blob, off, nbytes):
            return np.frombuffer(blob, np.uint8, count=int(lay["bytes"]))[int(lay[off]):int(lay[off]) + nbytes]
        alts = [part(lay, blob, "off_alt", int(lay["alt_pool_len"])) for lay, blob in blocks]
        rns = [part(lay, blob, "off_rnames", 4 * int(lay["rnames_len"])).view(np.uint32) for lay, blob in blocks]
        self._alt = alts[0] if len(alts) == 1 else np.concatenate(alts) if alts else np.zeros(0, np.uint8)   # (one rank: a view)
        self._rn = rns[0] if len(rns) == 1 else np.concatenate(rns) if rns else np.zeros(0, np.uint32)
        self._calls = calls

    @property
    def calls(self):
        self._assemble()
        return self._calls

    @property
    def alt_pool(self):
        self._assemble()
        return self._alt

    @property
    def rnames(self):
        self._assemble()
        return self._rn

    def detach(self):
        """The table in memory of its own (independent of the landing buffer)."""
        import numpy as np
        self._assemble()
        self._alt, self._rn = np.array(self._alt), np.array(self._rn)
        self._blocks = None
        return self

    def task_rows(self, task_id: int):
        """Row indices of a task's calls (output order)."""
        import numpy as np
        k = int(np.searchsorted(self.task_ids, task_id))
        if k >= len(self.task_ids) or int(self.task_ids[k]) != task_id:
            return np.zeros(0, np.int64)
        return np.arange(int(self.task_call_off[k]), int(self.task_call_off[k + 1]), dtype=np.int64)

    alt = abi.Result.alt
    rn = abi.Result.rn


def merge_blocks(blocks, task_ids_per_rank) -> GatheredResult:
    """blocks: per rank (layout dict, bytes-like of the rank's result block [records | read names | ALT bytes], see
    `snf_batch_export_device`); task_ids_per_rank[r][k] = global id of rank r's batch-local task k.  Pure host code."""
    return GatheredResult(task_ids=[int(i) for ids in task_ids_per_rank for i in ids], blocks=list(blocks),
                          task_ids_per_rank=[list(ids) for ids in task_ids_per_rank])


_LANDING = {}     # (device, slot) -> pinned host tensor the gathered blocks are copied into (two slots, used in turn)
_LANDING_TURN = [0]


def _landing(nbytes: int, pinned: bool):
    import torch
    slot = _LANDING_TURN[0] = (_LANDING_TURN[0] + 1) & 1
    t = _LANDING.get((pinned, slot))
    if t is None or t.numel() < nbytes:
        t = torch.empty(max(nbytes, 1 << 20) * 5 // 4, dtype=torch.uint8, pin_memory=pinned)
        _LANDING[(pinned, slot)] = t
    return t


def gather_results(block, layout: dict, task_ids, dst: int = 0, group=None, names=None, recv_buffer=None, task_ids_per_rank=None):
    """The final gather (the reference's parent receives whole `SVCall`s, parallel.py:757): every rank hands in
    its finalized result block - `block`, a uint8 tensor on the process group's device holding [records | read names | ALT
    bytes] as `Batch.export_device` (or a fetch) left it, with its `layout` - and rank `dst` returns the `GatheredResult` of
    all ranks (None elsewhere).  Collectives, in this order on every rank: one all-gather of the layouts (6 x int64), one
    gather of the blocks onto `dst` (every rank sends the largest block size, so `dst` receives world x max bytes over
    its point-to-point links), and - only when `names` is given ({local task index: {qname id: str}} of the supporting
    reads, for `--output-rnames`) - one gather_object.  `task_ids[k]` = global task id of the batch-local task k.
    `task_ids_per_rank` (every rank's list, e.g. from a deterministic `shard_lpt`) saves the all_gather_object that otherwise
    collects them.  `recv_buffer`: optional uint8 tensor on `dst` to receive into (reused across passes)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = block.device
    mine = torch.tensor([int(layout[f]) for f in LAYOUT_FIELDS] + [len(task_ids)], dtype=torch.int64, device=dev)
    lays = torch.zeros(world * mine.numel(), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(lays, mine, group=group)
    lays = lays.cpu().view(world, -1)
    nmax = int(lays[:, 5].max())
    nmax = (nmax + 255) & ~255
    if block.numel() < nmax:      # (the caller's buffer is at least its own block; pad to the common size)
        block = torch.cat([block, torch.zeros(nmax - block.numel(), dtype=torch.uint8, device=dev)])
    chunk = block[:nmax]
    ids_all = task_ids_per_rank
    if ids_all is None:
        ids_all = [None] * world
        dist.all_gather_object(ids_all, [int(i) for i in task_ids], group=group)
    if rank == dst:
        if recv_buffer is None or recv_buffer.numel() < world * max(nmax, 1):
            recv_buffer = torch.empty(world * max(nmax, 1), dtype=torch.uint8, device=dev)
        parts = [recv_buffer[r * nmax:(r + 1) * nmax] for r in range(world)]
        if nmax:
            dist.gather(chunk, gather_list=parts, dst=dst, group=group)
    elif nmax:
        dist.gather(chunk, dst=dst, group=group)
    names_all = None
    if names is not None:
        names_all = [None] * world if rank == dst else None
        dist.gather_object({int(task_ids[k]): v for k, v in names.items()}, names_all, dst=dst, group=group)
    if rank != dst:
        return None
    # the blocks land in (pinned) host memory: one copy per rank of exactly its bytes, at 256-byte aligned offsets
    sizes = [int(lays[r, 5]) for r in range(world)]
    offs, tot = [], 0
    for n in sizes:
        offs.append(tot); tot += (n + 255) & ~255
    land = _landing(tot, dev.type == "cuda")
    for r in range(world):
        if sizes[r]:
            land[offs[r]:offs[r] + sizes[r]].copy_(recv_buffer[r * nmax:r * nmax + sizes[r]], non_blocking=True)
    if dev.type == "cuda":
        torch.cuda.current_stream(dev).synchronize()
    host = land.numpy()
    blocks = []
    for r in range(world):
        lay = dict(zip(LAYOUT_FIELDS, (int(x) for x in lays[r, :6])))
        blocks.append((lay, host[offs[r]:offs[r] + sizes[r]]))
    out = merge_blocks(blocks, ids_all)
    if names_all is not None:
        out.names = {}
        for d in names_all:
            out.names.update(d)
    return out


class SharedLanding:
    """Node-local landing of the results (one process per GPU on ONE node): every rank owns `slots` shared-memory segments
    under /dev/shm, `Batch.set_result_memory` makes its kernels write the finalized result straight into one of them - over
    that GPU's own PCIe link -, and rank `dst` maps all of them.  The gather (`gather_results_shared`) then only exchanges the
    layouts: no result byte crosses a second link or is copied on the host, and N GPUs land their results through N links
    instead of funnelling them through the one of rank `dst` (the block gather of `gather_results` is what a multi-node job,
    or a node without /dev/shm, uses).  A segment is [ block: records | read names ][ ALT section at `block_bytes` ]."""

    def __init__(self, slots: int, block_bytes: int, alt_bytes: int, dst: int = 0, group=None, directory: str = "/dev/shm"):
        import os
        import numpy as np
        import torch.distributed as dist
        self.world, self.rank, self.dst = dist.get_world_size(group), dist.get_rank(group), dst
        self.slots, self.block_bytes, self.alt_bytes = int(slots), (int(block_bytes) + 4095) & ~4095, (int(alt_bytes) + 4095) & ~4095
        tag = [None]
        if self.rank == dst:
            tag[0] = "snf_%d_%s" % (os.getpid(), os.urandom(4).hex())
        dist.broadcast_object_list(tag, src=dst, group=group)
        self._paths = {(r, k): os.path.join(directory, "%s_r%d_s%d" % (tag[0], r, k)) for r in range(self.world) for k in range(self.slots)}
        size = self.block_bytes + self.alt_bytes
        self._own = []
        for k in range(self.slots):
            path = self._paths[(self.rank, k)]
            with open(path, "wb") as f:
                f.truncate(size)
            self._own.append(np.memmap(path, np.uint8, "r+", shape=(size,)))
        dist.barrier(group)                                  # every segment exists
        self._all = {}
        if self.rank == dst:
            for (r, k), path in self._paths.items():
                self._all[(r, k)] = self._own[k] if r == self.rank else np.memmap(path, np.uint8, "r", shape=(size,))
        dist.barrier(group)                                  # ... and is mapped where it is read: the names can go
        for k in range(self.slots):
            try:
                os.unlink(self._paths[(self.rank, k)])
            except OSError:
                pass

    def memory(self, slot: int):
        """(block, alt) of this rank's segment `slot`, for `Batch.set_result_memory`."""
        m = self._own[slot]
        return m[:self.block_bytes], m[self.block_bytes:]

    def view(self, rank: int, slot: int):
        return self._all[(rank, slot)]


def gather_results_shared(landing: SharedLanding, slot: int, layout: dict, task_ids, group=None, task_ids_per_rank=None,
                          device=None):
    """The gather over a `SharedLanding`: this rank's finalized result lies in its segment `slot` (`Batch.fetch_layout` returned
    `layout`); one all-gather of the layouts (9 x int64) and rank `dst` returns the `GatheredResult` over the segments of all
    ranks (None elsewhere) - views, valid until the owning batches run their next pass into the same segments (`detach()`
    copies).  `device`: where the small collective's tensor lives (the process group's device; default: CPU)."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lay = dict(layout)
    lay["off_alt"] = landing.block_bytes
    lay["bytes"] = landing.block_bytes + int(lay["alt_pool_len"])
    if int(lay["off_rnames"]) + 4 * int(lay["rnames_len"]) > landing.block_bytes or int(lay["alt_pool_len"]) > landing.alt_bytes:
        raise ValueError("the result does not fit the shared segment")
    mine = torch.tensor([int(lay[f]) for f in LAYOUT_FIELDS] + [len(task_ids), int(slot), 0], dtype=torch.int64, device=device or "cpu")
    lays = torch.zeros(world * mine.numel(), dtype=torch.int64, device=mine.device)
    dist.all_gather_into_tensor(lays, mine, group=group)
    ids_all = task_ids_per_rank
    if ids_all is None:
        ids_all = [None] * world
        dist.all_gather_object(ids_all, [int(i) for i in task_ids], group=group)
    if rank != landing.dst:
        return None
    lays = lays.cpu().view(world, -1)
    blocks = []
    for r in range(world):
        d = dict(zip(LAYOUT_FIELDS, (int(x) for x in lays[r, :6])))
        blocks.append((d, landing.view(r, int(lays[r, 7]))[:d["bytes"]]))
    return merge_blocks(blocks, ids_all)


def gather_sets_shared(landing: SharedLanding, entries, max_entries: int, set_task_ids, group=None, device=None):
    """The gather of one pass of the strong-scaling shape (ONE genome over N ranks, the reference's pull queue, `sniffles:495-530`,
    `parallel.py:652-717`): the contig tasks are partitioned into SETS, every rank claims sets from a `TaskQueue` and runs each as
    one device batch whose result its kernels store into a segment of the rank's `SharedLanding`.  `entries`: what this rank
    produced in this pass, [(slot, layout from `Batch.fetch_layout`, set index)] - possibly empty, at most `max_entries`.  ONE
    all-gather of (1 + 8 x max_entries) int64 per rank - whichever sets a rank happened to claim, every rank issues the same
    collective - and rank `dst` returns the `GatheredResult` of the whole genome (None elsewhere): per set the task ids
    `set_task_ids[set index]` in batch order, blocks read in place (views; `detach()` copies).  No result byte is copied on the
    host or crosses a second link."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if len(entries) > max_entries:
        raise ValueError("more sets in one pass than the gather was sized for")
    row = [len(entries)]
    for slot, layout, set_index in entries:
        if int(layout["off_rnames"]) + 4 * int(layout["rnames_len"]) > landing.block_bytes or int(layout["alt_pool_len"]) > landing.alt_bytes:
            raise ValueError("the result does not fit the shared segment")
        row += [int(layout["n_calls"]), int(layout["rnames_len"]), int(layout["alt_pool_len"]), int(layout["off_rnames"]),
                landing.block_bytes, landing.block_bytes + int(layout["alt_pool_len"]), int(slot), int(set_index)]
    row += [0] * (1 + 8 * max_entries - len(row))
    mine = torch.tensor(row, dtype=torch.int64, device=device or "cpu")
    lays = torch.zeros(world * mine.numel(), dtype=torch.int64, device=mine.device)
    dist.all_gather_into_tensor(lays, mine, group=group)
    if rank != landing.dst:
        return None
    lays = lays.cpu().view(world, -1).tolist()
    blocks, ids, seen = [], [], set()
    for r in range(world):
        for k in range(int(lays[r][0])):
            f = lays[r][1 + 8 * k:9 + 8 * k]
            d = dict(zip(LAYOUT_FIELDS, f[:6]))
            if f[7] in seen:
                raise RuntimeError(f"set {f[7]} was served twice in one pass")
            seen.add(f[7])
            blocks.append((d, landing.view(r, int(f[6]))[:d["bytes"]]))
            ids.append(list(set_task_ids[int(f[7])]))
    return merge_blocks(blocks, ids)


def result_block(res):
    """(layout, bytes) of a fetched stage-1 result in the export format - for ranks whose result is already on the host
    (tests over gloo; `Batch.export_device` is the device form)."""
    import numpy as np
    n = len(res.calls)
    off_rn = (n * abi.CALL_DTYPE.itemsize + 255) & ~255
    off_alt = (off_rn + 4 * len(res.rnames) + 255) & ~255
    blob = np.zeros(off_alt + len(res.alt_pool), np.uint8)
    blob[:n * abi.CALL_DTYPE.itemsize] = np.frombuffer(res.calls.tobytes(), np.uint8)
    blob[off_rn:off_rn + 4 * len(res.rnames)] = np.frombuffer(np.ascontiguousarray(res.rnames, np.uint32).tobytes(), np.uint8)
    blob[off_alt:] = res.alt_pool
    return dict(n_calls=n, rnames_len=len(res.rnames), alt_pool_len=len(res.alt_pool), off_rnames=off_rn, off_alt=off_alt,
                bytes=len(blob)), blob


_STORE = None


def set_store(store) -> None:
    """The key-value store `TaskQueue` counts on when none is passed: the one the process group was initialised with
    (`dist.init_process_group(..., store=store)`), or any `torch.distributed.Store` every rank can reach."""
    global _STORE
    _STORE = store


def default_store():
    """A store shared by all ranks.  `set_store` wins; otherwise a TCPStore client of the rendezvous the process group itself
    was built from (MASTER_ADDR / MASTER_PORT of the env:// rendezvous; rank 0 hosts the server there) - public API only."""
    global _STORE
    if _STORE is None:
        import os
        import torch.distributed as dist
        _STORE = dist.TCPStore(os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]), dist.get_world_size(),
                               is_master=False, timeout=__import__("datetime").timedelta(seconds=120), wait_for_workers=False)
    return _STORE


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
            store = default_store()
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
