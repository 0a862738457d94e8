"""N > 1 path on CPU: world_size-2 gloo, contig tasks sharded longest-first, per-rank hot path (kernel bodies via
the host emulation), all-gather of the call records, and equality with a single-process run of all tasks."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tasks():
    from sniffles_amd import synth
    names = ["chr18", "chr19", "chr20", "chr21", "chr22"]
    return [synth.gen_task(i, c, int(synth.GRCH38[c] * 0.01), 30, 1) for i, c in enumerate(names)]


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import emu.emu as E
    from sniffles_amd import abi, dist as sdist, lib
    from sniffles_amd.config import SnifflesConfig
    tasks = _tasks()
    mine = sdist.shard_lpt([t.contig_len for t in tasks], world)[rank]
    cfg = SnifflesConfig()
    with lib.Batch(cfg, [tasks[i] for i in mine], _lib=E.lib()) as b:
        b.call_candidates(); b.finalize()
        res = b.fetch(1)
    cap = 4096
    buf = torch.zeros(cap * abi.CALL_DTYPE.itemsize, dtype=torch.uint8)
    raw = res.calls.tobytes()
    buf[:len(raw)] = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
    counts, gathered = sdist.gather_calls(buf, len(res.calls), cap, world)
    per_rank = sdist.unpack_gathered(counts, gathered, cap)
    if rank == 0:
        # key the records by (global task id, sv_id): task_index is rank-local, map it back through the shard lists
        shards = sdist.shard_lpt([t.contig_len for t in tasks], world)
        keys = []
        for r, arr in enumerate(per_rank):
            for c in arr:
                keys.append((shards[r][int(c["task_index"])], int(c["sv_id"]), int(c["pos"]), int(c["svlen"]), int(c["filter"]),
                             int(c["support"]), int(c["gt_a"]), int(c["gt_b"])))
        q.put(sorted(keys))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather_equals_single_process():
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import emu.emu as E
    from sniffles_amd import lib, dist as sdist
    from sniffles_amd.config import SnifflesConfig
    E.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    tasks = _tasks()
    with lib.Batch(SnifflesConfig(), tasks, _lib=E.lib()) as b:
        b.call_candidates(); b.finalize()
        res = b.fetch(1)
    exp = sorted((int(c["task_index"]), int(c["sv_id"]), int(c["pos"]), int(c["svlen"]), int(c["filter"]), int(c["support"]),
                  int(c["gt_a"]), int(c["gt_b"])) for c in res.calls)
    assert got == exp and len(exp) > 50
    # sharding is a partition and is balanced
    shards = sdist.shard_lpt([t.contig_len for t in tasks], 2)
    assert sorted(shards[0] + shards[1]) == list(range(len(tasks)))
    loads = [sum(tasks[i].contig_len for i in s) for s in shards]
    assert max(loads) <= 1.35 * min(loads)


def _queue_worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import time
    import emu.emu as E
    from sniffles_amd import abi, dist as sdist, lib
    from sniffles_amd.config import SnifflesConfig
    tasks = _tasks()
    cfg = SnifflesConfig()
    queue = sdist.TaskQueue([t.n_leads for t in tasks])
    recs = []
    for i in queue:                       # one task per claim; rank 1 is slowed down so that rank 0 takes more
        with lib.Batch(cfg, [tasks[i]], _lib=E.lib()) as b:
            b.call_candidates(); b.finalize()
            recs.append(b.fetch(1).calls.copy())
        if rank == 1:
            time.sleep(1.0)
    claims = sdist.gather_claims(queue.claimed, world)
    # further queues in the same process group start from zero again (own store key each), also when they are built
    # ahead of time without a barrier, and several host threads of a rank may drain one queue
    q2, q3 = sdist.TaskQueue([3, 1, 2, 5], barrier=False), sdist.TaskQueue([1] * 40, barrier=False)
    dist.barrier()
    claims2 = sdist.gather_claims(list(q2), world)
    import threading
    ths = [threading.Thread(target=lambda: [time.sleep(0.001) for _ in q3]) for _ in range(3)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    claims3 = sdist.gather_claims(q3.claimed, world)
    cap = 4096
    mine = np.concatenate(recs) if recs else np.zeros(0, abi.CALL_DTYPE)
    off = 0                               # task_index is batch-local (always 0 here): store the position in the claim list
    for k, arr in enumerate(recs):
        mine["task_index"][off:off + len(arr)] = k
        off += len(arr)
    buf = torch.zeros(cap * abi.CALL_DTYPE.itemsize, dtype=torch.uint8)
    raw = mine.tobytes()
    buf[:len(raw)] = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
    counts, gathered = sdist.gather_calls(buf, len(mine), cap, world)
    per_rank = sdist.unpack_gathered(counts, gathered, cap)
    if rank == 0:
        keys = sorted((claims[r][int(c["task_index"])], int(c["sv_id"]), int(c["pos"]), int(c["svlen"]), int(c["filter"]),
                       int(c["support"]), int(c["gt_a"]), int(c["gt_b"])) for r, arr in enumerate(per_rank) for c in arr)
        q.put((keys, claims, claims2, claims3))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_task_queue_equals_single_process():
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import emu.emu as E
    from sniffles_amd import lib
    from sniffles_amd.config import SnifflesConfig
    E.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_queue_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, claims, claims2, claims3 = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    tasks = _tasks()
    with lib.Batch(SnifflesConfig(), tasks, _lib=E.lib()) as b:
        b.call_candidates(); b.finalize()
        res = b.fetch(1)
    exp = sorted((int(c["task_index"]), int(c["sv_id"]), int(c["pos"]), int(c["svlen"]), int(c["filter"]), int(c["support"]),
                  int(c["gt_a"]), int(c["gt_b"])) for c in res.calls)
    assert got == exp
    assert sorted(claims[0] + claims[1]) == list(range(len(tasks)))     # every task exactly once
    assert len(claims[0]) > len(claims[1]) >= 1                           # the faster rank came back for more
    assert sorted(claims2[0] + claims2[1]) == [0, 1, 2, 3]               # a second queue is a fresh queue
    assert (claims2[0] + claims2[1])[0] in (3,) or 3 in (claims2[0][:1] + claims2[1][:1])   # heaviest claimed first
    assert sorted(claims3[0] + claims3[1]) == list(range(40))            # threads x ranks: still every item exactly once


def _scatter_worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import emu.emu as E
    import golden_util as gu
    from sniffles_amd import dist as sdist, parallel
    from test_combine import group_record, make_cfg
    from test_combine_task import BlocksReader
    doc = gu.load("combine_task_6samples")
    exp = doc["expected"]
    sc = exp["scatter"]
    cfg = make_cfg(doc["reference_args"], exp["n_samples"])
    cfg.sample_ids_vcf = [(s, f"S{s}") for s in range(exp["n_samples"])]
    cfg.threads = sc["threads"]
    readers = {s: BlocksReader(exp["contig"], exp["samples"][s]) for s in range(exp["n_samples"])}
    task = parallel.CombineTask(id=7, sv_id=0, contig=exp["contig"], start=0, end=exp["contig_len"], config=cfg, _lib=E.lib())
    task.TARGET_WORK_PER_TASK = sc["target_work_per_task"]
    parts = task.scatter()                                    # the same cuts on every rank
    queue = sdist.TaskQueue([len(p.block_indices) for p in parts])   # the parts of ONE contig go to whoever is free
    mine = {}
    for i in queue:
        mine[parts[i].id] = [group_record(c) for c in parts[i].execute(readers)]
    out = [None] * world
    dist.all_gather_object(out, mine)
    if rank == 0:
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_combine_scatter_equals_reference_parts():
    """A merge of ONE contig over two ranks: the parts of CombineTask.scatter (the reference's cuts) are claimed from the
    work queue, every part's calls equal the reference's for that part, every part is done exactly once."""
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import emu.emu as E
    import golden_util as gu
    E.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_scatter_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    per_rank = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = gu.load("combine_task_6samples")["expected"]["scatter"]["tasks"]
    assert sorted(list(per_rank[0]) + list(per_rank[1])) == sorted(w["id"] for w in want)
    merged = {**per_rank[0], **per_rank[1]}
    for w in want:
        assert len(merged[w["id"]]) == len(w["calls"])
        for g, e in zip(merged[w["id"]], w["calls"]):
            assert gu.diff_records([g], [e]) == []
