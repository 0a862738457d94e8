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


def vcf_text(cfg, tasks, results_of_task):
    """VCF records (no header) of the tasks in id order straight from record tables: `results_of_task(ti)` -> (res, rows)."""
    import io
    from sniffles_amd import vcf
    buf = io.StringIO()
    w = vcf.VCF(cfg, buf)
    for ti in sorted(tasks, key=lambda t: t.task_id):
        res, rows = results_of_task(ti)
        w.write_records(res, ti, rows)
    return buf.getvalue()


def single_process_text(cfg, tasks, L, mode):
    """Every task in ONE batch of ONE process, CallTask.execute's filter and sort on the device."""
    from sniffles_amd import lib
    with lib.Batch(cfg, tasks) as b:
        b.set_output(mode)
        b.call_candidates(); b.finalize()
        res = b.fetch(1)
    index = {t.task_id: k for k, t in enumerate(tasks)}
    return vcf_text(cfg, tasks, lambda ti: (res, np.arange(int(res.task_call_off[index[ti.task_id]]), int(res.task_call_off[index[ti.task_id] + 1])))), res


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import emu.emu as E
    E.lib()                                              # the host tier becomes the library of this test
    from sniffles_amd import abi, dist as sdist, lib
    from sniffles_amd.config import SnifflesConfig
    tasks = _tasks()
    shards = sdist.shard_lpt([t.contig_len for t in tasks], world)
    mine = shards[rank]
    cfg = SnifflesConfig(output_rnames=True)
    with lib.Batch(cfg, [tasks[i] for i in mine]) as b:
        b.set_output(abi.OUT_EXECUTE | abi.OUT_DEVICE)
        b.call_candidates(); b.finalize()
        # the block leaves the batch as it would for RCCL: device-to-device into a tensor of the process group's device
        send = torch.zeros(1 << 22, dtype=torch.uint8)
        lay = b.export_device(send.data_ptr(), send.numel())
        res = b.fetch(1)
    # names of the supporting reads of this rank's calls (the parent has no qname tables of other ranks' contigs)
    names = {}
    for k, g in enumerate(mine):
        rows = range(int(res.task_call_off[k]), int(res.task_call_off[k + 1]))
        ids = np.unique(np.concatenate([res.rn(i) for i in rows])) if len(rows) else []
        names[k] = {int(i): tasks[g].qname(int(i)) for i in ids}
    merged = sdist.gather_results(send, lay, [tasks[i].task_id for i in mine], dst=0, names=names)
    again = sdist.gather_results(send, lay, [tasks[i].task_id for i in mine], dst=0,
                                 task_ids_per_rank=[[tasks[i].task_id for i in s] for s in shards])
    if rank == 0:
        assert again.calls.tobytes() == merged.calls.tobytes() and again.alt_pool.tobytes() == merged.alt_pool.tobytes()
        text = vcf_text(cfg, tasks, lambda ti: (merged, merged.task_rows(ti.task_id)))
        # read names through the gathered name tables only
        names_ok = all(merged.names[int(c["task_index"])][int(i)] == tasks[int(c["task_index"])].qname(int(i))
                       for k, c in enumerate(merged.calls) for i in merged.rn(k))
        q.put((text, merged.calls["task_index"].tolist(), merged.calls["pos"].tolist(), int(len(merged.alt_pool)), names_ok))
    else:
        assert merged is None and again is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather_equals_single_process():
    """Rank 0's result after `dist.gather_results` - records, INS ALT bytes, supporting read names, ordered by task id then
    position (`sniffles:544`, `parallel.py:270-271`) - written as VCF text equals the text of a single process over all tasks."""
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import emu.emu as E
    from sniffles_amd import abi, dist as sdist
    from sniffles_amd.config import SnifflesConfig
    E.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    text, task_col, pos_col, alt_bytes, names_ok = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    tasks = _tasks()
    exp_text, res = single_process_text(SnifflesConfig(output_rnames=True), tasks, E.lib(), abi.OUT_EXECUTE)
    assert text == exp_text and text.count("\n") > 20 and "SVTYPE=INS" in text and "RNAMES=" in text
    assert names_ok and alt_bytes == len(res.alt_pool) > 1000
    # ordered by task id, then position
    assert task_col == sorted(task_col)
    assert all(pos_col[i] <= pos_col[i + 1] for i in range(len(pos_col) - 1) if task_col[i] == task_col[i + 1])
    # sharding is a partition and is balanced
    shards = sdist.shard_lpt([t.contig_len for t in tasks], 2)
    assert sorted(shards[0] + shards[1]) == list(range(len(tasks)))
    loads = [sum(tasks[i].contig_len for i in s) for s in shards]
    assert max(loads) <= 1.35 * min(loads)


def _shared_worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import emu.emu as E
    E.lib()                                              # the host tier becomes the library of this test
    from sniffles_amd import abi, dist as sdist, lib
    from sniffles_amd.config import SnifflesConfig
    tasks = _tasks()
    shards = sdist.shard_lpt([t.contig_len for t in tasks], world)
    mine = shards[rank]
    cfg = SnifflesConfig()
    landing = sdist.SharedLanding(slots=2, block_bytes=1 << 21, alt_bytes=1 << 20)
    ids = [tasks[i].task_id for i in mine]
    texts = []
    with lib.Batch(cfg, [tasks[i] for i in mine]) as b:
        b.set_output(abi.OUT_EXECUTE)
        for slot in (0, 1, 0):                               # three passes: both segments, and one of them again
            b.set_result_memory(*landing.memory(slot))
            b.call_candidates(); b.finalize()
            lay = b.fetch_layout()
            merged = sdist.gather_results_shared(landing, slot, lay, ids)
            if rank == 0:
                texts.append(vcf_text(cfg, tasks, lambda ti: (merged, merged.task_rows(ti.task_id))))
            else:
                assert merged is None
            dist.barrier()                                   # (the parent is done with the segments before the next pass writes them)
        # a result that does not fit the caller's memory is refused, never truncated
        small = np.zeros(512, np.uint8)
        b.set_result_memory(small, small.copy())
        b.call_candidates(); b.finalize()
        try:
            b.fetch_layout()
            refused = False
        except lib.SnifflesAmdError as e:
            refused = "does not fit" in str(e)
        b.set_result_memory(None, None)
        b.call_candidates(); b.finalize()
        own = b.fetch(1)
    if rank == 0:
        q.put((texts, refused, int(len(own.calls))))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shared_landing_equals_single_process():
    """`SharedLanding` + `gather_results_shared`: every rank's kernels write the result into a shared-memory segment
    (`Batch.set_result_memory`), rank 0 reads all segments in place - its VCF text equals the single process's, pass after pass;
    memory that is too small is refused; the library's own buffers come back."""
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import emu.emu as E
    from sniffles_amd import abi
    from sniffles_amd.config import SnifflesConfig
    E.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_shared_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    texts, refused, n_own = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    exp_text, _ = single_process_text(SnifflesConfig(), _tasks(), E.lib(), abi.OUT_EXECUTE)
    assert texts == [exp_text] * 3 and "SVTYPE=INS" in exp_text
    assert refused and n_own > 10


def _queue_worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import time
    import emu.emu as E
    E.lib()                                              # the host tier becomes the library of this test
    from sniffles_amd import abi, dist as sdist, lib
    from sniffles_amd.config import SnifflesConfig
    tasks = _tasks()
    cfg = SnifflesConfig()
    queue = sdist.TaskQueue([t.n_leads for t in tasks])
    blocks = []
    for i in queue:                       # one task per claim; rank 1 is slowed down so that rank 0 takes more
        with lib.Batch(cfg, [tasks[i]]) as b:
            b.set_output(abi.OUT_EXECUTE)
            b.call_candidates(); b.finalize()
            blocks.append(sdist.result_block(b.fetch(1)))
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
    # the claims of a rank become one block (task_index = global task id), then the gather
    local = sdist.merge_blocks(blocks, [[tasks[i].task_id] for i in queue.claimed])
    lay, blob = sdist.result_block(local)
    ident = list(range(max(t.task_id for t in tasks) + 1))
    merged = sdist.gather_results(torch.from_numpy(blob), lay, ident, dst=0)
    if rank == 0:
        have = set(int(t) for t in merged.calls["task_index"])
        merged.task_ids = np.asarray(sorted(t.task_id for t in tasks), np.int64)
        merged.task_call_off = np.searchsorted(merged.calls["task_index"], np.concatenate([merged.task_ids, [1 << 30]]))
        text = vcf_text(cfg, tasks, lambda ti: (merged, merged.task_rows(ti.task_id)))
        q.put((text, claims, claims2, claims3, sorted(have)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_task_queue_equals_single_process():
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import emu.emu as E
    from sniffles_amd import abi
    from sniffles_amd.config import SnifflesConfig
    E.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_queue_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    text, claims, claims2, claims3, have = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    tasks = _tasks()
    exp_text, _ = single_process_text(SnifflesConfig(), tasks, E.lib(), abi.OUT_EXECUTE)
    assert text == exp_text and len(have) >= 4
    assert sorted(claims[0] + claims[1]) == list(range(len(tasks)))     # every task exactly once
    assert len(claims[0]) > len(claims[1]) >= 1                           # the faster rank came back for more
    assert sorted(claims2[0] + claims2[1]) == [0, 1, 2, 3]               # a second queue is a fresh queue
    assert (claims2[0] + claims2[1])[0] in (3,) or 3 in (claims2[0][:1] + claims2[1][:1])   # heaviest claimed first
    assert sorted(claims3[0] + claims3[1]) == list(range(40))            # threads x ranks: still every item exactly once


def _sets_worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import time
    import emu.emu as E
    E.lib()                                              # the host tier becomes the library of this test
    from sniffles_amd import abi, dist as sdist, lib
    from sniffles_amd.config import SnifflesConfig
    tasks = _tasks()
    cfg = SnifflesConfig()
    sets = [sorted(g) for g in sdist.shard_lpt([t.n_leads for t in tasks], 3) if g]       # contig SETS: the claimable unit
    set_ids = [[tasks[i].task_id for i in g] for g in sets]
    weights = [sum(tasks[i].n_leads for i in g) for g in sets]
    # every rank holds every set resident (it serves whichever it claims); a segment per (generation, set)
    handles = [lib.Batch(cfg, [tasks[i] for i in g]) for g in sets]
    for b in handles:
        b.set_output(abi.OUT_EXECUTE)
    landing = sdist.SharedLanding(slots=2 * len(sets), block_bytes=1 << 21, alt_bytes=1 << 20)
    texts, served = [], []
    for p in range(3):
        queue = sdist.TaskQueue(weights, key="sets")
        entries = []
        for g in queue:
            slot = (p & 1) * len(sets) + g
            handles[g].set_result_memory(*landing.memory(slot))
            handles[g].call_candidates(); handles[g].finalize()
            entries.append((slot, handles[g].fetch_layout(), g))
            if rank == (p & 1):
                time.sleep(0.5)                           # the slow rank changes from pass to pass: so does who serves what
        served.append(len(entries))
        merged = sdist.gather_sets_shared(landing, entries, len(sets), set_ids)
        if rank == 0:
            assert merged.task_ids.tolist() == sorted(t.task_id for t in tasks)
            texts.append(vcf_text(cfg, tasks, lambda ti: (merged, merged.task_rows(ti.task_id))))
        else:
            assert merged is None
        dist.barrier()
    too_many = False
    try:
        sdist.gather_sets_shared(landing, [(0, dict(n_calls=0, rnames_len=0, alt_pool_len=0, off_rnames=0), 0)] * 4, 3, set_ids)
    except ValueError:
        too_many = True
    counts = sdist.gather_claims(served, world)
    for b in handles:
        b.close()
    if rank == 0:
        q.put((texts, counts, too_many))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_contig_sets_from_a_queue_land_shared_equals_single_process():
    """The strong-scaling shape of `bench.py --scaling strong` (one genome over N ranks, the reference's pull queue,
    `sniffles:495-530`): contig sets claimed from `TaskQueue`, each ONE device batch whose result lands in the rank's shared
    segment; `dist.gather_sets_shared` gives rank 0 the whole genome - its VCF text equals the single process's in every pass,
    whichever rank served which set."""
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import emu.emu as E
    from sniffles_amd import abi
    from sniffles_amd.config import SnifflesConfig
    E.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 36500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_sets_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    texts, counts, too_many = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    exp_text, _ = single_process_text(SnifflesConfig(), _tasks(), E.lib(), abi.OUT_EXECUTE)
    assert texts == [exp_text] * 3 and "SVTYPE=INS" in exp_text
    assert all(a + b == 3 for a, b in zip(counts[0], counts[1]))          # every set exactly once per pass
    assert counts[0] != counts[1] or counts[0][0] != counts[0][1]           # ... served by whoever was free
    assert too_many


def _scatter_worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import emu.emu as E
    E.lib()                                              # the host tier becomes the library of this test
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
    task = parallel.CombineTask(id=7, sv_id=0, contig=exp["contig"], start=0, end=exp["contig_len"], config=cfg)
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
    E.lib()                                              # the host tier becomes the library of this test
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


def _bench_two_ranks(extra, port, **env_extra):
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one rank per device), on this GPU-less box:
    SNF_BENCH_EMU=1 = gloo + CPU tensors + the kernels through the host emulation.  The code that runs is the code of the
    N > 1 GPU run: process group, work queue, result export per pass, `dist.gather_results` on the communication thread."""
    import json
    import subprocess
    env = dict(os.environ, SNF_BENCH_EMU="1", OMP_NUM_THREADS="1", **env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--scale", "0.004",
           "--no-cpu-baseline", "--no-wall-clock"] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_two_ranks_weak_scaling_emu():
    """The default N > 1 line: results into node-shared memory by every rank's own kernels, layouts gathered."""
    d = _bench_two_ranks([], 32500 + (os.getpid() % 500))
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["tasks"] == 48 and d["value"] > 0
    assert d["config"]["gathered_on_rank0"]["ranks"] == 2 and d["config"]["gathered_on_rank0"]["records"] == d["config"]["calls"] > 0
    assert "SharedLanding" in d["config"]["parallelism"]
    assert d["ranks_seen"] == 2 and d["sets_served"] == [4, 4]             # four passes per rank, each over the rank's genome replica
    # the same line carries the strong-scaling value: ONE genome over the two ranks, its sets served by whoever claimed them
    st = d["strong"]
    assert st["scaling"] == "strong" and st["value"] > 0 and st["ranks_seen"] == 2 and sum(st["sets_served"]) == 2 * st["steps"]
    assert st["gathered_on_rank0"]["tasks"] == 24 and st["gathered_on_rank0"]["records"] == st["calls"] > 0


def test_bench_two_ranks_weak_scaling_block_gather_emu():
    """The same line with the result blocks gathered through the process group (`dist.gather_results`: what a multi-node job takes)."""
    d = _bench_two_ranks([], 34500 + (os.getpid() % 500), SNF_BENCH_GATHER="rccl")
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["tasks"] == 48 and d["value"] > 0
    assert d["config"]["gathered_on_rank0"]["ranks"] == 2 and d["config"]["gathered_on_rank0"]["records"] == d["config"]["calls"] > 0
    assert "RCCL gather" in d["config"]["parallelism"]


def test_bench_two_ranks_falls_back_when_shm_is_full_emu():
    """/dev/shm too small for the result segments (a container's 64 MB default): the line is produced over the block gather."""
    d = _bench_two_ranks([], 35500 + (os.getpid() % 500), SNF_BENCH_SHM_FULL="1")
    assert d["n_gpus"] == 2 and d["config"]["gathered_on_rank0"]["records"] == d["config"]["calls"] > 0
    assert "RCCL gather" in d["config"]["parallelism"]


def test_bench_two_ranks_strong_scaling_emu():
    d = _bench_two_ranks(["--scaling", "strong"], 33500 + (os.getpid() % 500))
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["tasks"] == 24 and d["value"] > 0
    assert d["config"]["gathered_on_rank0"]["records"] == d["config"]["calls"] > 0 and d["config"]["gathered_on_rank0"]["tasks"] == 24
    assert "2 contig sets" in d["config"]["parallelism"] and "SharedLanding" in d["config"]["parallelism"]


def test_bench_two_ranks_strong_scaling_block_gather_emu():
    """... and where /dev/shm is not an option (several nodes): the blocks of the sets a rank served, gathered over the process group."""
    d = _bench_two_ranks(["--scaling", "strong"], 37500 + (os.getpid() % 500), SNF_BENCH_GATHER="rccl")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["gathered_on_rank0"]["records"] == d["config"]["calls"] > 0
    assert "RCCL gather" in d["config"]["parallelism"]


def test_bench_population_merge_line_emu():
    """`bench.py --config 4` (the population merge; also a block of the driver's default line) on this GPU-less box: the host tier's
    kernels, a tiny population - the line's plumbing (phases, the merged text against the object path, the handle the records are
    written into) must not rot unseen.  Never a measurement."""
    import json
    import subprocess
    env = dict(os.environ, SNF_BENCH_EMU="1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "4", "--scale", "0.004", "--samples", "3", "--steps", "1", "--warmup", "1"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    c = d["config"]
    assert d["unit"] == "candidates/s" and d["value"] > 0 and c["baseline_config"] == 4 and c["samples"] == 3
    assert c["text_equals_object_path"] is True and c["vcf_bytes"] > 1000 and 0 < c["combined_calls"] <= c["candidates"]
    assert {"walk_blocks", "sort_and_windows", "resolve_groups_gpu", "build_svcalls"} <= set(c["host_phases_ms"])
    # ... and the checks of the line: the group assignment against the C oracle, every merged record against the text the unmodified
    # reference's own writer prints for its CombineTask.execute on the same population (where the reference is staged)
    assert d["verified"] is True and c["verified"] is True
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_combine_pool
    if ref_combine_pool.available():
        v = d["verified_vs_reference"]
        assert v["ok"] is True and v["differences"] == [] and v["records_compared"] == c["combined_calls"] > 50 and v["contigs_compared"] == 24
        assert c["verified_vs_reference"] is True
