"""TEST INFRASTRUCTURE ONLY - the C oracle (oracle/snf_oracle.c) over a whole workload on all host cores.

The reference parallelises one process per contig task (`sniffles:495-530`, `parallel.py:652-680`: a pool of worker
processes pulling whole contigs), so its parallelism is capped by the number of contigs.  This module restates that
schedule for bench.py's `cpu_baseline` leg and `--verify`: P = min(host cores, tasks) worker processes, tasks
assigned longest-first, every process generates its own seeded inputs, all wait on a barrier, then run the oracle on
their tasks one after another.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline / verify legs may
import it; nothing under sniffles_amd/ does.

Workers are started with the `spawn` method: the parent (bench.py) holds a HIP context, which must not be forked.
"""
from __future__ import annotations

import multiprocessing as mp
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(wid, specs, cfg_kw, want_results, barrier, out_q):
    try:
        for p in (ROOT, HERE):
            if p not in sys.path:
                sys.path.insert(0, p)
        import oracle
        from sniffles_amd import synth
        from sniffles_amd.config import SnifflesConfig
        cfg = SnifflesConfig(**cfg_kw)
        tis = [(key, synth.gen_task(**kw)) for key, kw in specs]
        oracle.lib()
        barrier.wait(timeout=600)
        t_start = time.time()
        for key, ti in tis:
            t0 = time.perf_counter()
            res = oracle.run(cfg, [ti], True)
            wall = time.perf_counter() - t0
            item = dict(key=key, worker=wid, n_leads=ti.n_leads, n_reads=ti.n_reads, hot_s=oracle.hot_seconds(), wall_s=wall,
                        t_start=t_start, t_end=time.time(), n_calls=int(res.calls.shape[0]))
            if want_results:
                item["result"] = res
            out_q.put(item)
        out_q.put(dict(done=wid))
    except BaseException as e:  # noqa: BLE001 - reported to the parent
        import traceback
        out_q.put(dict(error=f"worker {wid}: {e!r}\n{traceback.format_exc()}"))


def run_tasks(specs: list, cfg_kw: dict, weights=None, want_results: bool = True, max_procs: int = None) -> dict:
    """specs: [(key, kwargs of synth.gen_task)], one per contig task.  Returns
    {items: {key: {...}}, procs, cores, wall_s (barrier -> last task done, dense coverage build included),
     hot_all_core_s (slowest process' call_candidates + finalize seconds), hot_single_core_s (sum over tasks)}."""
    import oracle
    oracle.build()   # once, before the workers race for it
    cores = os.cpu_count() or 1
    n = len(specs)
    procs = max(1, min(cores, n, max_procs or n))
    weights = list(weights) if weights is not None else [1] * n
    order = sorted(range(n), key=lambda i: (-weights[i], i))
    shards, load = [[] for _ in range(procs)], [0] * procs
    for i in order:
        r = min(range(procs), key=lambda k: (load[k], k))
        load[r] += weights[i]
        shards[r].append(specs[i])
    ctx = mp.get_context("spawn")
    barrier, q = ctx.Barrier(procs), ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(w, shards[w], cfg_kw, want_results, barrier, q), daemon=True) for w in range(procs)]
    for p in ps:
        p.start()
    items, done, err = {}, 0, None
    import queue as _queue
    while done < procs and err is None:
        try:
            m = q.get(timeout=5)
        except _queue.Empty:
            dead = [p.exitcode for p in ps if p.exitcode not in (None, 0)]
            if dead:   # a worker died before it could report (import error, killed)
                err = f"oracle worker exited with code {dead[0]}"
            continue
        if "error" in m:
            err = m["error"]
        elif "done" in m:
            done += 1
        else:
            items[m["key"]] = m
    for p in ps:
        if err is not None:
            p.terminate()
        p.join(timeout=30)
    if err is not None:
        raise RuntimeError(err)
    per_proc = {}
    for m in items.values():
        per_proc[m["worker"]] = per_proc.get(m["worker"], 0.0) + m["hot_s"]
    return dict(items=items, procs=procs, cores=cores,
                wall_s=max(m["t_end"] for m in items.values()) - min(m["t_start"] for m in items.values()),
                hot_all_core_s=max(per_proc.values()), hot_single_core_s=sum(m["hot_s"] for m in items.values()))
