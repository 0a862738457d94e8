"""TEST INFRASTRUCTURE ONLY - the UNMODIFIED reference (`sniffles.parallel.Task.call_candidates` + `finalize_candidates`,
`/root/reference/src/sniffles/parallel.py:104-201`) over a whole workload on the host cores of this box.

This is bench.py's `cpu_baseline` leg with `kind = "reference"` (SURVEY.md 8d "CPU baseline timing"): the same seeded
signature tables the GPU pass works on are turned into the reference's own `Lead` objects and fed through the reference's own
`LeadProvider.record_lead` / `record_hap_ref` and dense `uint16` coverage vector (`ref_harness.build_task`, untimed: that is
extraction, excluded on both sides); then every process waits on a barrier and runs the two reference functions on its contig
tasks.  One OS process per contig task, pool = min(host cores, tasks) - the reference's own schedule (`sniffles:495-530`,
`parallel.py:652-717`: workers pull whole contigs; its parallelism is capped by the number of contigs).

The reference is imported from its checkout when that exists, else from the byte-compiled staged build `oracle/_ref`
(`oracle/make_ref.py`), which is how it reaches the GPU box.  Workers are spawned (the parent holds a HIP context).

With `want_results` a worker also returns what `CallTask.execute` would send to the parent (`parallel.py:265-271`:
`[s for s in svcalls if s.qc]`, `sorted(key=pos)`) as canonical records (`ref_harness.call_record`), so the caller can
compare the GPU's execute-mode block with the reference ITSELF on the bench workload.
"""
from __future__ import annotations

import multiprocessing as mp
import os
import pickle
import sys
import time
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(wid, specs, extra_args, want_results, barrier, out_q):
    try:
        for p in (ROOT, HERE):
            if p not in sys.path:
                sys.path.insert(0, p)
        import ref_harness as rh
        from sniffles_amd import synth
        rh.load_reference()
        built = []
        t_b0 = time.perf_counter()
        for key, kw in specs:
            ti = synth.gen_task(**kw)
            cfg = rh.make_config(tuple(extra_args), ti.qc_nm_threshold)
            built.append((key, ti, cfg, rh.build_task(ti, cfg)))
        build_s = time.perf_counter() - t_b0
        barrier.wait(timeout=3600)
        for key, ti, cfg, task in built:
            # CallTask.execute (parallel.py:255-271) with the default options: qc = True
            t0 = time.perf_counter()
            cands = task.call_candidates(True, cfg)
            t1 = time.perf_counter()
            svcalls = task.finalize_candidates(cands, False, cfg)
            t2 = time.perf_counter()
            item = dict(key=key, worker=wid, n_leads=int(ti.n_leads), n_reads=int(ti.n_reads), call_s=t1 - t0, final_s=t2 - t1,
                        hot_s=t2 - t0, candidates=len(cands), build_s=build_s / max(1, len(specs)))
            if not cfg.no_qc:
                svcalls = [s for s in svcalls if s.qc]
            if cfg.sort:
                svcalls = sorted(svcalls, key=lambda s: s.pos)
            item["kept"] = len(svcalls)
            item["coverage_average_total"] = float(task.coverage_average_total)
            if want_results:
                recs = [rh.call_record(c, "final") for c in svcalls]
                item["records_z"] = zlib.compress(pickle.dumps(recs, protocol=4), 1)
            out_q.put(item)
        out_q.put(dict(done=wid))
    except BaseException as e:  # noqa: BLE001 - reported to the parent
        import traceback
        out_q.put(dict(error=f"reference worker {wid}: {e!r}\n{traceback.format_exc()}"))


def available() -> bool:
    if HERE not in sys.path:
        sys.path.insert(0, HERE)
    import make_ref
    return make_ref.ref_root() is not None


def kind() -> str:
    if HERE not in sys.path:
        sys.path.insert(0, HERE)
    import make_ref
    r = make_ref.ref_root()
    return "absent" if r is None else ("checkout" if r == make_ref.SRC_ROOT else "staged build oracle/_ref (byte-compiled by oracle/make_ref.py)")


def run_tasks(specs: list, extra_args=(), weights=None, want_results: bool = False, max_procs: int = None) -> dict:
    """specs: [(key, kwargs of synth.gen_task)], one per contig task.  Returns
    {items: {key: {...}}, procs, cores, hot_all_core_s (slowest process' call_candidates + finalize seconds = the wall clock of the
     pool once every process holds its lead tables), hot_single_core_s (sum over tasks), build_wall_s}."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
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
    t_start = time.perf_counter()
    ps = [ctx.Process(target=_worker, args=(w, shards[w], tuple(extra_args), want_results, barrier, q), daemon=True) for w in range(procs)]
    for p in ps:
        p.start()
    items, done, err = {}, 0, None
    import queue as _queue
    while done < procs and err is None:
        try:
            m = q.get(timeout=5)
        except _queue.Empty:
            dead = [p.exitcode for p in ps if p.exitcode not in (None, 0)]
            if dead:
                err = f"reference worker exited with code {dead[0]}"
            continue
        if "error" in m:
            err = m["error"]
        elif "done" in m:
            done += 1
        else:
            if "records_z" in m:
                m["records"] = pickle.loads(zlib.decompress(m.pop("records_z")))
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
    return dict(items=items, procs=procs, cores=cores, total_wall_s=time.perf_counter() - t_start,
                hot_all_core_s=max(per_proc.values()), hot_single_core_s=sum(m["hot_s"] for m in items.values()),
                build_single_core_s=sum(m["build_s"] for m in items.values()))
