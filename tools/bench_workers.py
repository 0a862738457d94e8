#!/usr/bin/env python3
"""The per-task seam in the reference's deployment shape: P worker PROCESSES that share ONE MI355X, each working through its contig
tasks with `Task.call_candidates` + `Task.finalize_candidates` (sniffles_amd.parallel - the reference's worker loop, `sniffles:495-530`,
`parallel.py:652-717`, calls exactly these two per task) or with the one-step `CallTask.execute_calls`.

The mirror image of oracle/ref_pool.py, which times the unmodified reference the same way: spawned workers (the parent holds a HIP
context), contigs assigned longest-first, every worker builds its inputs BEFORE a barrier - the seeded signature tables and, for
the `leads` input form, the `Lead` objects fed through `LeadProvider.record_lead` / `record_read` (extraction's work, untimed on
both sides) - then runs its tasks; the figure is the slowest worker's time over its tasks.  Input forms:
  columns   the lead provider already holds typed columns (what this package's extraction produces): no object walk
  leads     a LeadProvider filled with Lead objects through `record_lead`, as a ported `iter_region` would fill it (a lead becomes a row
            of typed columns when it is recorded - where the reference pays its binning); `call_candidates` starts with `to_task_input`
            (a copy of the finished columns + the string-rank remap) - its share is reported
Every worker keeps two tasks in flight (`Task.prepare`: task k + 1 uploads and runs while task k's records become objects).

    python tools/bench_workers.py [P ...]          (default 4 8 24)
"""
from __future__ import annotations

import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(wid, specs, cfg_kw, form, shape, device, barrier, out_q, hw_queues=0, rehearsal=None):
    try:
        if ROOT not in sys.path:
            sys.path.insert(0, ROOT)
        # hardware queues this process may open (read by the HIP runtime when it starts): P processes x their streams
        if hw_queues:
            # must not oversubscribe the device's queues, or the driver time-slices them
            os.environ["GPU_MAX_HW_QUEUES"] = str(hw_queues)
        from sniffles_amd import leadprov, lib, parallel, pipeline, synth
        from sniffles_amd.config import SnifflesConfig
        # tests only: the plumbing of this file on a GPU-less box (host tier of the test suite)
        if os.environ.get("SNF_BENCH_EMU") == "1":
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import emu.emu as E
            E.lib()
        cfg = SnifflesConfig(**cfg_kw)
        built = []
        for key, kw in specs:
            ti = synth.gen_task(**kw)
            if form == "leads":
                lp = leadprov.LeadProvider(cfg, 0, ti.contig, contig_len=ti.contig_len)
                for ld in leadprov.iter_leads(ti):
                    lp.record_lead(ld, 0)
                for a, b, c in zip(ti.read_start.tolist(), ti.read_end.tolist(), ti.read_hp.tolist()):
                    lp.record_read(a, b, c)
            else:
                lp = pipeline._Extracted(ti)
            built.append((key, ti, lp))
        # process warm-up: HIP context, the pinned arena, a device slab of this worker's largest task - or, through a GPU server
        # (SNF_GPU_SERVER: this process never opens the device), the connection, the worker's input segment and the server's own arenas
        if built and os.environ.get("SNF_GPU_SERVER"):
            key, big, lp_big = max(built, key=lambda x: x[1].n_leads)
            wt = parallel.CallTask(id=big.task_id, sv_id=0, contig=big.contig, start=0, end=big.contig_len, config=cfg, device=device)
            wt.lead_provider = pipeline._Extracted(big)
            cfg.qc_nm_threshold = big.qc_nm_threshold
            wt.execute_calls(cfg); wt.close()
        elif built:
            big = max(built, key=lambda x: x[1].n_leads)[1]
            with lib.Batch(cfg, [big], device=device) as b:
                b.run_pass(); b.fetch(1, copy=False)
        if rehearsal is not None and os.environ.get("SNF_GPU_SERVER"):
            # through a GPU server the steady state is a SERVER that has seen batches of this size before (its staging arena and result
            # segments grow with the largest batch so far - a pinned allocation of ~0.15 s that also stalls its other threads): one
            # untimed round of the same calls from a common start, as a long-running server would have behind it
            rehearsal.wait(timeout=3600)
            for key, ti, lp in built:
                wt = parallel.CallTask(id=ti.task_id, sv_id=0, contig=ti.contig, start=0, end=ti.contig_len, config=cfg, device=device)
                wt.lead_provider = lp
                cfg.qc_nm_threshold = ti.qc_nm_threshold
                wt.call_svs(cfg); wt.close()
        barrier.wait(timeout=3600)
        t_all0 = time.perf_counter()
        tasks = []
        for key, ti, lp in built:
            # (the object form carries the task's tandem-repeat regions as the reference's Task does; the column form has them inside)
            trs = list(zip(ti.tr_start.tolist(), ti.tr_end.tolist())) if form == "leads" and ti.tr_start is not None else None
            t = parallel.CallTask(id=ti.task_id, sv_id=0, contig=ti.contig, start=0, end=ti.contig_len, config=cfg, tandem_repeats=trs,
                                  device=device)
            t.lead_provider = lp
            tasks.append(t)
        ex = True if shape == "execute" else None
        ingest_s = 0.0
        if form == "leads":      # the object walk alone, for the share (the timed loop below does it again inside the calls)
            t0 = time.perf_counter()
            for key, ti, lp in built:
                lp.to_task_input(ti.task_id, 0, None if ti.tr_start is None else list(zip(ti.tr_start.tolist(), ti.tr_end.tolist())),
                                 ti.qc_nm_threshold)
            ingest_s = time.perf_counter() - t0
            t_all0 = time.perf_counter()
        n_out = 0
        served = []

        # (the reference's build_leadtab leaves the task's NM threshold on the config: the lead provider's columns take it from there)
        def prepare(k):
            cfg.qc_nm_threshold = built[k][1].qc_nm_threshold
            tasks[k].prepare(cfg, execute=ex)
        if tasks:
            prepare(0)
        for k, t in enumerate(tasks):
            if k + 1 < len(tasks):
                prepare(k + 1)
            if shape == "execute":
                n_out += len(t.execute_calls(cfg))
            else:
                # the two calls and what CallTask.execute does with their result (parallel.py:265-271): `[s for s in svcalls if s.qc]`,
                # `sorted(key=pos)` - every candidate's `qc` is read, every kept call is a finished object when the loop moves on
                n_out += len(t.call_svs(cfg))
            if getattr(t, "served_timing", None):
                served.append(dict(t.served_timing, leads=built[k][1].n_leads))
            t.close()
        hot = time.perf_counter() - t_all0
        out_q.put(dict(worker=wid, hot_s=hot, ingest_s=ingest_s, n_out=n_out, tasks=len(tasks), leads=sum(x[1].n_leads for x in built), served=served))
    except BaseException as e:  # noqa: BLE001 - reported to the parent
        import traceback
        out_q.put(dict(error=f"worker {wid}: {e!r}\n{traceback.format_exc()}"))


def run(specs: list, cfg_kw: dict, procs: int, form: str = "columns", shape: str = "api", device: int = 0, weights=None,
        hw_queues: int = 0, gpu_server: bool = False) -> dict:
    if gpu_server:       # ONE process on the device (sniffles_amd.server), the workers hand their tasks over in shared memory
        from sniffles_amd import server
        srv = server.start(device=device)
        os.environ["SNF_GPU_SERVER"] = srv.address           # (spawned workers inherit it)
        try:
            out = run(specs, cfg_kw, procs, form, shape, device, weights, 0, False)
            out["gpu_server"] = True
            return out
        finally:
            os.environ.pop("SNF_GPU_SERVER", None)
            srv.stop()
    """specs: [(key, kwargs of synth.gen_task)].  Returns {procs, hot_all_s (slowest worker), hot_sum_s, ingest_all_s (slowest worker's
    object walk, `leads` form), n_out, setup_wall_s}."""
    n = len(specs)
    procs = max(1, min(procs, n))
    weights = list(weights) if weights is not None else [kw.get("contig_len", 1) for _, kw in specs]
    order = sorted(range(n), key=lambda i: (-weights[i], i))
    shards, load = [[] for _ in range(procs)], [0] * procs
    for i in order:
        r = min(range(procs), key=lambda k: (load[k], k))
        load[r] += weights[i]
        shards[r].append(specs[i])
    ctx = mp.get_context("spawn")
    barrier, q = ctx.Barrier(procs + 1), ctx.Queue()
    rehearsal = ctx.Barrier(procs)
    t0 = time.perf_counter()
    ps = [ctx.Process(target=_worker, args=(w, shards[w], cfg_kw, form, shape, device, barrier, q, hw_queues, rehearsal),
                      daemon=True) for w in range(procs)]
    for p in ps:
        p.start()
    import queue as _queue
    err = None
    try:
        barrier.wait(timeout=1800)            # every worker holds its inputs and a warm device context
    except Exception as e:  # noqa: BLE001
        err = f"barrier: {e!r}"
    t1 = time.perf_counter()
    got = []
    while len(got) < procs and err is None:
        try:
            m = q.get(timeout=5)
        except _queue.Empty:
            dead = [p.exitcode for p in ps if p.exitcode not in (None, 0)]
            if dead:
                err = f"worker exited with code {dead[0]}"
            continue
        if "error" in m:
            err = m["error"]
        else:
            got.append(m)
    wall = time.perf_counter() - t1
    for p in ps:
        if err is not None:
            p.terminate()
        p.join(timeout=30)
    if err is not None:
        raise RuntimeError(err)
    return dict(procs=procs, form=form, shape=shape, hw_queues_per_process=hw_queues or None, gpu_server=bool(os.environ.get("SNF_GPU_SERVER")),
                hot_all_ms=round(max(m["hot_s"] for m in got) * 1e3, 1), wall_ms=round(wall * 1e3, 1),
                hot_sum_ms=round(sum(m["hot_s"] for m in got) * 1e3, 1), ingest_all_ms=round(max(m["ingest_s"] for m in got) * 1e3, 1),
                n_out=sum(m["n_out"] for m in got), setup_wall_s=round(t1 - t0, 1),
                # through a GPU server: the slowest worker's calls split into task input / packing / hand-over + server batch
                slowest_worker_served=[{k: (round(v, 2) if isinstance(v, float) else v) for k, v in d.items()}
                                       for d in max(got, key=lambda m: m["hot_s"]).get("served", [])][:6] or None)


def genome_specs(coverage=30.0, scale=1.0, gen=None):
    sys.path.insert(0, ROOT)
    from sniffles_amd import synth
    return [(ci, dict(task_id=ci, contig=c, contig_len=max(200000, int(synth.GRCH38[c] * scale)), coverage=coverage, seed=1, **(gen or {})))
            for ci, c in enumerate(synth.CONTIGS)]


if __name__ == "__main__":
    import json
    P = [int(x) for x in sys.argv[1:] if not x.startswith("q")] or [4, 8, 24]
    Q = [int(x[1:]) for x in sys.argv[1:] if x.startswith("q")] or [0]      # q2: GPU_MAX_HW_QUEUES=2 in every worker
    specs = genome_specs()
    for p in P:
        for hq in Q:
            for form, shape in (("columns", "api"), ("columns", "execute"), ("leads", "api")):
                print(json.dumps(run(specs, {}, p, form, shape, hw_queues=hq)), flush=True)
