"""TEST INFRASTRUCTURE ONLY - the C oracle's resolve_block_groups (oracle/snf_oracle.c, cluster.py:356-390 with the exact
edit-distance DP in place of edlib) over a sample of flush windows on all host cores.  Used by the cpu_baseline / verify
leg of bench.py --config 4 (tools/bench_population.py) and by tests; nothing under sniffles_amd/ imports it.

A window travels as plain data: (svtype, [(pos, svlen, support, sample_id, mate_contig, mate_ref_start, alt bytes), ...]).
Workers are spawned (the parent holds a HIP context)."""
from __future__ import annotations

import multiprocessing as mp
import os
import sys
import time
from types import SimpleNamespace

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def window_of(svtype, cands) -> tuple:
    return (svtype, [(c.pos, c.svlen, c.support, c.sample_internal_id,
                      c.bnd_info.mate_contig if c.bnd_info is not None else None,
                      c.bnd_info.mate_ref_start if c.bnd_info is not None else 0,
                      c.alt.encode("latin-1") if isinstance(c.alt, str) else bytes(c.alt)) for c in cands])


def _cands(rows):
    out = []
    for pos, svlen, support, sid, mc, mp_, alt in rows:
        bi = SimpleNamespace(mate_contig=mc, mate_ref_start=mp_) if mc is not None else None
        out.append(SimpleNamespace(pos=pos, svlen=svlen, support=support, sample_internal_id=sid, bnd_info=bi, alt=alt))
    return out


def _worker(wid, windows, cfg_kw, n_samples, out_q):
    try:
        for p in (ROOT, HERE):
            if p not in sys.path:
                sys.path.insert(0, p)
        import oracle
        from sniffles_amd import cluster
        from sniffles_amd.config import SnifflesConfig
        cfg = SnifflesConfig(**cfg_kw)
        cfg.snf_input_info = [dict(internal_id=s) for s in range(n_samples)]
        cfg.mode = "combine"
        oracle.lib()
        res, t = [], 0.0
        for key, (svtype, rows) in windows:
            keep = []
            q, out = cluster.pack_problem(svtype, _cands(rows), [], keep)
            t0 = time.perf_counter()
            oracle.combine_resolve(cfg, q)
            t += time.perf_counter() - t0
            res.append((key, out[:len(rows)].tolist()))
        out_q.put(dict(worker=wid, results=res, seconds=t))
    except BaseException as e:  # noqa: BLE001 - reported to the parent
        import traceback
        out_q.put(dict(error=f"worker {wid}: {e!r}\n{traceback.format_exc()}"))


def run_windows(windows: list, cfg_kw: dict, n_samples: int, max_procs: int = None) -> dict:
    """windows: [(key, window_of(...))].  Returns {groups: {key: [group per candidate]}, procs, cores, slowest_s, sum_s}."""
    import queue as _queue

    import oracle
    oracle.build()
    cores = os.cpu_count() or 1
    procs = max(1, min(cores, len(windows), max_procs or 64))
    # round-robin by descending size so that the processes finish together
    order = sorted(range(len(windows)), key=lambda i: -sum(len(r[6]) for r in windows[i][1][1]))
    shards = [[windows[i] for i in order[w::procs]] for w in range(procs)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(w, shards[w], cfg_kw, n_samples, q), daemon=True) for w in range(procs)]
    for p in ps:
        p.start()
    groups, secs, err, done = {}, [], None, 0
    while done < procs and err is None:
        try:
            m = q.get(timeout=5)
        except _queue.Empty:
            dead = [p.exitcode for p in ps if p.exitcode not in (None, 0)]
            if dead:
                err = f"oracle worker exited with code {dead[0]}"
            continue
        if "error" in m:
            err = m["error"]
        else:
            done += 1
            secs.append(m["seconds"])
            groups.update(dict(m["results"]))
    for p in ps:
        if err is not None:
            p.terminate()
        p.join(timeout=30)
    if err is not None:
        raise RuntimeError(err)
    return dict(groups=groups, procs=procs, cores=cores, slowest_s=max(secs), sum_s=sum(secs))
