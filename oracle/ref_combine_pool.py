"""TEST INFRASTRUCTURE ONLY - the UNMODIFIED reference's `CombineTask.execute` (`/root/reference/src/sniffles/parallel.py:444-572`) over a
synthetic population on the host cores of this box: the `cpu_baseline` of bench.py --config 4 with kind "reference (edlib stand-in)".

One OS process per contig task (the reference's own schedule: workers pull whole contigs).  A worker builds its contig's samples with the
reference's own calling path (`Task.call_candidates` + `finalize_candidates` through `ref_harness.build_task`), stores the candidates into
the reference's SNF blocks (`SNFile.store` / `annotate_block_coverages`; only the gzip / pickle file layer is replaced by an in-memory
stand-in - all untimed: that is the samples' own runs), waits on a barrier and then runs `CombineTask.execute` on them: the timed part.
`sv.align` (edlib, absent from this image) is patched to `oracle.edit_distance_myers` - the bit-parallel algorithm edlib implements, in C,
pinned to the exact DP - so that the reference is not charged an O(nm) DP it would not run.  **Parity unpinned** against edlib itself.

The reference comes from its checkout when that exists, else from the byte-compiled staged build `oracle/_ref` (`oracle/make_ref.py`).
"""
from __future__ import annotations

import multiprocessing as mp
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(wid, contigs, n_samples, coverage, barrier, out_q, want_records):
    try:
        for p in (ROOT, HERE):
            if p not in sys.path:
                sys.path.insert(0, p)
        import oracle as oc
        import ref_harness as rh
        from sniffles_amd import synth
        oc.build()
        rh.load_reference()
        prepared = []
        for ci, c, L in contigs:
            tis = [synth.gen_task(ci, c, L, coverage, seed=100 + s, site_seed=501) for s in range(n_samples)]      # = tools/bench_population.py
            prepared.append((ci, tis))
        out = []
        first = [True]

        def wait_once():
            if first[0]:
                first[0] = False
                barrier.wait(timeout=7200)
        # (the samples of a contig are called right before its merge: the barrier is taken before the FIRST merge of a worker, once all
        #  its other contigs' samples exist too would cost memory - a worker holds one contig in the usual one-process-per-contig layout)
        for ci, tis in prepared:
            timing = {}
            doc, calls, cfg = rh.run_reference_combine_task(tis, (), align="myers", before_execute=wait_once, timing=timing, task_id=ci,
                                                            with_objects=True)
            item = dict(key=ci, worker=wid, execute_s=timing["execute_s"], candidates=timing["candidates"], combined=len(doc["calls"]))
            if want_records:
                # the merged records as the reference's OWN writer prints them (vcf.py:216-350), in the order its result object emits a
                # task's calls (result.py:137-149: sorted by position, stable): what bench.py --config 4 diffs its text against
                item["vcf"] = rh.reference_vcf_records(sorted(calls, key=lambda c: c.pos), cfg)
            out.append(item)
        for item in out:
            out_q.put(item)
        out_q.put(dict(done=wid))
    except BaseException as e:  # noqa: BLE001 - reported to the parent
        import traceback
        out_q.put(dict(error=f"reference combine worker {wid}: {e!r}\n{traceback.format_exc()}"))


def available() -> bool:
    if HERE not in sys.path:
        sys.path.insert(0, HERE)
    import make_ref
    return make_ref.ref_root() is not None


def run(contigs: list, n_samples: int, coverage: float, want_records: bool = False, max_procs: int = None) -> dict:
    """contigs: [(contig index, name, length)].  Returns {items: {ci: {...}}, procs, cores, hot_all_core_s (slowest process' CombineTask.execute
    seconds), hot_single_core_s, total_wall_s}."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    n = len(contigs)
    procs = max(1, min(cores, n, max_procs or n))
    order = sorted(range(n), key=lambda i: (-contigs[i][2], i))
    shards, load = [[] for _ in range(procs)], [0] * procs
    for i in order:
        r = min(range(procs), key=lambda k: (load[k], k))
        load[r] += contigs[i][2]
        shards[r].append(contigs[i])
    ctx = mp.get_context("spawn")
    barrier, q = ctx.Barrier(procs), ctx.Queue()
    t0 = time.perf_counter()
    ps = [ctx.Process(target=_worker, args=(w, shards[w], n_samples, coverage, barrier, q, want_records), daemon=True) for w in range(procs)]
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
                err = f"reference combine worker exited with code {dead[0]}"
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
    per = {}
    for m in items.values():
        per[m["worker"]] = per.get(m["worker"], 0.0) + m["execute_s"]
    return dict(items=items, procs=procs, cores=cores, total_wall_s=time.perf_counter() - t0, hot_all_core_s=max(per.values()),
                hot_single_core_s=sum(m["execute_s"] for m in items.values()), candidates=sum(m["candidates"] for m in items.values()),
                combined=sum(m["combined"] for m in items.values()))
