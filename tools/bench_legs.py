"""bench.py's legs outside the timed region: the wall clock of one genome through the drop-in boundary (batched, per task, worker
processes), the other BASELINE.json configs, the CPU baselines (the unmodified reference `oracle/_ref`, the C oracle) and the checks of the
bench workload's results against both."""
from __future__ import annotations

import json
import os
import sys
import time

from tools.bench_common import EMU, REFERENCE_CPYTHON, ROOT, WORKLOADS, dev_sync, set_dev, task_specs

def bind_to_gpu_numa(torch, local_rank):
    """One process per GPU, bound to the CPUs of the NUMA node the GPU hangs off (what a launcher does for every rank): the
    host threads that drive the batches, their pinned buffers and the staging arena then sit next to the PCIe root of the
    device.  (Measured on the 2-socket GPU box: no difference for one rank - four runs each 2.10-2.41 ms unbound, 2.12-2.30
    bound; the run-to-run spread of ~10 % has another cause.  Kept for the N-rank launches.)  Best effort (sysfs); SNF_BENCH_NO_NUMA=1
    turns it off.  Returns what was done, for the output line."""
    if os.environ.get("SNF_BENCH_NO_NUMA") == "1":
        return "off"
    try:
        p = torch.cuda.get_device_properties(local_rank)
        bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        if node < 0:
            return "gpu has no numa node"
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return f"node {node}: no allowed cpu"
        os.sched_setaffinity(0, cpus)
        return f"node {node} ({len(cpus)} cpus)"
    except Exception as e:  # noqa: BLE001 - measurement hygiene only
        return f"unavailable ({type(e).__name__})"


def worker_processes(specs, cfg_kw, device):
    """The per-task seam in the reference's deployment shape (tools/bench_workers.py): P worker processes share this GPU, each runs
    Task.call_candidates + finalize_candidates over its contigs (longest first), two tasks in flight per worker; `leads`: the lead
    providers hold Lead objects (the one walk that turns them into columns is inside call_candidates), `columns`: typed columns."""
    from tools import bench_workers
    out = {}
    # (processes, input form, call shape, hardware queues per process): from eight processes on every worker is held to two hardware
    # queues - P x (four streams each) would oversubscribe the device's queues and the driver would time-slice them (same box:
    # P = 24 columns / api 720 ms with two queues each against 1 219 ms)
    # ... and P = 24 - the pool size the reference's CPU baseline uses - through ONE GPU server process (sniffles_amd.server): two dozen
    # processes that each open the device are time-sliced by the driver (0.4-0.9 s for the genome, whatever bounds the passes in flight)
    plan = [(4, "columns", "api", 0), (4, "leads", "api", 0), (8, "columns", "api", 2), (8, "leads", "api", 2),
            (24, "columns", "api", "server"), (24, "leads", "api", "server"), (24, "columns", "api", 2)]
    if os.environ.get("SNF_BENCH_WORKERS"):      # e.g. "8" or "4,24"
        want = {int(x) for x in os.environ["SNF_BENCH_WORKERS"].split(",") if x}
        plan = [p for p in plan if p[0] in want]
    for procs, form, shape, hq in plan:
        key = f"P{procs}_{form}_{shape}" + ("_server" if hq == "server" else "")
        try:
            if hq == "server":
                out[key] = bench_workers.run(specs, cfg_kw, procs, form, shape, device, gpu_server=True)
                continue
            out[key] = bench_workers.run(specs, cfg_kw, procs, form, shape, device, hw_queues=hq)
        except Exception as e:  # noqa: BLE001 - an extra measurement must not take the line down
            out[key] = f"failed: {type(e).__name__}: {str(e)[:300]}"
    out["note"] = ("hot_all_ms = the slowest worker's time over its tasks from a common barrier (inputs built and device context warm "
                   "before it, as "
                   "oracle/ref_pool.py times the reference); ingest_all_ms = that worker's Lead objects -> columns walk alone; one "
                   "MI355X shared by all workers")
    return out


def wall_clock(cfg, tasks, device, specs=None, cfg_kw=None):
    """One genome end to end through the drop-in boundary, outside the timed region (milliseconds): `batched` = all
    contig tasks in one device batch (the library's native shape); `per_task_api` = the reference's own call sequence,
    Task.call_candidates + Task.finalize_candidates task by task, SVCall objects out (sniffles_amd.parallel)."""
    from sniffles_amd import lib, parallel, pipeline, sv
    os.environ["SNF_PROF"] = "1"          # the library prints its own split of the upload to stderr
    t0 = time.perf_counter()
    from sniffles_amd import abi
    b = lib.Batch(cfg, tasks, device=device)
    t1 = time.perf_counter()
    del os.environ["SNF_PROF"]
    # what the reference's workers hand to the parent - CallTask.execute's result (parallel.py:264-271): the QC-passing calls of
    # every task sorted by position, filtered and ordered on the device; those (26.8 k of the 94 k candidates) become objects
    b.set_output(abi.OUT_EXECUTE)
    b.call_candidates(); b.finalize(); b.sync()
    t2 = time.perf_counter()
    res = b.fetch(1, copy=False)          # views of the library's pinned result block, as sniffles_amd.parallel.Task reads them
    t3 = time.perf_counter()
    n = 0
    for t, ti in enumerate(tasks):
        lo, hi = int(res.task_call_off[t]), int(res.task_call_off[t + 1])
        calls = sv.materialize_candidates(res, ti, lo, hi)
        sv.apply_final(calls, res, ti, lo)
        for c in calls:
            c.finalize()
        n += len(calls)
    t4 = time.perf_counter()
    # for the record: every candidate as an object (what Task.finalize_candidates returns with keep_qc_fails - the --snf shape)
    b.set_output(abi.OUT_CANDIDATES)
    b.call_candidates(); b.finalize(); b.sync()
    res_all = b.fetch(1, copy=False)
    tc = time.perf_counter()
    n_all = 0
    for t, ti in enumerate(tasks):
        lo, hi = int(res_all.task_call_off[t]), int(res_all.task_call_off[t + 1])
        calls = sv.materialize_candidates(res_all, ti, lo, hi)
        sv.apply_final(calls, res_all, ti, lo)
        n_all += len(calls)
    all_ms = (time.perf_counter() - tc) * 1e3
    del calls
    b.set_output(abi.OUT_EXECUTE)
    b.call_candidates(); b.finalize(); b.sync()
    res = b.fetch(1, copy=False)
    # the same records as VCF text without the objects in between (vcf.VCF.write_records, the BAM -> VCF flow with objects=False)
    vcf_ms, vcf_bytes = None, None
    try:
        import io
        import numpy as np
        from sniffles_amd import vcf
        if not getattr(cfg, "sample_ids_vcf", None):
            cfg.sample_ids_vcf = [(0, "SAMPLE")]
        buf = io.StringIO()
        w = vcf.VCF(cfg, buf)
        if w.can_write_records():
            tv = time.perf_counter()
            for t, ti in enumerate(tasks):
                lo, hi = int(res.task_call_off[t]), int(res.task_call_off[t + 1])
                keep = lo + np.flatnonzero(res.calls["qc"][lo:hi] != 0)
                keep = keep[np.argsort(res.calls["pos"][keep], kind="stable")]
                w.write_records(res, ti, keep)
            vcf_ms, vcf_bytes = round((time.perf_counter() - tv) * 1e3, 2), len(buf.getvalue())
    except Exception as e:                      # never let the extra measurement take the bench line down
        vcf_ms = f"failed: {type(e).__name__}: {e}"
    b.close()
    # the same upload again: the batch above has returned its device slab to the library's cache, this one reuses it.  An upload
    # that has to hipMalloc its slab (the first batch of a process, or one created while the earlier batches are alive - the case
    # above, behind the two timed batches) pays 60-200 ms for the allocation; a pipeline pays that once per live batch
    tw = time.perf_counter()
    b2 = lib.Batch(cfg, tasks, device=device)
    warm_ms = (time.perf_counter() - tw) * 1e3
    b2.close()
    batched = dict(vcf_text_from_records_ms=vcf_ms, vcf_text_bytes=vcf_bytes, upload_ms=round((t1 - t0) * 1e3, 2),
                   pass_ms=round((t2 - t1) * 1e3, 2), d2h_ms=round((t3 - t2) * 1e3, 2),
                   materialise_ms=round((t4 - t3) * 1e3, 2), end_to_end_ms=round((t4 - t0) * 1e3, 2), svcalls=n,
                   materialise_all_candidates_ms=round(all_ms, 2), candidates=n_all,
                   upload_GBps=round(_input_bytes(tasks) / max(1e-9, t1 - t0) / 1e9, 2),
                   upload_slab_reused_ms=round(warm_ms, 2), end_to_end_slab_reused_ms=round((t4 - t1) * 1e3 + warm_ms, 2))
    # the reference's worker loop, one process: per contig task the two-call seam (Task.call_candidates + finalize_candidates, every
    # candidate an object) or the one-step drop-in (CallTask.execute_calls: upload, pass, objects of the kept calls).  `pipelined`: the loop
    # keeps two tasks in flight (Task.prepare: task k + 1 uploads and runs on the device while task k's records become objects)
    def per_task(shape, pipelined):
        ts = []
        for ti in tasks:
            task = parallel.CallTask(id=ti.task_id, sv_id=0, contig=ti.contig, start=0, end=ti.contig_len, config=cfg,
                                     tandem_repeats=None, device=device)
            task.lead_provider = pipeline._Extracted(ti)
            ts.append(task)
        ex = True if shape == "execute" else None
        ta = time.perf_counter()
        n_ = 0
        if pipelined and ts:
            ts[0].prepare(cfg, execute=ex)
        for k, task in enumerate(ts):
            if pipelined and k + 1 < len(ts):
                ts[k + 1].prepare(cfg, execute=ex)
            if shape == "execute":
                n_ += len(task.execute_calls(cfg))
            else:
                # Task.call_candidates + Task.finalize_candidates and CallTask.execute's own tail on their result (parallel.py:265-271):
                # `[s for s in svcalls if s.qc]`, `sorted(key=pos)` - reads `qc` of every candidate, `pos` of every kept call
                n_ += len(task.call_svs(cfg))
            task.close()
        return (time.perf_counter() - ta) * 1e3, n_
    exe_serial_ms, n3 = per_task("execute", False)
    exe_ms, _ = per_task("execute", True)
    api_serial_ms, n2 = per_task("api", False)
    api_ms, _ = per_task("api", True)
    # the INPUT half of the object boundary: Lead objects -> LeadProvider.record_lead / record_read -> TaskInput columns
    # (leadprov.py:400-418 on the reference's side).  Measured on the smallest contig task of the workload (building the Lead objects
    # themselves is the extraction's work and is not timed); the genome figure is that rate x all signatures
    ingest = None
    try:
        from sniffles_amd import leadprov
        ti_s = min(tasks, key=lambda t: t.n_leads)
        objs = list(leadprov.iter_leads(ti_s))
        rs_, re_, hp_ = ti_s.read_start.tolist(), ti_s.read_end.tolist(), ti_s.read_hp.tolist()
        lp = leadprov.LeadProvider(cfg, 0, ti_s.contig, contig_len=ti_s.contig_len)
        ti0 = time.perf_counter()
        for ld in objs:
            lp.record_lead(ld, 0)
        for a_, b_, c_ in zip(rs_, re_, hp_):
            lp.record_read(a_, b_, c_)
        ti1 = time.perf_counter()
        lp.to_task_input(ti_s.task_id, 0, None, ti_s.qc_nm_threshold)
        ti2 = time.perf_counter()
        n_all_leads = sum(t.n_leads for t in tasks)
        per_lead = (ti2 - ti0) / max(1, ti_s.n_leads)
        ingest = dict(contig=ti_s.contig, leads=int(ti_s.n_leads), reads=int(ti_s.n_reads), record_ms=round((ti1 - ti0) * 1e3, 2),
                      to_task_input_ms=round((ti2 - ti1) * 1e3, 2), us_per_lead=round(per_lead * 1e6, 3),
                      ingest_ms_genome_one_core=round(per_lead * n_all_leads * 1e3, 1),
                      ingest_ms_largest_task=round(per_lead * max(t.n_leads for t in tasks) * 1e3, 1),
                      note="record_lead turns the Lead into a row of typed columns when it is recorded (_snf_fast.LeadSink: where the "
                           "reference pays its binning, outside the call seam); to_task_input - inside Task.call_candidates - is a copy of the "
                           "finished columns + the string-rank remap; us_per_lead / the genome figures = record + to_task_input together; one "
                           "process per contig in the reference's layout: the largest task bounds the wall clock")
        del objs
    except Exception as e:  # noqa: BLE001
        ingest = f"failed: {type(e).__name__}: {e}"
    workers = None
    if specs is not None and not EMU and os.environ.get("SNF_BENCH_NO_WORKERS") != "1":
        workers = worker_processes(specs, cfg_kw or {}, device)
    return dict(batched=batched, ingest=ingest, worker_processes=workers,
                # (the better of the two loop shapes is the leg's figure: preparing the next task pays for the one-step shape, whose host
                # part is short; in the two-call shape one interpreter creates 94 k objects and the helper thread only adds hand-overs)
                per_task_api=dict(end_to_end_ms=round(min(api_ms, api_serial_ms), 2), two_tasks_in_flight_ms=round(api_ms, 2),
                                  one_task_at_a_time_ms=round(api_serial_ms, 2), tasks=len(tasks), svcalls=n2),
                per_task_execute=dict(end_to_end_ms=round(min(exe_ms, exe_serial_ms), 2), two_tasks_in_flight_ms=round(exe_ms, 2),
                                      one_task_at_a_time_ms=round(exe_serial_ms, 2), tasks=len(tasks), svcalls=n3),
                note="one genome, inputs in host numpy columns; upload = snf_batch_create + add_task + upload; "
                     "batched = all contig tasks in one device batch, the objects of what CallTask.execute returns (QC-passing calls, "
                     "sorted; "
                     "materialise_all_candidates_ms: every candidate instead); per_task_api = 24 x Task.call_candidates + "
                     "finalize_candidates + CallTask.execute's filter on `qc` and sort by `pos` "
                     "(the reference's two-call shape; candidates are stand-ins that become objects when touched - the kept calls do); "
                     "per_task_execute = 24 x "
                     "CallTask.execute_calls; both with two "
                     "tasks in flight (Task.prepare: the next task uploads and runs while this one's records become objects), "
                     "one_task_at_a_time_ms without; "
                     "d2h = results in the library's pinned block (read in place); materialise = SVCall Python objects (host); "
                     "vcf_text_from_records = the QC-passing records as VCF lines straight from the record table (no objects)")


def _input_bytes(tasks):
    n = 0
    for t in tasks:
        n += sum(int(a.nbytes) for a in t.leads.values()) + int(t.seq_pool.nbytes)
        n += int(t.read_start.nbytes) + int(t.read_end.nbytes) + int(t.read_hp.nbytes)
    return n


def other_configs(ctx) -> dict:
    """The other BASELINE.json configs in the default line (compact: a few steps each, verified against the oracle), so that
    they are measured wherever the headline is: configs[0] (chr20 only), [2] (60x HiFi), [3] (--mosaic) through the same
    passes as the headline, configs[4] (10-sample merge) through tools/bench_population.  Outside the headline's timed region."""
    import copy
    import threading

    import torch

    from sniffles_amd import abi, lib, synth
    from sniffles_amd.config import SnifflesConfig
    args, local_rank = ctx["args"], ctx["local_rank"]
    out = {}
    for k in (0, 2, 3):
        t_all = time.time()
        try:
            wl = WORKLOADS[k]
            a = copy.copy(args); a.config = k
            cfg = SnifflesConfig(**wl["cfg"])
            specs = task_specs(a, wl, 0, 0, 1)
            tasks = [synth.gen_task(**kw) for _, kw in specs]
            W, steps, warm = 2, 24, 4
            if k == 0:
                # (a small batch is replayed as a HIP graph: its first few launches cost milliseconds each - not the steady state)
                steps, warm = 48, 8
            hs = [lib.Batch(cfg, tasks, device=(0 if EMU else local_rank)) for _ in range(W)]
            for h in hs:
                h.set_output(abi.OUT_EXECUTE)

            def passes(n_each):
                def body(h):
                    set_dev(torch, local_rank)
                    for _ in range(n_each):
                        # one pass = snf_batch_pass (call_candidates + finalize; replayed as a graph for small batches), as the
                        # headline runs it
                        h.run_pass(); h.fetch_raw(1)
                ths = [threading.Thread(target=body, args=(h,)) for h in hs]
                for t in ths:
                    t.start()
                for t in ths:
                    t.join()
            passes(warm)
            dev_sync(torch)
            t0 = time.perf_counter()
            passes(steps // W)
            dev_sync(torch)
            dt = time.perf_counter() - t0
            t1 = time.perf_counter()
            hs[0].run_pass(); n_ret = hs[0].fetch_raw(1)
            lat = (time.perf_counter() - t1) * 1e3
            hs[0].set_output(abi.OUT_CANDIDATES); hs[0].call_candidates(); hs[0].finalize(); got = hs[0].fetch(1)
            hs[0].set_output(abi.OUT_EXECUTE); hs[0].call_candidates(); hs[0].finalize(); exe = hs[0].fetch(1)
            base, ver = cpu_baseline_and_verify(a, wl, got, [ci for ci, _ in specs], exe, cfg)
            n_sig = sum(t.n_leads for t in tasks)
            out[str(k)] = dict(workload=wl["name"], signatures=n_sig, steps=steps // W * W, batches_in_flight=W,
                               ms_per_step=round(dt / (steps // W * W) * 1e3, 3), ms_one_batch_in_flight=round(lat, 3),
                               signatures_per_s=round(n_sig * (steps // W * W) / dt), candidates=int(len(got.calls)),
                               records_returned=int(n_ret),
                               verified=ver["ok"], differences=ver["differences"], cpu_all_core_sig_s=round(base["all_core_sig_s"]),
                               cpu_cores=base["cores"])
            for h in hs:
                h.close()
            if not args.no_reference_baseline:       # ... and against the unmodified reference itself, on this box
                try:
                    rc = reference_check(a, wl, exe, tasks, specs)
                    if rc is not None:
                        out[str(k)].update(rc)
                        out[str(k)]["vs_reference_all_cores"] = round(out[str(k)]["signatures_per_s"] / max(1,
                        rc["reference_all_core_sig_s"]), 1)
                except Exception as e:  # noqa: BLE001
                    out[str(k)]["reference_error"] = f"{type(e).__name__}: {str(e)[:300]}"
            out[str(k)]["seconds"] = round(time.time() - t_all, 1)
        except Exception as e:  # noqa: BLE001 - the headline must not die with a side measurement
            out[str(k)] = dict(error=f"{type(e).__name__}: {e}")
    try:
        t_all = time.time()
        from tools import bench_population
        a = copy.copy(args); a.config = 4; a.steps = 2; a.warmup = 1
        # (the reference leg of the merge on a bounded sample: the whole workload takes two minutes of host time)
        a.reference_sample_contigs = 4
        r = bench_population.run(dict(ctx, args=a))
        out["4"] = dict(workload=r["config"]["workload"], metric=r["metric"], candidates=r["config"]["candidates"],
                        combined_calls=r["config"]["combined_calls"],
                        ms_per_step=round(r["ms_per_step"], 1), candidates_per_s=round(r["value"]), steps=r["steps"],
                        verified=r.get("verified"), verified_vs_reference=r.get("verified_vs_reference"),
                        kernel_ms=r["config"].get("rank0", {}).get("kernel_ms"), parity_unpinned=r["config"].get("parity_unpinned"),
                        cpu_baseline={k: (r.get("cpu_baseline") or {}).get(k) for k in ("kind", "value", "unit", "cores",
                                                                                        "hot_all_core_s", "vs_baseline",
                                                                                         "same_population", "whole_merge_estimate",
                                                                                         "reference_error", "sample")},
                        seconds=round(time.time() - t_all, 1))
    except Exception as e:  # noqa: BLE001
        out["4"] = dict(error=f"{type(e).__name__}: {e}")
    return out


def execute_mode_differences(got, exe, cfg) -> list:
    """SNF_OUT_EXECUTE against its definition (parallel.py:265-271) applied to the candidate-mode result on the host."""
    import numpy as np
    diffs = []
    keep = []
    for t in range(len(got.task_status)):
        lo, hi = int(got.task_call_off[t]), int(got.task_call_off[t + 1])
        idx = np.arange(lo, hi)
        if not cfg.no_qc:
            idx = idx[got.calls["qc"][lo:hi] != 0]
        if cfg.sort:
            idx = idx[np.argsort(got.calls["pos"][idx], kind="stable")]
        keep.append(idx)
    idx = np.concatenate(keep) if keep else np.zeros(0, np.int64)
    if len(idx) != len(exe.calls) or exe.task_call_off.tolist() != np.concatenate([[0], np.cumsum([len(k) for k in keep])]).tolist():
        return [f"execute mode: {len(exe.calls)} records, expected {len(idx)}"]
    for f in exe.calls.dtype.names:
        if f in ("alt_off", "rn_off"):
            continue
        a, e = exe.calls[f], got.calls[f][idx]
        if not np.array_equal(a, e, equal_nan=a.dtype.kind == "f"):
            diffs.append(f"execute mode: field {f} differs")

    def gather(pool, off, ln):
        ln = np.maximum(ln.astype(np.int64), 0)
        first = np.cumsum(ln) - ln
        return pool[np.repeat(off.astype(np.int64) - first, ln) + np.arange(int(ln.sum()), dtype=np.int64)]
    if not np.array_equal(gather(exe.alt_pool, exe.calls["alt_off"], exe.calls["alt_len"]),
                          gather(got.alt_pool, got.calls["alt_off"][idx], got.calls["alt_len"][idx])):
        diffs.append("execute mode: ALT bytes differ")
    if not np.array_equal(gather(exe.rnames, exe.calls["rn_off"], exe.calls["rn_len"]),
                          gather(got.rnames, got.calls["rn_off"][idx], got.calls["rn_len"][idx])):
        diffs.append("execute mode: read names differ")
    return diffs


def cpu_baseline_and_verify(args, wl, got, task_keys, exe=None, cfg=None):
    """The C oracle (scalar restatement of the reference, oracle/snf_oracle.c) over the WHOLE workload on this box's host
    cores: one process per contig task, at most one per core (the reference's schedule, `sniffles:495-530`).  A reported
    baseline, not the target.  With `got` (the HIP results of the bench batch) the same run is the checker of --verify."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cpu_pool
    from sniffles_amd import records
    specs = task_specs(args, wl, 0, 0, 1)
    r = cpu_pool.run_tasks(specs, wl["cfg"], weights=[kw["contig_len"] for _, kw in specs], want_results=got is not None)
    n = sum(m["n_leads"] for m in r["items"].values())
    base = dict(value=n / r["hot_all_core_s"], unit="signatures/s", cores=r["procs"], kind="port",
                cores_used=r["procs"], host_cores=r["cores"],
                all_core_sig_s=n / r["hot_all_core_s"], single_core_sig_s=n / r["hot_single_core_s"],
                reference_cpython_sig_s=REFERENCE_CPYTHON["sig_s"], reference_cpython_host=REFERENCE_CPYTHON["host"],
                sample=f"the whole workload ({len(specs)} contig tasks, {n} signatures), one oracle process per contig "
                       f"({r['procs']} processes, longest contig first), call_candidates + finalize_candidates only: slowest "
                       f"process {r['hot_all_core_s']:.3f} s, sum over tasks {r['hot_single_core_s']:.3f} s "
                       f"(wall incl. the dense coverage vector each task builds first: {r['wall_s']:.2f} s)")
    ver = None
    if got is not None:
        diffs, n_calls = [], 0
        contig_of = {key: kw["contig"] for key, kw in specs}
        for t, key in enumerate(task_keys):     # task t of the bench batch is the contig with this key
            exp = r["items"][key]["result"]
            n_calls += int(exp.calls.shape[0])
            for d in records.diff_results(got, t, exp, 0):
                diffs.append(f"task {t} ({contig_of[key]}): {d}")
        if exe is not None:
            diffs += execute_mode_differences(got, exe, cfg)
        ver = dict(ok=not diffs, tasks=len(specs), calls_compared=n_calls,
                   records_returned=(int(len(exe.calls)) if exe is not None else None),
                   what="every field of every candidate record, ALT bytes, supporting reads and coverage_average_total of the "
                        "bench batch vs the C oracle on the same inputs; the block the timed passes return (--output execute) vs "
                        "CallTask.execute's filter + sort applied to those candidates", differences=diffs[:5])
    return base, ver


def reference_differences(r, exe, tasks, task_keys):
    """The execute-mode block `exe` against what the unmodified reference's CallTask.execute keeps (the records a ref_pool run `r`
    returned), record by record and field by field: (differences, records compared)."""
    from sniffles_amd import records
    got = records.records(exe, tasks, "final")
    diffs, n_cmp = [], 0
    for t, key in enumerate(task_keys):
        exp = r["items"][key]["records"]
        g = got[t]
        n_cmp += len(exp)
        if isinstance(g, dict) or len(g) != len(exp):
            diffs.append(f"task {t}: {len(exp)} reference records, got {g if isinstance(g, dict) else len(g)}")
            continue
        for a, b in zip(g, exp):
            if a != b:
                diffs.append(f"task {t} {b['id']}: " + ", ".join(k for k in b if a.get(k) != b.get(k)))
                if len(diffs) > 5:
                    break
    return diffs, n_cmp


def reference_check(a, wl, exe, tasks, specs):
    """A side configuration against the LIVE reference on this box (oracle/_ref through oracle/ref_pool.py): the reference's rate on the
    host cores and the record-by-record comparison of the execute-mode block.  None where the staged reference is absent."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_pool
    if not ref_pool.available():
        return None
    extra = ["--mosaic"] if wl["cfg"].get("mosaic") else []
    r = ref_pool.run_tasks(specs, extra, weights=[kw["contig_len"] for _, kw in specs], want_results=True,
                           max_procs=int(os.environ.get("SNF_BENCH_REF_PROCS", "0")) or None)
    diffs, n_cmp = reference_differences(r, exe, tasks, [ci for ci, _ in specs])
    n = sum(m["n_leads"] for m in r["items"].values())
    return dict(verified_vs_reference=not diffs, records_compared=n_cmp, differences=diffs[:5],
                reference_all_core_sig_s=round(n / r["hot_all_core_s"]),
                reference_hot_all_core_s=round(r["hot_all_core_s"], 3), reference_procs=r["procs"],
                reference_leg_s=round(r["total_wall_s"], 1))


def reference_baseline(args, wl, exe, tasks, task_keys, out):
    """`cpu_baseline` with kind = "reference" (SURVEY.md 8d): the UNMODIFIED reference's `Task.call_candidates` +
    `finalize_candidates` (`parallel.py:104-201`) on the same 24 signature tables, one OS process per contig task, pool =
    min(tasks, host cores) - its own schedule (`sniffles:495-530`).  The reference is the byte-compiled staged build `oracle/_ref`
    (or the checkout in the build container).  With `exe` (the execute-mode block of the timed passes) the same run checks
    the GPU's records against the reference ITSELF: what `CallTask.execute` sends to the parent, record by record."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_pool
    if not ref_pool.available():
        return None
    from sniffles_amd import records
    specs = task_specs(args, wl, 0, 0, 1)
    extra = ["--mosaic"] if wl["cfg"].get("mosaic") else []
    r = ref_pool.run_tasks(specs, extra, weights=[kw["contig_len"] for _, kw in specs], want_results=exe is not None,
                           max_procs=int(os.environ.get("SNF_BENCH_REF_PROCS", "0")) or None)
    n = sum(m["n_leads"] for m in r["items"].values())
    ref_sig_s = n / r["hot_all_core_s"]
    base = dict(value=ref_sig_s, unit="signatures/s", cores=r["procs"], kind="reference", host_cores=r["cores"],
                all_core_sig_s=ref_sig_s, single_core_sig_s=n / r["hot_single_core_s"],
                hot_all_core_s=round(r["hot_all_core_s"], 3), hot_single_core_s=round(r["hot_single_core_s"], 2),
                reference=ref_pool.kind(),
                sample=f"the whole workload ({len(specs)} contig tasks, {n} signatures): the unmodified reference's Task.call_candidates + "
                       f"finalize_candidates, one process per contig ({r['procs']} processes on {r['cores']} usable cores, longest "
                       f"contig first; "
                       f"the reference cannot use more processes than contigs), every process starts at a barrier once its Lead tables / "
                       f"coverage vector are built (untimed: {r['build_single_core_s']:.0f} core-seconds of record_lead / record_hap_ref): "
                       f"slowest process {r['hot_all_core_s']:.2f} s, sum over tasks {r['hot_single_core_s']:.1f} s; whole leg "
                       f"{r['total_wall_s']:.0f} s")
    # speed-ups against the reference on THIS box (north_star: >= 20x wall clock at 1 MI355X vs all host cores)
    vs = dict(gpu_pass=round(out["value"] / ref_sig_s, 1))
    wc = out.get("wall_clock") or {}
    if wc.get("batched"):
        vs["wall_clock_batched"] = round(r["hot_all_core_s"] * 1e3 / wc["batched"]["end_to_end_ms"], 1)
        vs["wall_clock_per_task_api"] = round(r["hot_all_core_s"] * 1e3 / wc["per_task_api"]["end_to_end_ms"], 1)
        if wc.get("per_task_execute"):
            vs["wall_clock_per_task_execute"] = round(r["hot_all_core_s"] * 1e3 / wc["per_task_execute"]["end_to_end_ms"], 1)
        if isinstance(wc.get("worker_processes"), dict):
            vs["wall_clock_worker_processes"] = {k: round(r["hot_all_core_s"] * 1e3 / m["hot_all_ms"], 1)
                                                 for k, m in wc["worker_processes"].items() if isinstance(m, dict) and m.get("hot_all_ms")}
    vs["note"] = ("reference all-core seconds for one genome / this package's seconds for one genome: gpu_pass = the timed step (inputs "
                  "in HBM, "
                  "result block on the host); wall_clock_batched = numpy columns -> upload -> pass -> SVCall objects; per_task_api = 24 x "
                  "Task.call_candidates / finalize_candidates")
    base["vs_baseline"] = vs
    if exe is not None:
        diffs, n_cmp = reference_differences(r, exe, tasks, task_keys)
        base["verified_vs_reference"] = dict(ok=not diffs, records_compared=n_cmp, differences=diffs[:5],
                                             what="the execute-mode block of the timed passes (every field: POS, END, SVLEN, SVTYPE, "
                                                  "support, GT/GQ/DR/DV, "
                                                  "filters, fp64 statistics, INS consensus ALT, supporting read names) vs what the "
                                                  "unmodified reference's "
                                                  "CallTask.execute keeps (parallel.py:265-271) on the same signature tables, on this box")
        out["verified_vs_reference"] = not diffs
    return base
