"""`Task.call_candidates` / `Task.finalize_candidates` with the reference's signatures
(reference `src/sniffles/parallel.py:104-201`), running on the MI355X through the C-ABI library.

    task = Task(id=0, sv_id=0, contig="chr20", start=0, end=contig_len, config=config)
    task.lead_provider = lp                    # sniffles_amd.leadprov.LeadProvider (record_lead / record_read)
    task.tandem_repeats = [(start, end), ...]  # or None
    cands = task.call_candidates(keep_qc_fails, config)
    calls = task.finalize_candidates(cands, keep_qc_fails, config)

Differences forced by the device boundary: `svcall.postprocess` is a handle to the cluster in HBM, not a
`Cluster` with Python `Lead`s, and `finalize_candidates` takes calls of this task's `call_candidates` (any selection, any order).
Failure modes of the reference are preserved: a task whose first candidate is a BND raises
UnboundLocalError exactly like `postprocessing.coverage` does.
"""
from __future__ import annotations

import os

from . import lib, sv
from .abi import TASK_ERR_UNBOUND_END

# SNF_EAGER_CALLS=1: `call_candidates` builds every candidate object at once (the behaviour up to round 5) instead of stand-ins
LAZY_CALLS = os.environ.get("SNF_EAGER_CALLS") != "1"


class Task:
    def __init__(self, id, sv_id, contig, start, end, config, assigned_process_id=None, lead_provider=None,
                 tandem_repeats=None, regions=None, device: int = 0):
        self.id, self.sv_id, self.contig, self.start, self.end = id, sv_id, contig, start, end
        self.config = config
        self.assigned_process_id = assigned_process_id
        self.lead_provider = lead_provider
        self.tandem_repeats = tandem_repeats
        self.regions = regions
        self.device = device
        self.coverage_average_total = None
        self._batch = None
        self._ti = None
        self._prep = None
        self._lazy = None
        self._served = False

    def prepare(self, config, execute=None):
        """Start this task's upload and its pass in the background; the next `call_candidates` (execute=None) / `call_records` /
        `execute_calls` (execute=False / True) on this task picks the running work up instead of starting it.  A worker loop that
        calls `tasks[k + 1].prepare(config)` before it turns task k's records into objects overlaps the two: the task's columns and
        the structs of the C-ABI are built here, the helper thread makes ONE library call (`snf_batch_open`: create, upload, enqueue -
        the interpreter lock is released for all of it); every task has its own batch handle and streams.  The reference's worker
        processes get the same overlap from being several (`sniffles:495-530`).  Optional: a task that was not prepared does the same
        work when it is called."""
        from .abi import OUT_CANDIDATES, OUT_EXECUTE
        if self._prep is not None or self._server() is not None:      # (through a GPU server the call itself hands the task over)
            return
        self._release()
        self._ti = self._task_input(config)
        from .soa import DeviceTaskInput
        if isinstance(self._ti, DeviceTaskInput):      # columns already in HBM: the call itself hands them over device-to-device
            return
        run = lib.Batch.RUN_CANDIDATES if execute is None else (lib.Batch.RUN_PASS | (OUT_EXECUTE if execute else OUT_CANDIDATES))
        self._prep = {"mode": execute, "pending": lib.Batch.open_in_background(config, [self._ti], device=self.device, run=run)}

    def _prepared(self, execute) -> bool:
        """True: `prepare` has already enqueued exactly this call's device work on this task's batch."""
        box, self._prep = self._prep, None
        if box is None:
            return False
        if box["mode"] != execute:            # prepared for another call than the one that comes: redone, not reused
            box["pending"].discard()
            return False
        self._batch = box["pending"].result()
        self._adopt()
        return True

    def _task_input(self, config):
        lp = self.lead_provider
        if lp.contig_len is None and lp.end is None:
            lp.end = self.end
        return lp.to_task_input(self.id, self.sv_id, self.tandem_repeats, getattr(config, "qc_nm_threshold", 0.02))

    def _adopt(self):
        lp = self.lead_provider
        lp.device_batch = self._batch   # SNFile.annotate_block_coverages(lead_provider) reads the coverage from HBM
        lp.task_input = self._ti        # cluster.resolve(svtype, lead_provider, ...) reads the clusters back (seam B3)

    def _open(self, config):
        self._ti = self._task_input(config)
        self._release()
        self._batch = lib.Batch(config, [self._ti], device=self.device)
        self._adopt()

    def close(self):
        """Release the task's device memory.  The batch outlives finalize_candidates because the SNF writer needs the
        read table afterwards (CallTask.execute, parallel.py:278-291); it goes with the task otherwise."""
        if getattr(self, "_prep", None) is not None:
            self._prep["pending"].discard()
            self._prep = None
        self._release()

    def _release(self):
        self._drop_lazy()
        if self._batch is not None:
            self._batch.close()
            self._batch = None
        if self.lead_provider is not None and getattr(self.lead_provider, "device_batch", None) is not None:
            self.lead_provider.device_batch = None

    # ---- through a GPU server (sniffles_amd.server: SNF_GPU_SERVER names its socket): many worker processes, one process on the device
    def _server(self):
        if os.environ.get("SNF_GPU_SERVER"):
            from . import server
            return server.client()
        return None

    def _served_result(self, srv, config):
        """This task through the server: its FINALIZED record table (views of the batch's result segment) and the reply to release."""
        import time
        self.close()
        t0 = time.perf_counter()
        self._ti = self._task_input(config)
        t1 = time.perf_counter()
        rep = srv.run_task(config, self._ti)
        # (where a served call's time goes, for tools/bench_workers.py: the task input, the hand-over + the server's batch)
        self.served_timing = dict(task_input_ms=(t1 - t0) * 1e3, server_ms=(time.perf_counter() - t1) * 1e3, batch_tasks=rep.msg.get("batch_tasks"),
                                  pack_ms=getattr(srv, "last_pack_ms", None))
        res = rep.result
        if int(res.task_status[0]) == TASK_ERR_UNBOUND_END:
            rep.release()
            raise UnboundLocalError("local variable 'end' referenced before assignment")
        self.coverage_average_total = float(res.coverage_average_total[0])
        return rep, res

    def _call_candidates_served(self, srv, config, svcall_cls, bnd_cls) -> list:
        rep, res = self._served_result(srv, config)
        self._finalized = False
        if sv.lazy_calls_supported(self._ti, svcall_cls):
            # stand-ins over the final records; `qc` as the candidate stage has it (True) until finalize_candidates
            src = sv.LazySource(res, self._ti, svcall_cls, bnd_cls, None, None, keep_all=bool(config.no_qc or getattr(config, "snf", None) is not None))
            src.alt_pool, src.final, src.on_detach = res.alt_pool, True, rep.release
            self._lazy = src
            out = src.make(all_qc=True)
        else:
            out = sv.materialize_candidates(res, self._ti, 0, len(res.calls), svcall_cls, bnd_cls)
            sv.apply_final(out, res, self._ti)
            rep.release()
        self._served = True
        self.sv_id += len(out)
        return out

    def call_candidates(self, keep_qc_fails, config, svcall_cls=sv.SVCall, bnd_cls=sv.SVCallBNDInfo) -> list:
        srv = self._server()
        if srv is not None:
            return self._call_candidates_served(srv, config, svcall_cls, bnd_cls)
        self._served = False
        if not self._prepared(None):
            self._open(config)
            self._batch.call_candidates()
        self._finalized = False
        if getattr(config, "dev_dump_clusters", False):      # cluster.py:316-324, one file per SV type
            from . import cluster
            from .soa import SVTYPES
            lp = self.lead_provider
            for svtype in SVTYPES:
                text = cluster.dump_clusters_bed(lp, config, svtype)
                if text:                                     # (the reference returns before the dump when a type has no seeds)
                    filename = f"{config.vcf}.clusters.{svtype}.{self.contig}.{getattr(lp, 'start', None) or 0}.{lp.end if getattr(lp, 'end', None) is not None else self.end}.bed"
                    print(f"Dumping clusters to {filename}")
                    with open(filename, "w") as h:
                        h.write(text)
        res = self._batch.fetch(0, copy=False)      # turned into objects (or their stand-ins) right here
        if int(res.task_status[0]) == TASK_ERR_UNBOUND_END:
            raise UnboundLocalError("local variable 'end' referenced before assignment")
        self._drop_lazy()
        if LAZY_CALLS and sv.lazy_calls_supported(self._ti, svcall_cls):
            # a real list of stand-ins that become `svcall_cls` objects when they are first touched (sv.LazySource): the reference's
            # consumers read `.qc` of every candidate and the rest only of the calls they keep (parallel.py:265-271)
            self._lazy = sv.LazySource(res, self._ti, svcall_cls, bnd_cls, sv.SVCallPostprocessingInfo, self._batch,
                                       keep_all=bool(config.no_qc or getattr(config, "snf", None) is not None))
            out = self._lazy.make()
        else:
            out = sv.materialize_candidates(res, self._ti, 0, len(res.calls), svcall_cls, bnd_cls, sv.SVCallPostprocessingInfo, self._batch)
        self.sv_id += len(out)
        self.coverage_average_total = float(res.coverage_average_total[0])
        return out

    def _drop_lazy(self):
        """Before the batch's result block is handed on (the task runs again, or is closed): the stand-ins of the previous candidate list
        that are still held become calls, their source lets go of the block (sv.LazySource.detach)."""
        src = getattr(self, "_lazy", None)
        self._lazy = None
        if src is not None:
            src.detach()

    def call_records(self, config, execute: bool = False):
        """call_candidates + finalize_candidates without the `SVCall` objects: the finalized record table of the task
        (`lib.Result`) and its input, for consumers that format or count straight from the records (vcf.VCF.write_records).
        `execute`: only what `CallTask.execute` keeps (QC-passing calls unless `config.no_qc`, sorted by position when
        `config.sort`; parallel.py:265-271), filtered and ordered on the device.
        The arrays of the result are VIEWS of the batch's pinned block: they die with `close()` or the next call on this task
        (the buffer is handed to the next batch) - copy what has to outlive it."""
        from .abi import OUT_CANDIDATES, OUT_EXECUTE
        if not self._prepared(bool(execute)):
            self._open(config)
            self._batch.set_output(OUT_EXECUTE if execute else OUT_CANDIDATES)
            self._batch.run_pass()            # call_candidates + finalize (both only enqueue; the fetch is the one wait)
        res = self._batch.fetch(1, copy=False)
        if int(res.task_status[0]) == TASK_ERR_UNBOUND_END:
            raise UnboundLocalError("local variable 'end' referenced before assignment")
        self._finalized = True
        self.sv_id += int(self._batch.n_candidates())
        self.coverage_average_total = float(res.coverage_average_total[0])
        return res, self._ti

    def finalize_candidates(self, candidates, keep_qc_fails, config) -> list:
        """`candidates`: calls of this task's `call_candidates` - the list itself, or any selection of it in any order (the reference
        iterates whatever it is given, parallel.py:129-147; `GenotypeTask`-style callers filter in between): every call carries its
        place in the batch (`postprocess.index`, a stand-in its own index), the records are mapped through that."""
        import numpy as np
        if getattr(self, "_served", False):
            # through a GPU server the records were final from the start: the stand-ins take their final `qc`, calls that were touched
            # before carry their final fields already; `finalize()` (postprocess = None) holds for both
            if getattr(self, "_finalized", False):
                raise RuntimeError("finalize_candidates needs the candidates of this task's call_candidates")
            candidates = list(candidates)
            if self._lazy is not None:
                self._lazy.refresh_qc()
            self._finalized = True
            return candidates
        if self._batch is None or getattr(self, "_finalized", False):
            raise RuntimeError("finalize_candidates needs the candidates of this task's call_candidates")
        self._batch.finalize()
        res = self._batch.fetch(1, copy=False)
        candidates = list(candidates)
        n_rec = len(res.calls)
        src = getattr(self, "_lazy", None)
        if src is not None:
            src.set_final(res)          # the stand-ins take the final `qc`; whoever is touched from now on is built in its final state
        real, idx = [], []
        # (the stand-ins of this task's source need nothing: whoever touches them finds the final records)
        others = sv._load_fast().stub_others(candidates, src) if src is not None else candidates
        for c in others:
            if sv.is_stand_in(c):
                raise RuntimeError("a candidate of another task's (or an earlier) call_candidates was passed")
            pp = getattr(c, "postprocess", None)
            i = getattr(pp, "index", None)
            if pp is None or getattr(pp, "batch", None) is not self._batch or i is None or not 0 <= int(i) < n_rec:
                raise RuntimeError("finalize_candidates takes calls that this task's call_candidates returned (each carries its place in "
                                   "the batch in `postprocess`)")
            real.append(c); idx.append(int(i))
        if real:
            idx = np.asarray(idx, np.int64)
            if not (np.array_equal(res.calls["pos"][idx], np.fromiter((int(c.pos) for c in real), np.int64, len(real)))
                    and np.array_equal(res.calls["svlen"][idx], np.fromiter((int(c.svlen) for c in real), np.int64, len(real)))):
                raise RuntimeError("a candidate is not the call the batch holds at its place (POS / SVLEN were changed)")
            sv.apply_final(real, res, self._ti, finalize=True, idx=idx)      # (with `c.finalize()` for every call, parallel.py:199-200)
        self._finalized = True
        return candidates


class CallTask(Task):
    """CallTask.execute's hot-path tail (parallel.py:265-271): candidates -> finalize -> QC filter -> sort by pos."""

    def call_svs(self, config):
        cands = self.call_candidates(config.no_qc, config)
        calls = self.finalize_candidates(cands, config.no_qc, config)
        if not config.no_qc:
            calls = [s for s in calls if s.qc]
        return sorted(calls, key=lambda s: s.pos)

    def execute_calls(self, config, svcall_cls=sv.SVCall, bnd_cls=sv.SVCallBNDInfo) -> list:
        """What `CallTask.execute` hands to its `CallResult` (parallel.py:264-271) in one step: `call_candidates`,
        `finalize_candidates`, `[s for s in svcalls if s.qc]` (unless `config.no_qc`) and `sorted(key=pos)` all happen on the
        device (SNF_OUT_EXECUTE), and only the calls the worker would send to its parent become `SVCall` objects - for a 30x
        genome 26.8 k objects instead of the 94 k candidates the two-call form has to materialise first.  Same objects, same
        order as `call_svs()` (tests/test_dropin_api.py); `self.sv_id` advances by the number of candidates like there."""
        srv = self._server()
        if srv is not None:       # through a GPU server: the kept calls are picked and ordered here, from the finalized record table
            import numpy as np
            rep, res = self._served_result(srv, config)
            idx = np.arange(len(res.calls)) if config.no_qc else np.flatnonzero(res.calls["qc"] != 0)
            if getattr(config, "sort", True):
                idx = idx[np.argsort(res.calls["pos"][idx], kind="stable")]
            calls = sv.materialize_candidates(res, self._ti, 0, len(res.calls), svcall_cls, bnd_cls, idx=idx)
            sv.apply_final(calls, res, self._ti, finalize=True, idx=idx)
            self.sv_id += len(res.calls)
            rep.release()
            return calls
        res, ti = self.call_records(config, execute=True)
        calls = sv.materialize_candidates(res, ti, 0, len(res.calls), svcall_cls, bnd_cls)
        sv.apply_final(calls, res, ti, finalize=True)
        return calls

    def write_snf_part(self, svcandidates, snf_filename: str):
        """The SNF tail of CallTask.execute (parallel.py:278-295): the task's candidates (after finalize_candidates) go
        into 100-kb blocks with their downsampled coverage and are written as a part file; the returned record is what
        `SNFile.add_result` / `write_results` consume."""
        from . import snf
        with open(snf_filename, "wb") as handle:
            out = snf.SNFile(self.config, handle)
            for cand in svcandidates:
                out.store(cand)
            out.annotate_block_coverages(self.lead_provider)
            out.write_and_index()
        return snf.SNFPart(task_id=self.id, contig=self.contig, snf_filename=snf_filename, snf_index=out.get_index(),
                           snf_total_length=out.get_total_length(), snf_candidate_count=len(svcandidates),
                           coverage_average_total=self.coverage_average_total)


class GenotypeTask(Task):
    """`GenotypeTask.execute` of the reference (parallel.py:299-372): force-calling of known SVs (`--genotype-vcf`).
    The sample's own candidates come from the GPU hot path; every target SV takes the closest candidate of its type
    within the merge gates as `genotype_match_sv` (host bookkeeping over a 5-kb bin table, as in the reference), the
    targets are annotated with the sample's coverage on the device (`postprocessing.coverage`) and receive the
    reference-allele genotype that the VCF rewriter uses when nothing matched."""

    def __init__(self, *args, genotype_svs=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.genotype_svs = genotype_svs if genotype_svs is not None else []

    def execute(self):
        import math

        from . import postprocessing
        config = self.config
        svcandidates = self.call_candidates(False, config)
        self.finalize_candidates(svcandidates, True, config)
        binsize = 5000
        binedge = binsize // 10
        table = {svtype: {} for svtype in sv.TYPES}
        for target in self.genotype_svs:
            target.genotype_match_sv = None
            target.genotype_match_dist = math.inf
            if target.svtype not in table:
                continue
            b0 = int(target.pos / binsize)
            bins = [b0 * binsize]
            if target.pos % binsize < binedge:
                bins.append((b0 - 1) * binsize)
            if target.pos % binsize > binsize - binedge:
                bins.append((b0 + 1) * binsize)
            for b in bins:
                table[target.svtype].setdefault(b, []).append(target)
        for cand in svcandidates:
            if cand.svtype.startswith("SINGLE"):
                continue
            targets = table[cand.svtype].get(int(cand.pos / binsize) * binsize)
            if not targets:
                continue
            for target in targets:
                if cand.svtype == "BND":
                    dist = abs(target.pos - cand.pos)
                    ok = dist <= config.cluster_merge_bnd and cand.bnd_info.mate_contig == target.bnd_info.mate_contig
                else:
                    dist = abs(target.pos - cand.pos) + abs(abs(target.svlen) - abs(cand.svlen))
                    minlen = float(min(abs(target.svlen), abs(cand.svlen)))
                    ok = minlen > 0 and dist <= config.combine_match * math.sqrt(minlen) and dist <= config.combine_match_max
                if ok and dist < target.genotype_match_dist:
                    target.genotype_match_sv = cand
                    target.genotype_match_dist = dist
        postprocessing.coverage(self.genotype_svs, self.lead_provider)
        for target in self.genotype_svs:
            samples = [c for c in (target.coverage_start, target.coverage_center, target.coverage_end) if c is not None]
            if len(samples) == 0:
                return None
            depth = round(sum(samples) / len(samples))
            target.genotypes = {0: (0, 0, 0, depth, 0, (None, None)) if depth > 0 else config.genotype_none}
        return self.genotype_svs


class CombineTask(Task):
    """`CombineTask.execute` of the reference (parallel.py:444-572): the block / bin / flush-window driver of the
    multi-sample merge.  The candidates are read block by block from objects with the SNF reader interface
    (`read_blocks(contig, block_index) -> [ {svtype: [SVCall], "_COVERAGE": {bin: depth}} ] | None`, attribute `reqc`);
    every flush window is resolved on the GPU (`cluster.resolve_block_groups`: distance gates, running means and the
    edit distance of `SVGroup.align_call`), the keep / call decision and the combined calls are host bookkeeping
    (`SVGroup.call`).  Groups kept at the end of a window seed the next one, also across blocks, exactly as in the
    reference, so the windows of one SV type form a chain; different SV types are independent."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        bs = self.config.snf_block_size
        if self.regions:
            idx = set()
            for r in self.regions:
                start = r.start // bs * bs
                idx |= set(range(start, r.end + bs, bs))
            self.block_indices = sorted(idx)
        else:
            self.block_indices = list(range(self.start, self.end + bs, bs))

    TARGET_WORK_PER_TASK = 10000      # blocks x samples one task should handle (parallel.py:378)

    def clone(self, first_block: int, block_count: int, new_id: int = None) -> "CombineTask":
        """This task restricted to `block_count` consecutive blocks from `first_block` on (parallel.py:411-420)."""
        import copy
        obj = copy.copy(self)
        if new_id is not None:
            obj.id = new_id
        obj.block_indices = self.block_indices[first_block:first_block + block_count]
        obj.start = obj.block_indices[0]
        obj.end = obj.block_indices[-1] + obj.config.snf_block_size
        return obj

    def scatter(self) -> list:
        """`CombineTask.scatter` of the reference (parallel.py:422-442), cut for cut: when the task holds more than
        TARGET_WORK_PER_TASK blocks x samples (and more than one worker is configured) it falls apart into tasks of
        `total_blocks // TARGET_WORK_PER_TASK` consecutive blocks with ids id+1, id+2, ...  Every part is a task of its own -
        groups that would have been kept across a cut are flushed at the end of the part, exactly as the reference does -
        which is what makes the parts the unit of a multi-GPU merge of one contig (sniffles_amd.dist.shard_lpt / TaskQueue
        over the parts; the calls of a contig are the concatenation over its parts)."""
        total_blocks = len(self.block_indices) * len(self.config.sample_ids_vcf)
        if total_blocks <= self.TARGET_WORK_PER_TASK or getattr(self.config, "threads", 1) <= 1:
            return [self]
        blocks_per_task = total_blocks // self.TARGET_WORK_PER_TASK
        return [self.clone(fb, blocks_per_task, new_id=self.id + i + 1)
                for i, fb in enumerate(range(0, len(self.block_indices), blocks_per_task))]

    def execute(self, samples_snf: dict) -> list:
        """samples_snf: {internal_id: SNF reader}.  Returns the combined calls in the reference's emission order
        (`execute_many` with this task alone; see `sniffles_amd.candstore`)."""
        return CombineTask.execute_many([self], samples_snf)[0]

    @staticmethod
    def execute_many(tasks: list, samples_snf: dict, text_writer=None) -> list:
        """`execute` of several tasks (the contigs of a merge, or the parts of `scatter`) with ONE group-assignment launch for
        all of them: the flush windows of every task are walked first, all chains go to the GPU together - a contig alone
        leaves most of the device idle and its launch lasts as long as its slowest window - then every task is replayed.
        Returns the calls per task, each list exactly what `task.execute(samples_snf)` returns."""
        if not tasks:
            return []
        import os
        if os.environ.get("SNF_COMBINE_OBJECTS", "0") != "1" and sv._load_fast() is not None:
            from . import candstore
            return candstore.execute_many(tasks, samples_snf, text_writer)      # the columnar store: no Python per candidate or group
        if text_writer is not None:
            raise RuntimeError("merged VCF text without objects needs the columnar store (sniffles_amd._snf_fast)")
        return CombineTask._execute_many_objects(tasks, samples_snf)

    @staticmethod
    def _execute_many_objects(tasks: list, samples_snf: dict) -> list:
        """The object-by-object form of `execute_many` (the reference's own shape: `SVGroup` objects, `SVGroup.call`): the twin the
        tests compare the columnar store with, and the path without the C extension."""
        collected = [t._collect(samples_snf) for t in tasks]
        assigns = tasks[0]._resolve(tasks, [c for c, _ in collected])
        ids = set(samples_snf.keys())
        return [t._replay(c, e, a, ids) for t, (c, e), a in zip(tasks, collected, assigns)]

    def _resolve(self, tasks: list, chains_per_task: list) -> list:
        """Phase 2: all chains of all `tasks` in one `snf_combine_resolve_batch` call; per task {svtype: group numbers}."""
        from . import cluster
        flat, owner = [], []
        for k, chains in enumerate(chains_per_task):
            for t in sv.TYPES:
                if chains[t]["win_bin"]:
                    flat.append((t, chains[t]["cands"], chains[t]["win_off"], chains[t]["win_bin"], chains[t]["win_thr"]))
                    owner.append((k, t))
        outs = cluster.resolve_chains_batch(flat, self.config, device=self.device) if flat else []
        assigns = [dict() for _ in chains_per_task]
        for (k, t), o in zip(owner, outs):
            assigns[k][t] = o
        return assigns

    def _collect(self, samples_snf: dict):
        """Phase 1: the block / bin / flush-window walk of the reference (parallel.py:487-534)."""
        config = self.config
        bin_min_size = config.combine_min_size
        bin_max_candidates = max(25, int(len(config.snf_input_info) * 0.5))
        overlap_abs = config.combine_overlap_abs
        support_threshold = config.combine_support_threshold
        sample_internal_ids = set(samples_snf.keys())
        # ---- phase 1: windows
        chains = {svtype: dict(cands=[], win_off=[0], win_bin=[], win_thr=[]) for svtype in sv.TYPES}
        events = []   # (svtype, window index in its chain, curr_bin, size, samples_blocks of the block) in emission order
        regenotype = []
        for block_index in self.block_indices:
            samples_blocks = {sid: snf.read_blocks(self.contig, block_index) for sid, snf in samples_snf.items()}
            for svtype in sv.TYPES:
                bins = {}
                for sid, snf in samples_snf.items():
                    blocks = samples_blocks[sid]
                    reqc = getattr(snf, "reqc", False)
                    if blocks is None:
                        continue
                    for block in blocks:
                        for cand in block[svtype]:
                            if cand.support < support_threshold:
                                continue
                            if reqc:      # SNF written before 2.5.3: genotype again (parallel.py:507-508); one launch, below
                                regenotype.append(cand)
                            cand.sample_internal_id = sid
                            bins.setdefault(int(cand.pos / bin_min_size) * bin_min_size, []).append(cand)
                if len(bins) == 0:
                    continue
                ch = chains[svtype]
                size = 0
                svcands = []
                sorted_bins = sorted(bins)
                last_bin = sorted_bins[-1]
                for curr_bin in sorted_bins:
                    svcands.extend(bins[curr_bin])
                    size += bin_min_size
                    if (not getattr(config, "combine_exhaustive", False) and len(svcands) >= bin_max_candidates) or curr_bin == last_bin:
                        if len(svcands) == 0:
                            size = 0
                            continue
                        events.append((svtype, len(ch["win_bin"]), curr_bin, size, samples_blocks))
                        ch["cands"].extend(svcands)
                        ch["win_off"].append(len(ch["cands"]))
                        ch["win_bin"].append(curr_bin)
                        ch["win_thr"].append(float(max(size * 0.5, overlap_abs)))
                        size = 0
                        svcands = []
        if regenotype:
            from . import postprocessing
            postprocessing.genotype_svs(regenotype, config, device=self.device)
        return chains, events

    def _replay(self, chains, events, assign, sample_internal_ids):
        """Phase 3: SVGroup bookkeeping in the reference's emission order (parallel.py:536-572)."""
        config = self.config
        overlap_abs = config.combine_overlap_abs
        # ---- phase 3: replay
        active = {svtype: [] for svtype in sv.TYPES}     # kept groups, list order
        by_id = {svtype: {} for svtype in sv.TYPES}      # group number -> SVGroup (active ones only)
        next_id = {svtype: 0 for svtype in sv.TYPES}
        calls = []
        for svtype, w, curr_bin, size, samples_blocks in events:
            ch, out = chains[svtype], assign[svtype]
            lo, hi = ch["win_off"][w], ch["win_off"][w + 1]
            cands = ch["cands"][lo:hi]
            groups, gmap = list(active[svtype]), by_id[svtype]
            for i in sorted(range(len(cands)), key=lambda i: cands[i].support, reverse=True):   # stable, like the reference
                gid = int(out[lo + i])
                if gid in gmap:
                    gmap[gid].add_candidate(cands[i])
                else:
                    if gid != next_id[svtype]:   # new groups are numbered in creation order over the whole chain
                        raise RuntimeError("internal: the kernel's keep / flush decisions differ from the host replay")
                    next_id[svtype] += 1
                    gmap[gid] = sv.SVGroup.from_candidate(cands[i])
                    groups.append(gmap[gid])
            groups_call, keep = [], []
            for group in groups:
                coverage_bin = int(group.pos_mean / config.coverage_binsize_combine) * config.coverage_binsize_combine
                for other in sample_internal_ids - group.included_samples:
                    blk = samples_blocks[other]
                    coverage = blk[0]["_COVERAGE"].get(coverage_bin, 0) if blk is not None else 0
                    if other in group.coverages_nonincluded:
                        group.coverages_nonincluded[other] = max(coverage, group.coverages_nonincluded[other])
                    else:
                        group.coverages_nonincluded[other] = coverage
                if abs(group.pos_mean - curr_bin) < max(size * 0.5, overlap_abs):
                    keep.append(group)
                else:
                    groups_call.append(group)
            kept_ids = {id(g) for g in keep}
            by_id[svtype] = {gid: g for gid, g in gmap.items() if id(g) in kept_ids}
            active[svtype] = keep
            calls.extend(sv.call_groups(groups_call, config, self))
        for svtype in sv.TYPES:
            calls.extend(sv.call_groups(active[svtype], config, self))
        return calls
