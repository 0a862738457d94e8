"""`Task.call_candidates` / `Task.finalize_candidates` with the reference's signatures
(reference `src/sniffles/parallel.py:104-201`), running on the MI355X through the C-ABI library.

    task = Task(id=0, sv_id=0, contig="chr20", start=0, end=contig_len, config=config)
    task.lead_provider = lp                    # sniffles_amd.leadprov.LeadProvider (record_lead / record_read)
    task.tandem_repeats = [(start, end), ...]  # or None
    cands = task.call_candidates(keep_qc_fails, config)
    calls = task.finalize_candidates(cands, keep_qc_fails, config)

Differences forced by the device boundary: `svcall.postprocess` is a handle to the cluster in HBM, not a
`Cluster` with Python `Lead`s, and `finalize_candidates` must receive the list `call_candidates` returned.
Failure modes of the reference are preserved: a task whose first candidate is a BND raises
UnboundLocalError exactly like `postprocessing.coverage` does.
"""
from __future__ import annotations

from . import lib, sv
from .abi import TASK_ERR_UNBOUND_END


class Task:
    def __init__(self, id, sv_id, contig, start, end, config, assigned_process_id=None, lead_provider=None,
                 tandem_repeats=None, regions=None, device: int = 0, _lib=None):
        self.id, self.sv_id, self.contig, self.start, self.end = id, sv_id, contig, start, end
        self.config = config
        self.assigned_process_id = assigned_process_id
        self.lead_provider = lead_provider
        self.tandem_repeats = tandem_repeats
        self.regions = regions
        self.device = device
        self.coverage_average_total = None
        self._lib = _lib
        self._batch = None
        self._ti = None

    def _open(self, config):
        lp = self.lead_provider
        if lp.contig_len is None and lp.end is None:
            lp.end = self.end
        self._ti = lp.to_task_input(self.id, self.sv_id, self.tandem_repeats,
                                    getattr(config, "qc_nm_threshold", 0.02))
        if self._batch is not None:
            self._batch.close()
        self._batch = lib.Batch(config, [self._ti], device=self.device, _lib=self._lib)

    def call_candidates(self, keep_qc_fails, config, svcall_cls=sv.SVCall, bnd_cls=sv.SVCallBNDInfo) -> list:
        self._open(config)
        self._batch.call_candidates()
        res = self._batch.fetch(0)
        if int(res.task_status[0]) == TASK_ERR_UNBOUND_END:
            raise UnboundLocalError("local variable 'end' referenced before assignment")
        out = []
        for i in range(len(res.calls)):
            c = sv.fill_candidate(sv.new_call(svcall_cls), res, i, self._ti, bnd_cls)
            c.postprocess = sv.SVCallPostprocessingInfo(batch=self._batch, index=i)
            out.append(c)
        self.sv_id += len(out)
        self.coverage_average_total = float(res.coverage_average_total[0])
        return out

    def finalize_candidates(self, candidates, keep_qc_fails, config) -> list:
        if self._batch is None:
            raise RuntimeError("finalize_candidates needs the candidates of this task's call_candidates")
        self._batch.finalize()
        res = self._batch.fetch(1)
        if len(res.calls) != len(candidates):
            raise RuntimeError("candidate list does not match the batch (pass the list call_candidates returned)")
        passed = []
        for i, c in enumerate(candidates):
            sv.fill_final(c, res, i, self._ti)
            c.finalize()
            passed.append(c)
        self._batch.close()
        self._batch = None
        return passed


class CallTask(Task):
    """CallTask.execute's hot-path tail (parallel.py:265-271): candidates -> finalize -> QC filter -> sort by pos."""

    def call_svs(self, config):
        cands = self.call_candidates(config.no_qc, config)
        calls = self.finalize_candidates(cands, config.no_qc, config)
        if not config.no_qc:
            calls = [s for s in calls if s.qc]
        return sorted(calls, key=lambda s: s.pos)
