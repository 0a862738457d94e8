"""TEST INFRASTRUCTURE (build container only): the drop-in seams exercised IN SITU - the patches INTEGRATION.md section 4
describes are applied to the UNMODIFIED, imported reference (`/root/reference/src`, nothing is edited on disk), and the
reference's own `CallTask.execute` (`src/sniffles/parallel.py:255-297`) then runs around this package's library:

  patch 1 (`leadprov.py`)  `LeadProvider.record_lead` additionally appends to `sniffles_amd.leadprov.LeadProvider`; the reads the
                           reference's `iter_region` accepts (`leadprov.py:506-510`, `:567-571`) reach `record_read` through a
                           tap around `bam.fetch` - a read counts exactly when the reference's own `read_count` moved
  patch 2 (`parallel.py`)  `Task.call_candidates` / `Task.finalize_candidates` become the bodies of
                           `sniffles_amd.parallel.Task` with `svcall_cls=sniffles.sv.SVCall, bnd_cls=sniffles.sv.SVCallBNDInfo`

Everything else is the reference: `Task.build_leadtab`, `iter_region` (over `oracle/pysam_stub`), `CallTask.execute` (QC filter,
sort), `CallResult`, the pickling of the result through a `multiprocessing` pipe (`parallel.py:757`), the VCF writer.
Never imported by the product (`sniffles_amd/`); `tests/test_insitu_seam.py` is the only user.
"""
from __future__ import annotations

import contextlib
import io
import math
import multiprocessing
import struct
import threading

import ref_harness


class _ReadTap:
    """Around the alignment file `build_leadtab` hands to `iter_region`: yields the same records; a record the reference's
    filters accepted (its `read_count` moved while the record was being processed) is reported to `record_read` with the
    HP tag the reference itself reads (`leadprov.py:506`)."""

    def __init__(self, bam, provider):
        self._bam, self._lp = bam, provider

    def __getattr__(self, name):
        return getattr(self._bam, name)

    def fetch(self, *a, **k):
        lp = self._lp
        for read in self._bam.fetch(*a, **k):
            before = lp.read_count
            yield read
            if lp.read_count != before:
                hp = read.get_tag("HP") if read.has_tag("HP") else 0
                lp._amd.record_read(read.reference_start, read.reference_end, hp)
                lp._seen["reads"] += 1


@contextlib.contextmanager
def patched(emu_lib=None, device: int = 0):
    """The reference modules with patches 1 + 2 applied (restored on exit).  `emu_lib`: a test tier's library binding (None:
    the real HIP library)."""
    ref = ref_harness.load_reference()
    from sniffles_amd import leadprov as amd_leadprov, parallel as amd_parallel
    LP, Task = ref.leadprov.LeadProvider, ref.parallel.Task
    orig = dict(build=LP.build_leadtab, record=LP.record_lead, cc=Task.call_candidates, fc=Task.finalize_candidates)
    seen = ref.insitu_seen = dict(leads=0, reads=0, call_candidates=0, finalize_candidates=0, calls=0)   # proof the seams were taken

    def build_leadtab(self, regions, bam):
        self._amd = amd_leadprov.LeadProvider(self.config, self.read_id, self.contig,
                                              contig_len=int(bam.get_reference_length(self.contig)))
        self._seen = seen
        return orig["build"](self, regions, _ReadTap(bam, self))

    def record_lead(self, ld, pos_leadtab):
        orig["record"](self, ld, pos_leadtab)
        self._amd.record_lead(ld, pos_leadtab)
        seen["leads"] += 1

    def call_candidates(self, keep_qc_fails, config):
        t = amd_parallel.Task(id=self.id, sv_id=self.sv_id, contig=self.contig, start=self.start, end=self.end, config=config,
                              lead_provider=self.lead_provider._amd, tandem_repeats=self.tandem_repeats, device=device)
        self._amd_task = t
        out = t.call_candidates(keep_qc_fails, config, svcall_cls=ref.sv.SVCall, bnd_cls=ref.sv.SVCallBNDInfo)
        self.sv_id, self.coverage_average_total = t.sv_id, t.coverage_average_total
        seen["call_candidates"] += 1; seen["calls"] += len(out)
        return out

    def finalize_candidates(self, candidates, keep_qc_fails, config):
        seen["finalize_candidates"] += 1
        try:
            return self._amd_task.finalize_candidates(candidates, keep_qc_fails, config)
        finally:
            self._amd_task.close()

    LP.build_leadtab, LP.record_lead = build_leadtab, record_lead
    Task.call_candidates, Task.finalize_candidates = call_candidates, finalize_candidates
    try:
        yield ref
    finally:
        LP.build_leadtab, LP.record_lead = orig["build"], orig["record"]
        Task.call_candidates, Task.finalize_candidates = orig["cc"], orig["fc"]


def through_pipe(obj):
    """`pipe_worker.send(result)` / `recv()` of the reference's worker protocol (`parallel.py:757`): the object as the parent
    process receives it."""
    a, b = multiprocessing.Pipe()
    box = []
    th = threading.Thread(target=lambda: box.append(b.recv()))
    th.start()
    a.send(obj)
    th.join()
    a.close(); b.close()
    return box[0]


def sample_config(ref, recs, extra_args=(), fixed=None):
    """The main program's configuration of a `call_sample` run (sniffles:286-360), as ref_harness.run_reference_call_sample."""
    cfg = ref.config.SnifflesConfig(*(["--input", "x.bam", "--vcf", "out.vcf"] + list(extra_args)))
    cfg.mode = "call_sample"
    cfg.input_is_cram, cfg.input_mode = False, "rb"
    cfg.sample_ids_vcf = [(0, "SAMPLE")]
    for k, v in (fixed or {}).items():
        setattr(cfg, k, v)
    flags = [struct.unpack_from("<H", recs.blob, int(o) + 18)[0] for o in recs.rec_off[:-1]]
    total_mapped = sum(1 for f, r in zip(flags, recs.ref_id.tolist()) if r >= 0 and not f & 0x4)
    cfg.task_read_id_offset_mult = 10 ** 9 if total_mapped == 0 else 10 ** math.ceil(math.log(total_mapped) + 1)
    cfg.contig_lengths = [(c, int(n)) for c, n in zip(recs.ref_names, recs.ref_lens) if ref.util.should_process_contig(c, int(n), cfg)]
    return cfg


def run_call_sample(recs, extra_args=(), fixed=None, via=through_pipe, in_child=False):
    """The reference's `call_sample` flow over an in-memory BAM, one `CallTask.execute` per contig, every `CallResult` taken
    through `via` before the reference's VCF writer sees it.  With `in_child` every task executes in a forked worker process
    that sends its result back through the pipe - the reference's process layout (`parallel.py:730-760`).
    Returns dict(vcf=text, results=[CallResult as received], read_count=...)."""
    import pysam_stub
    ref = ref_harness.load_reference()
    from sniffles import vcf as ref_vcf
    cfg = sample_config(ref, recs, extra_args, fixed)
    buf = io.StringIO()
    vcf_out = ref_vcf.VCF(cfg, buf)
    vcf_out.write_header(cfg.contig_lengths)
    orig = ref.parallel.pysam.AlignmentFile
    ref.parallel.pysam.AlignmentFile = lambda *a, **k: pysam_stub.AlignmentFile(recs)
    results, read_count = [], 0
    try:
        for task_id, (contig, length) in enumerate(cfg.contig_lengths):
            task = ref.parallel.CallTask(id=task_id, contig=contig, start=0, end=length - 1, assigned_process_id=None,
                                         tandem_repeats=(getattr(recs, "tandem_repeats", None) or {}).get(contig),
                                         genotype_svs=None, sv_id=0, config=cfg, regions=None)   # sniffles:351
            if in_child:
                ctx = multiprocessing.get_context("fork")
                parent, child = ctx.Pipe()

                def work(task=task, child=child):
                    child.send(task.execute())
                    child.close()
                p = ctx.Process(target=work)
                p.start()
                result = parent.recv()
                p.join()
                assert p.exitcode == 0
            else:
                result = via(task.execute())
            read_count += result.processed_read_count
            results.append(result)
            result.emit(vcf_out=vcf_out, snf_out=None)
    finally:
        ref.parallel.pysam.AlignmentFile = orig
    return dict(vcf=buf.getvalue(), results=results, read_count=int(read_count))
