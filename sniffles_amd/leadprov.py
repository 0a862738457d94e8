"""Input side of the drop-in: `Lead` and `LeadProvider` with the reference's names
(reference `src/sniffles/leadprov.py:34-56, 358-472`).

The reference keeps `leadtab[svtype][bin] -> list[Lead]`, per-bin hap counters and a dense uint16 coverage
vector.  Here `record_lead` / `record_read` only append to arrival-ordered columns; binning, the 10-leads
per bin sequence cap, hap counters and coverage queries all happen on the GPU.
`record_read(ref_start, ref_end, hp)` replaces the pair `coverage[s:e] += 1` + `record_hap_ref(...)`
that `iter_region` issues per alignment (leadprov.py:510, 567-571).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

from .soa import (TaskInput, SVT, SRC, SVLEN_NONE, SEQ_NONE, PS_NONE, empty_leads, intern_sorted)
from .sv import SVCallBNDInfo


@dataclass
class Lead:
    read_id: int = None
    read_qname: str = None
    contig: str = None
    ref_start: int = None
    ref_end: int = None
    qry_start: int = None
    qry_end: int = None
    strand: str = None
    mapq: int = None
    nm: float = None
    source: str = None
    svtype: str = None
    svlen: Optional[int] = None
    seq: Optional[str] = None
    svtypes_starts_lens: list = None
    bnd_info: Optional[SVCallBNDInfo] = None
    hap: str = "0"
    phase_set: str = None
    is_sa: bool = False
    read_len: int = 0


class LeadProvider:
    def __init__(self, config, read_id_offset, contig: str, contig_len: int = None):
        self.config = config
        self.contig = contig
        self.contig_len = contig_len
        self.start = None
        self.end = None
        self.read_id = read_id_offset
        self.read_count = 0
        self._leads = []
        self._reads = []
        self._nmask = None

    def record_lead(self, ld: Lead, pos_leadtab: int = None) -> None:
        """Same call as the reference; `pos_leadtab` (the 100-bp bin) is recomputed on the GPU."""
        self._leads.append(ld)

    def record_read(self, ref_start: int, ref_end: int, hp: int = 0) -> None:
        self._reads.append((int(ref_start), int(ref_end), int(hp)))
        self.read_count += 1

    def _mask_N_coverage(self, regions=None, fasta=None) -> None:
        """`LeadProvider._mask_N_coverage` (leadprov.py:420-443): with `config.reference`, coverage reads as 0 wherever the
        reference base is 'N'.  `fasta`: anything with pysam's `fetch(contig[, start, end])` (the host's business; the
        device takes the mask as intervals).  Failures to open / fetch only skip the masking, as in the reference."""
        import logging
        from .soa import paint_nmask
        if not getattr(self.config, "reference", None) or fasta is None:
            return
        try:
            regs = None if regions is None else [(r.start, r.end) if hasattr(r, "start") else (r[-2], r[-1]) for r in regions]
            clen = self.contig_len if self.contig_len is not None else self.end
            # dense-paint semantics (list order, later regions overwrite), irregular input fails as in the reference -> unmasked
            self._nmask = paint_nmask(fasta.fetch, self.contig, regs, int(clen))
        except Exception as e:  # noqa: BLE001 - the reference logs and goes on unmasked
            self._nmask = None
            logging.warning(f"Unable to mask N regions in coverage vector, reference could not be fetched: {e}")

    def to_task_input(self, task_id: int, sv_id_start: int, tandem_repeats, qc_nm_threshold: float) -> TaskInput:
        n = len(self._leads)
        L = empty_leads(n)
        qn, qrank = intern_sorted([ld.read_qname for ld in self._leads])
        psn, psrank = intern_sorted([ld.phase_set for ld in self._leads if ld.phase_set is not None] + ["NULL"])
        cn, crank = intern_sorted([ld.bnd_info.mate_contig for ld in self._leads if ld.bnd_info is not None] + [self.contig])
        # column at a time: one list comprehension and one array conversion per field (an element-wise fill of the numpy
        # columns costs a scalar store per field and lead)
        ls = self._leads

        def col(name, values):
            L[name][:] = np.fromiter(values, L[name].dtype, n) if n else L[name]
        col("svtype", (SVT[ld.svtype] for ld in ls))
        col("ref_start", (ld.ref_start for ld in ls)); col("ref_end", (ld.ref_end for ld in ls))
        col("qry_start", (ld.qry_start for ld in ls)); col("qry_end", (ld.qry_end for ld in ls))
        col("svlen", (SVLEN_NONE if ld.svlen is None else ld.svlen for ld in ls))
        col("read_len", (ld.read_len or 0 for ld in ls))
        col("qname_id", (qrank[ld.read_qname] for ld in ls))
        col("read_id", (ld.read_id for ld in ls))
        col("strand", (1 if ld.strand == "-" else 0 for ld in ls))
        col("mapq", (ld.mapq for ld in ls))
        col("nm", (float("nan") if ld.nm is None else ld.nm for ld in ls))
        col("source", (SRC[ld.source] for ld in ls))
        col("hap", (int(ld.hap) for ld in ls))
        col("ps_rank", (PS_NONE if ld.phase_set is None else psrank[ld.phase_set] for ld in ls))
        col("is_sa", (bool(ld.is_sa) for ld in ls))
        seqs = [None if ld.seq is None else ld.seq.encode("latin-1") for ld in ls]
        lens = np.fromiter((SEQ_NONE if q is None else len(q) for q in seqs), np.int64, n) if n else np.zeros(0, np.int64)
        L["seq_len"][:] = lens
        have = lens >= 0
        offs = np.zeros(n, np.int64)
        if n:
            offs[have] = (np.cumsum(np.where(have, lens, 0)) - np.where(have, lens, 0))[have]
        L["seq_off"][:] = offs
        pool = b"".join(q for q in seqs if q is not None)
        for i, ld in enumerate(ls):
            if ld.bnd_info is not None:
                L["mate_contig"][i] = crank[ld.bnd_info.mate_contig]
                L["mate_ref_start"][i] = ld.bnd_info.mate_ref_start
                L["bnd_is_first"][i] = bool(ld.bnd_info.is_first)
                L["bnd_is_reverse"][i] = bool(ld.bnd_info.is_reverse)
        reads = sorted(self._reads, key=lambda r: r[0])  # BAM order == ascending start; stable
        clen = self.contig_len if self.contig_len is not None else self.end
        ti = TaskInput(task_id=task_id, contig=self.contig, contig_len=int(clen), sv_id_start=sv_id_start, leads=L,
                       seq_pool=np.frombuffer(bytes(pool), np.uint8).copy(),
                       read_start=np.array([r[0] for r in reads], np.int32),
                       read_end=np.array([r[1] for r in reads], np.int32),
                       read_hp=np.array([r[2] for r in reads], np.uint8),
                       tr_start=None if tandem_repeats is None else np.array([t[0] for t in tandem_repeats], np.int32),
                       tr_end=None if tandem_repeats is None else np.array([t[1] for t in tandem_repeats], np.int32),
                       qc_nm_threshold=qc_nm_threshold, qnames=qn, ps_names=psn, contig_names=cn)
        if self._nmask is not None:
            ti.nmask_start, ti.nmask_end = self._nmask
        ti.validate()
        return ti
