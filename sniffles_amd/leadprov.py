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


def iter_leads(ti):
    """The rows of a TaskInput as `Lead` objects, in arrival order - what a ported `iter_region` would hand to `record_lead`
    (the inverse of `LeadProvider.to_task_input`; used by tests and by bench.py's ingest measurement)."""
    from .soa import SVTYPES, SOURCES
    L = ti.leads
    pool = ti.seq_pool.tobytes()
    cols = {k: L[k].tolist() for k in L}
    for i in range(ti.n_leads):
        svt = SVTYPES[cols["svtype"][i]]
        sl, so, svlen = cols["seq_len"][i], cols["seq_off"][i], cols["svlen"][i]
        ld = Lead(read_id=cols["read_id"][i], read_qname=ti.qname(cols["qname_id"][i]), contig=ti.contig,
                  ref_start=cols["ref_start"][i], ref_end=cols["ref_end"][i], qry_start=cols["qry_start"][i], qry_end=cols["qry_end"][i],
                  strand="-" if cols["strand"][i] else "+", mapq=cols["mapq"][i], nm=cols["nm"][i], source=SOURCES[cols["source"][i]], svtype=svt,
                  svlen=None if svlen == int(SVLEN_NONE) else svlen, seq=None if sl < 0 else pool[so:so + sl].decode("latin-1"),
                  hap=str(cols["hap"][i]), phase_set=ti.ps_name(cols["ps_rank"][i]), is_sa=bool(cols["is_sa"][i]), read_len=cols["read_len"][i])
        if svt == "BND":
            ld.bnd_info = SVCallBNDInfo(ti.contig_name(cols["mate_contig"][i]), cols["mate_ref_start"][i], bool(cols["bnd_is_first"][i]),
                                        bool(cols["bnd_is_reverse"][i]))
        yield ld


class LeadProvider:
    def __init__(self, config, read_id_offset, contig: str, contig_len: int = None, record_time_columns: bool = None):
        self.config = config
        self.contig = contig
        self.contig_len = contig_len
        self.start = None
        self.end = None
        self.read_id = read_id_offset
        self.read_count = 0
        self._leads = []
        import array
        self._rs, self._re, self._rhp = array.array("i"), array.array("i"), array.array("B")      # typed growable read columns
        self._nmask = None
        # Where the reference pays its binning - at record time (leadprov.py:400-418) - a lead becomes a row of typed columns here
        # (`_snf_fast.LeadSink`: attributes read once, names interned, `seq` appended to the pool), so that `to_task_input` inside
        # `Task.call_candidates` is a copy of finished columns.  The row is a snapshot: a lead mutated after `record_lead` keeps its
        # recorded values (the reference's callers record finished leads).  `record_time_columns=False` / SNF_LEADS_LATE=1 (or no C
        # extension): the objects are kept and walked once in `to_task_input`, as before.
        self._sink = None
        import os
        if record_time_columns is None:
            record_time_columns = os.environ.get("SNF_LEADS_LATE") != "1"
        if record_time_columns:
            from .sv import _load_fast
            fast = _load_fast()
            if fast is not None and hasattr(fast, "LeadSink"):
                self._sink = fast.LeadSink(SVT, SRC, int(SVLEN_NONE), int(SEQ_NONE), int(PS_NONE), contig)
                self.record_lead = self._sink.record          # record_lead(ld[, pos_leadtab]) straight into the sink

    def record_lead(self, ld: Lead, pos_leadtab: int = None) -> None:
        """Same call as the reference; `pos_leadtab` (the 100-bp bin) is recomputed on the GPU."""
        self._leads.append(ld)

    def record_read(self, ref_start: int, ref_end: int, hp: int = 0) -> None:
        self._rs.append(int(ref_start)); self._re.append(int(ref_end)); self._rhp.append(int(hp))
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

    def _columns_fast(self):
        """The lead columns in ONE walk over the Lead objects (`_snf_fast.lead_columns`, csrc/snf_pyfast.c) instead of one Python
        generator pass per field: (columns, qnames, ps names, contig names, pool), or None when the extension is not built."""
        from .sv import _load_fast
        fast = _load_fast()
        if fast is None or not hasattr(fast, "lead_columns"):
            return None
        if self._sink is not None:
            if self._leads:
                raise RuntimeError("leads were appended behind the record-time sink")
            n = len(self._sink)
            L = empty_leads(n)
            ql, pl, cl, pool = self._sink.take(L)
        else:
            n = len(self._leads)
            L = empty_leads(n)
            ql, pl, cl, pool = fast.lead_columns(self._leads, L, SVT, SRC, int(SVLEN_NONE), int(SEQ_NONE), int(PS_NONE), self.contig)

        def ranks(first_seen, extra=()):
            """first-seen indices -> ranks in Python string order (the reference breaks ties on string order)"""
            have = set(first_seen) if extra else ()
            names = list(first_seen) + [x for x in extra if x not in have]
            if hasattr(fast, "rank_strings"):
                in_order, rank = fast.rank_strings(names)
                return in_order, np.frombuffer(rank, np.int64)
            order = sorted(range(len(names)), key=names.__getitem__)
            rank = np.empty(len(names), np.int64)
            rank[order] = np.arange(len(names))
            return [names[j] for j in order], rank
        qn, qr = ranks(ql)
        if n:
            L["qname_id"][:] = qr[L["qname_id"]]
        psn, pr = ranks(pl, ("NULL",))
        if n:
            has = L["ps_rank"] != PS_NONE
            L["ps_rank"][has] = pr[L["ps_rank"][has]]
        cn, cr = ranks(cl)
        if n:
            bnd = L["mate_contig"] >= 0
            L["mate_contig"][bnd] = cr[L["mate_contig"][bnd]]
            L["mate_contig"][~bnd] = 0
        return L, qn, psn, cn, pool

    def to_task_input(self, task_id: int, sv_id_start: int, tandem_repeats, qc_nm_threshold: float) -> TaskInput:
        fastcols = self._columns_fast() if not getattr(self, "_force_py", False) else None
        if fastcols is not None:
            L, qn, psn, cn, pool = fastcols
            return self._finish_task_input(L, qn, psn, cn, pool, task_id, sv_id_start, tandem_repeats, qc_nm_threshold)
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
        return self._finish_task_input(L, qn, psn, cn, bytes(pool), task_id, sv_id_start, tandem_repeats, qc_nm_threshold)

    def _finish_task_input(self, L, qn, psn, cn, pool, task_id, sv_id_start, tandem_repeats, qc_nm_threshold) -> TaskInput:
        rs, re_, rhp = (np.frombuffer(a, dt) if len(a) else np.zeros(0, dt) for a, dt in ((self._rs, np.int32), (self._re, np.int32), (self._rhp, np.uint8)))
        order = np.argsort(rs, kind="stable")  # BAM order == ascending start; stable
        clen = self.contig_len if self.contig_len is not None else self.end
        ti = TaskInput(task_id=task_id, contig=self.contig, contig_len=int(clen), sv_id_start=sv_id_start, leads=L,
                       seq_pool=np.frombuffer(bytes(pool), np.uint8).copy(),
                       read_start=np.ascontiguousarray(rs[order]), read_end=np.ascontiguousarray(re_[order]), read_hp=np.ascontiguousarray(rhp[order]),
                       tr_start=None if tandem_repeats is None else np.array([t[0] for t in tandem_repeats], np.int32),
                       tr_end=None if tandem_repeats is None else np.array([t[1] for t in tandem_repeats], np.int32),
                       qc_nm_threshold=qc_nm_threshold, qnames=qn, ps_names=psn, contig_names=cn)
        if self._nmask is not None:
            ti.nmask_start, ti.nmask_end = self._nmask
        ti.validate()
        return ti
