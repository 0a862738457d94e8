"""Output record types with the reference's names and fields (reference `src/sniffles/sv.py:31-223`).

`SVCall` / `SVCallBNDInfo` here are plain dataclasses carrying the same attributes the reference's
downstream code reads (VCF writer, SNF writer, CallTask.execute).  When the library is used inside the
reference package, pass the reference's own classes to `materialize(..., svcall_cls=sniffles.sv.SVCall,
bnd_cls=sniffles.sv.SVCallBNDInfo)` so pickles carry the `sniffles.sv` module path.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

from .abi import FILTERS, Result, none_if_nan
from .records import bnd_alt, call_id
from .soa import SVTYPES

TYPES = ["INS", "DEL", "DUP", "INV", "BND"]
SINGLE_TYPES = ["SINGLE_LEFT", "SINGLE_RIGHT"]
ALL_TYPES = TYPES + SINGLE_TYPES


@dataclass
class SVCallBNDInfo:
    mate_contig: str
    mate_ref_start: int
    is_first: bool
    is_reverse: bool


@dataclass
class SVCallPostprocessingInfo:
    """Stands in for `SVCallPostprocessingInfo(cluster=...)`: the cluster lives in HBM; this is its handle."""
    batch: object
    index: int

    @property
    def cluster(self):
        raise AttributeError("cluster leads stay on the GPU; use the call's fields (support, rnames, ...) instead")


@dataclass
class SVCall:
    contig: str
    pos: int
    id: str
    ref: str
    alt: str
    qual: int
    filter: str
    info: dict
    svtype: str
    svlen: int
    end: int
    genotypes: dict
    precise: bool
    support: int
    rnames: Optional[list]
    qc: bool
    nm: float
    postprocess: Optional[SVCallPostprocessingInfo]
    svlens: Optional[list] = None
    fwd: int = None
    rev: int = None
    coverage_upstream: int = 0
    coverage_downstream: int = 0
    coverage_start: int = 0
    coverage_center: int = 0
    coverage_end: int = 0
    sample_internal_id: int = None
    bnd_info: SVCallBNDInfo = None
    support_inline: int = None
    support_splits: int = None
    raw_vcf_line: Optional[str] = None
    raw_vcf_line_index: Optional[int] = None

    def set_info(self, k, v):
        self.info[k] = v

    def get_info(self, k):
        return self.info[k] if k in self.info else None

    def has_info(self, k):
        return k in self.info

    @property
    def is_single_break(self) -> bool:
        return self.svtype.startswith("SINGLE")

    def finalize(self):
        self.postprocess = None


def _ps(code, ti):
    if code == -1:
        return None
    if code == -2:
        return "NULL"
    return ti.ps_name(int(code))


def fill_candidate(call, res: Result, i: int, ti, bnd_cls=SVCallBNDInfo):
    """Candidate-stage fields of SVCall `call` from record i (sv.call_from, sv.py:497-598)."""
    c = res.calls[i]
    svtype = SVTYPES[int(c["svtype"])]
    call.contig, call.pos, call.end, call.svtype, call.svlen = ti.contig, int(c["pos"]), int(c["end"]), svtype, int(c["svlen"])
    call.id = call_id(svtype, int(c["sv_id"]), ti.task_id)
    call.ref, call.alt = "N", f"<{svtype}>"
    call.qual, call.filter, call.qc = int(c["qual"]), FILTERS[int(c["filter"])], bool(c["qc"])
    call.precise, call.support = bool(c["precise"]), int(c["support"])
    call.fwd, call.rev, call.nm = int(c["fwd"]), int(c["rev"]), float(c["nm"])
    call.rnames = [ti.qname(int(q)) for q in res.rn(i)]
    call.genotypes = dict()
    info = dict()
    if svtype == "BND":
        bi = bnd_cls(mate_contig=ti.contig_name(int(c["mate_contig"])), mate_ref_start=int(c["mate_ref_start"]),
                     is_first=bool(c["bnd_is_first"]), is_reverse=bool(c["bnd_is_reverse"]))
        call.bnd_info = bi
        call.alt = bnd_alt(bi.mate_contig, bi.mate_ref_start, bi.is_first, bi.is_reverse)
        info["CHR2"] = bi.mate_contig
    elif svtype == "INS":
        info["SUPPORT_LONG"] = int(c["support_long"])
    elif svtype == "DEL":
        info["SUPPORT_SA"] = int(c["support_sa"])
    info["STDEV_POS"] = float(c["stdev_pos"])
    sl = none_if_nan(c["stdev_len"])
    if sl is not None:
        info["STDEV_LEN"] = sl
    call.info = info
    (call.coverage_upstream, call.coverage_start, call.coverage_center, call.coverage_end,
     call.coverage_downstream) = (int(x) for x in c["cov"])
    return call


def fill_final(call, res: Result, i: int, ti):
    """Fields set by Task.finalize_candidates (parallel.py:129-201) from record i."""
    c = res.calls[i]
    call.qc, call.filter = bool(c["qc"]), FILTERS[int(c["filter"])]
    call.info["COVERAGE_VAR"] = None  # qc_coverage_samples() always yields (True, None), postprocessing.py:373-374
    if c["ph_set"]:
        call.info["PHASE"] = (f"{int(c['ph_hp'])},{_ps(int(c['ph_ps']), ti)},{int(c['ph_hp_support'])},"
                              f"{int(c['ph_ps_support'])},{'PASS' if c['ph_hp_pass'] else 'FAIL'},"
                              f"{'PASS' if c['ph_ps_pass'] else 'FAIL'}")
    if c["gt_set"]:
        hp = None if c["gt_hp"] < 0 else str(int(c["gt_hp"]))
        call.genotypes[0] = (int(c["gt_a"]), int(c["gt_b"]), int(c["gt_gq"]), int(c["gt_dr"]), int(c["gt_dv"]),
                             (hp, _ps(int(c["gt_ps"]), ti)))
        call.info["VAF"] = float(c["vaf"])
    alt = res.alt(i)
    if alt is not None:
        call.alt = alt
    return call


def new_call(svcall_cls=SVCall):
    return svcall_cls(contig=None, pos=0, id="", ref="N", alt="", qual=0, filter="PASS", info=dict(), svtype="",
                      svlen=0, end=0, genotypes=dict(), precise=False, support=0, rnames=None, qc=True, nm=-1,
                      postprocess=None)
