"""Struct-of-arrays layout for SV signatures ("leads") at the C-ABI boundary.

One `TaskInput` = one contig task (reference: one `CallTask` per contig,
`src/sniffles/sniffles:298-358`).  A `Lead` of the reference
(`src/sniffles/leadprov.py:34-56`) becomes one row of the arrays below, in
arrival (BAM) order, which is the order `LeadProvider.record_lead`
(`leadprov.py:400-418`) sees them.  Strings that the hot path only compares
(read names, mate contigs, phase sets) are interned to integers whose ORDER
matches Python string order, because the reference breaks ties on string
order (`util.most_common`, `util.py:91-103`).

The same layout is consumed by the CPU oracle (tests only) and by the HIP
library (`include/sniffles_amd.h`).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import numpy as np

# svtype codes follow sv.ALL_TYPES order (`sv.py:31-33`)
SVTYPES = ["INS", "DEL", "DUP", "INV", "BND", "SINGLE_LEFT", "SINGLE_RIGHT"]
SVT = {n: i for i, n in enumerate(SVTYPES)}
N_SVTYPES = 7

SOURCES = ["INLINE", "SPLIT_PRIM", "SPLIT_SUP", "BND_SA"]
SRC = {n: i for i, n in enumerate(SOURCES)}

SVLEN_NONE = np.int32(-2 ** 31)   # Lead.svlen is None (clipped long-INS hint, leadprov.py:639-653)
SEQ_NONE = np.int32(-1)           # Lead.seq is None
PS_NONE = np.int32(-1)            # Lead.phase_set is None (BND_SA leads, leadprov.py:113-131)

LEAD_FIELDS = [
    ("ref_start", np.int32), ("ref_end", np.int32),
    ("qry_start", np.int32), ("qry_end", np.int32),
    ("svlen", np.int32), ("read_len", np.int32),
    ("qname_id", np.uint32), ("read_id", np.uint32),
    ("ps_rank", np.int32),
    ("mate_contig", np.int32), ("mate_ref_start", np.int32),
    ("seq_len", np.int32), ("seq_off", np.int64),
    ("nm", np.float64),
    ("svtype", np.uint8), ("strand", np.uint8), ("mapq", np.uint8),
    ("source", np.uint8), ("hap", np.uint8), ("is_sa", np.uint8),
    ("bnd_is_first", np.uint8), ("bnd_is_reverse", np.uint8),
]


def nmask_intervals(sequence, contig_len: int = None):
    """`_mask_N_coverage` masks `coverage[mask == 78]` (leadprov.py:434-441): the runs of the byte 'N' (upper case only) of the
    reference sequence as sorted, disjoint intervals (start[], end[]), clipped to the contig."""
    if isinstance(sequence, str):
        sequence = sequence.encode("ascii")
    a = np.frombuffer(sequence, np.uint8) if not isinstance(sequence, np.ndarray) else sequence
    if contig_len is not None:
        a = a[:contig_len]
    m = np.concatenate([[0], (a == 78).astype(np.int8), [0]])
    d = np.diff(m)
    return np.flatnonzero(d == 1).astype(np.int32), np.flatnonzero(d == -1).astype(np.int32)


def paint_nmask(fetch, contig: str, regions, contig_len: int):
    """The mask `_mask_N_coverage(regions)` paints (leadprov.py:431-441), as sorted disjoint intervals (start[], end[]):
    `mask = zeros(len(coverage))`, then per region IN LIST ORDER `mask[start:end] = fetch(contig, start, end)` - a later region
    overwrites an earlier one where they overlap - and coverage is zeroed where the mask byte is 'N'.  `fetch`: pysam's
    `FastaFile.fetch`.  `regions`: [(start, end)], or None for the whole contig (`fetch(contig)`, the mask then is the boolean index
    itself and must have the coverage vector's length).  Raises what the reference's numpy statements raise (a fetched sequence
    whose length is neither the slice's nor 1; a whole-contig sequence of another length than the coverage vector): the caller
    logs and leaves the task unmasked, like the reference."""
    def as_bytes(x):
        return x.encode("ascii") if isinstance(x, str) else bytes(x)
    if regions is None:
        seq = as_bytes(fetch(contig))
        if len(seq) != contig_len:      # `self.coverage[mask == 78] = 0`: boolean index of another length -> IndexError
            raise IndexError(f"boolean index did not match indexed array: dimension is {contig_len} but corresponding boolean dimension is {len(seq)}")
        return nmask_intervals(seq)
    cur = []                            # sorted disjoint [s, e) painted 'N' so far
    for start, end in regions:
        start, end = int(start), int(end)
        seq = as_bytes(fetch(contig, start, end))
        lo, hi, _ = slice(start, end).indices(contig_len)
        width = max(0, hi - lo)
        if len(seq) != width and len(seq) != 1:
            raise ValueError(f"could not broadcast input array from shape ({len(seq)},) into shape ({width},)")
        if width == 0:
            continue
        if len(seq) == 1 and width != 1:
            seq = seq * width
        cut = []                        # the slice is overwritten: what was painted inside it goes first
        for s, e in cur:
            if e <= lo or s >= hi:
                cut.append((s, e))
            else:
                if s < lo:
                    cut.append((s, lo))
                if e > hi:
                    cut.append((hi, e))
        ns, ne = nmask_intervals(seq)
        cut.extend((int(a) + lo, int(b) + lo) for a, b in zip(ns.tolist(), ne.tolist()))
        cut.sort()
        cur = []
        for s, e in cut:                # neighbours that touch become one interval (the library wants them disjoint and sorted)
            if cur and s <= cur[-1][1]:
                cur[-1] = (cur[-1][0], max(cur[-1][1], e))
            else:
                cur.append((s, e))
    return np.array([s for s, _ in cur], np.int32), np.array([e for _, e in cur], np.int32)


@dataclass
class TaskInput:
    """All inputs of one contig task, SoA."""
    task_id: int
    contig: str
    contig_len: int
    sv_id_start: int = 0
    # per lead arrays (length n_leads), see LEAD_FIELDS
    leads: dict = field(default_factory=dict)
    seq_pool: np.ndarray = field(default_factory=lambda: np.zeros(0, np.uint8))
    # per read (alignment record) arrays: coverage + REF haplotype counts
    read_start: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    read_end: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    read_hp: np.ndarray = field(default_factory=lambda: np.zeros(0, np.uint8))
    # tandem repeats, already padded (util.load_tandem_repeats, util.py:121-147); None = no annotation
    tr_start: Optional[np.ndarray] = None
    tr_end: Optional[np.ndarray] = None
    # LeadProvider._mask_N_coverage (leadprov.py:420-443): intervals [start, end) where the reference base is 'N' (coverage
    # reads as 0 there); None = no reference given
    nmask_start: Optional[np.ndarray] = None
    nmask_end: Optional[np.ndarray] = None
    # side channel written by iter_region (leadprov.py:577-578)
    qc_nm_threshold: float = 0.02
    # string tables for materialisation (host only)
    qnames: Optional[list] = None        # qname_id -> str (None: synthesised as f"q{id}")
    ps_names: Optional[list] = None      # ps_rank -> str, sorted in Python str order
    contig_names: Optional[list] = None  # mate_contig rank -> str, sorted in Python str order

    @property
    def n_leads(self) -> int:
        return int(self.leads["ref_start"].shape[0]) if self.leads else 0

    @property
    def n_reads(self) -> int:
        return int(self.read_start.shape[0])

    def check_layout(self) -> None:
        """Types, shapes and contiguity only - what has to hold before raw pointers are handed to the library.  The values
        (type codes, haplotypes, sequence ranges, read intervals) are checked by the library while it stages the columns
        (`snf_batch_upload` fails with the reason), so the binding does not walk every column a second time."""
        n = self.n_leads
        for name, dt in LEAD_FIELDS:
            a = self.leads[name]
            if a.dtype != dt or a.shape != (n,) or not a.flags.c_contiguous:
                raise ValueError(f"lead field {name}: want {np.dtype(dt)}[{n}] contiguous, got {a.dtype}{a.shape}")
        if self.seq_pool.dtype != np.uint8:
            raise ValueError("seq_pool must be uint8")
        for a, dt in ((self.read_start, np.int32), (self.read_end, np.int32), (self.read_hp, np.uint8)):
            if a.dtype != dt or a.shape != (self.n_reads,):
                raise ValueError("read arrays malformed")
        if (self.tr_start is None) != (self.tr_end is None):
            raise ValueError("tr_start/tr_end must both be set or both None")

    def validate(self) -> None:
        self.check_layout()
        sl, so = self.leads["seq_len"], self.leads["seq_off"]
        m = sl >= 0
        if m.any() and int((so[m] + sl[m]).max()) > self.seq_pool.shape[0]:
            raise ValueError("seq_off+seq_len exceeds seq_pool")
        if np.any(self.leads["hap"] > 2):
            raise ValueError("hap must be 0,1,2 (leadprov.py:403 indexes a 3-array with int(ld.hap))")
        if np.any(self.leads["svtype"] >= N_SVTYPES):
            raise ValueError("svtype code out of range")

    def set_nmask(self, sequence) -> None:
        """`sequence`: the contig's reference bases (str / bytes / uint8 array, as `fasta.fetch(contig)` returns them)"""
        self.nmask_start, self.nmask_end = nmask_intervals(sequence, self.contig_len)

    def qname(self, qid: int) -> str:
        return self.qnames[qid] if self.qnames is not None else f"q{qid}"

    def ps_name(self, rank: int) -> Optional[str]:
        if rank < 0:
            return None
        return self.ps_names[rank] if self.ps_names is not None else str(rank)

    def contig_name(self, rank: int) -> str:
        return self.contig_names[rank] if self.contig_names is not None else f"ctg{rank}"


def empty_leads(n: int) -> dict:
    d = {name: np.zeros(n, dt) for name, dt in LEAD_FIELDS}
    d["seq_len"][:] = SEQ_NONE
    d["ps_rank"][:] = PS_NONE
    return d


def concat_leads(parts: list) -> dict:
    return {name: np.ascontiguousarray(np.concatenate([p[name] for p in parts])) if parts else np.zeros(0, dt)
            for name, dt in LEAD_FIELDS}


def concat_tasks(parts: list) -> "TaskInput":
    """The tasks of several regions of ONE contig as one task, in list order: what `LeadProvider.build_leadtab` holds after
    it has walked a region list (leadprov.py:445-472) - leads and reads appended region by region (a read that overlaps two
    regions is there twice, as in the reference), the NM threshold of the last region (iter_region's side channel is
    overwritten per region).  The parts must come from the same record table (same read-name and contig tables)."""
    first = parts[0]
    if len(parts) == 1:
        return first
    ps_union = sorted(set(n for p in parts for n in (p.ps_names or [])))
    ps_rank = {n: i for i, n in enumerate(ps_union)}
    lead_parts, pool_parts, pool_len = [], [], 0
    for p in parts:
        if p.contig != first.contig or p.qnames is not first.qnames and p.qnames != first.qnames:
            raise ValueError("concat_tasks: the parts must be regions of one contig of one record table")
        d = {k: v.copy() for k, v in p.leads.items()}
        has_seq = d["seq_len"] >= 0
        d["seq_off"][has_seq] += pool_len
        if p.ps_names is not None and len(p.ps_names):
            remap = np.array([ps_rank[n] for n in p.ps_names], np.int32)
            has_ps = d["ps_rank"] >= 0
            d["ps_rank"][has_ps] = remap[d["ps_rank"][has_ps]]
        lead_parts.append(d)
        pool_parts.append(p.seq_pool)
        pool_len += int(p.seq_pool.shape[0])
    return TaskInput(task_id=first.task_id, contig=first.contig, contig_len=first.contig_len, sv_id_start=first.sv_id_start,
                     leads=concat_leads(lead_parts), seq_pool=np.ascontiguousarray(np.concatenate(pool_parts)),
                     read_start=np.ascontiguousarray(np.concatenate([p.read_start for p in parts])),
                     read_end=np.ascontiguousarray(np.concatenate([p.read_end for p in parts])),
                     read_hp=np.ascontiguousarray(np.concatenate([p.read_hp for p in parts])),
                     tr_start=first.tr_start, tr_end=first.tr_end, qc_nm_threshold=parts[-1].qc_nm_threshold,
                     qnames=first.qnames, ps_names=ps_union if any(p.ps_names is not None for p in parts) else None,
                     contig_names=first.contig_names)


def intern_sorted(strings) -> tuple:
    """Intern strings to ranks whose integer order equals Python str order."""
    uniq = sorted(set(strings))
    rank = {s: i for i, s in enumerate(uniq)}
    return uniq, rank


class DeviceTaskInput(TaskInput):
    """A task whose leads, sequence pool and read table were born in HBM (sniffles_amd.extract.Extractor) and stay there:
    `lib.Batch` hands them to the clustering path device-to-device (`snf_batch_add_task_device`).  The host keeps what it
    needs to format results (names, phase-set table, sizes); the columns are copied to the host only if somebody asks for
    `.leads` / `.seq_pool` / `.read_*` (e.g. the cluster views of seam B3)."""

    def __init__(self, extractor, n_leads: int, n_reads: int, **kw):
        self._extractor, self._n_leads, self._n_reads, self._host = extractor, int(n_leads), int(n_reads), None
        super().__init__(**kw)

    def _pull(self):
        if self._host is None:
            self._host = self._extractor.host_columns()
        return self._host

    @property
    def n_leads(self) -> int:
        return self._n_leads

    @property
    def n_reads(self) -> int:
        return self._n_reads

    leads = property(lambda self: self._pull()["leads"], lambda self, v: None)
    seq_pool = property(lambda self: self._pull()["seq_pool"], lambda self, v: None)
    read_start = property(lambda self: self._pull()["read_start"], lambda self, v: None)
    read_end = property(lambda self: self._pull()["read_end"], lambda self, v: None)
    read_hp = property(lambda self: self._pull()["read_hp"], lambda self, v: None)

    def validate(self) -> None:      # the columns were produced by the extraction kernels
        pass
