"""Golden-case catalogue: seeded synthetic tasks, fuzz tasks and hand-made edge cases.

Every case is (name, TaskInput builder, config kwargs, reference CLI args).  The
expected outputs under tests/golden/ are produced from the UNMODIFIED reference
by oracle/make_golden.py (build container only) and are what pins the oracle.
"""
from __future__ import annotations

import numpy as np

from sniffles_amd import synth
from sniffles_amd.soa import (TaskInput, SVT, SRC, SVLEN_NONE, SEQ_NONE, PS_NONE, empty_leads, intern_sorted)


def mk_task(leads, reads, contig_len, trs=None, task_id=0, sv_id_start=0, qc_nm_threshold=0.02,
            contig="chrT") -> TaskInput:
    """Hand-made task.  `leads`: list of dicts with reference Lead attribute names
    (svtype, ref_start, svlen, read, strand '+'/'-', seq, hap, ps, mate=(contig,pos,is_first,is_reverse) ...)."""
    n = len(leads)
    L = empty_leads(n)
    qn, qrank = intern_sorted([str(d.get("read", f"r{i}")) for i, d in enumerate(leads)])
    ps_all = [d["ps"] for d in leads if d.get("ps") is not None] + ["NULL"]
    psn, psrank = intern_sorted(ps_all)
    ctg_all = [d["mate"][0] for d in leads if d.get("mate")] + ["chr1"]
    cn, crank = intern_sorted(ctg_all)
    pool = bytearray()
    read_ids = {}
    for i, d in enumerate(leads):
        t = d["svtype"]
        L["svtype"][i] = SVT[t]
        rs = int(d["ref_start"])
        svlen = d.get("svlen", 0)
        L["ref_start"][i] = rs
        L["svlen"][i] = SVLEN_NONE if svlen is None else int(svlen)
        L["ref_end"][i] = d.get("ref_end", rs - abs(svlen or 0) if t == "DEL" else rs)
        L["qry_start"][i] = d.get("qry_start", 1000 + i * 7)
        L["qry_end"][i] = d.get("qry_end", L["qry_start"][i] + (svlen if (t == "INS" and svlen) else 0))
        L["read_len"][i] = d.get("read_len", 20000)
        q = str(d.get("read", f"r{i}"))
        L["qname_id"][i] = qrank[q]
        L["read_id"][i] = d.get("read_id", read_ids.setdefault(q, len(read_ids) + 1))
        L["strand"][i] = 1 if d.get("strand", "+") == "-" else 0
        L["mapq"][i] = d.get("mapq", 60)
        L["nm"][i] = d.get("nm", 0.01)
        L["source"][i] = SRC[d.get("source", "BND_SA" if t == "BND" else "INLINE")]
        L["hap"][i] = int(d.get("hap", 0))
        ps = d.get("ps", None if t == "BND" else "NULL")
        L["ps_rank"][i] = PS_NONE if ps is None else psrank[ps]
        L["is_sa"][i] = int(d.get("is_sa", 0))
        seq = d.get("seq")
        if seq is not None:
            L["seq_off"][i] = len(pool)
            L["seq_len"][i] = len(seq)
            pool += seq.encode("latin-1")
        else:
            L["seq_len"][i] = SEQ_NONE
        if d.get("mate"):
            mc, mp, fi, rv = d["mate"]
            L["mate_contig"][i] = crank[mc]
            L["mate_ref_start"][i] = mp
            L["bnd_is_first"][i] = int(fi)
            L["bnd_is_reverse"][i] = int(rv)
    rs = np.array([r[0] for r in reads], np.int32)
    re_ = np.array([r[1] for r in reads], np.int32)
    rh = np.array([r[2] if len(r) > 2 else 0 for r in reads], np.uint8)
    o = np.argsort(rs, kind="stable")
    ti = TaskInput(task_id=task_id, contig=contig, contig_len=contig_len, sv_id_start=sv_id_start, leads=L,
                   seq_pool=np.frombuffer(bytes(pool), np.uint8).copy(),
                   read_start=rs[o], read_end=re_[o], read_hp=rh[o],
                   tr_start=None if trs is None else np.array([t[0] for t in trs], np.int32),
                   tr_end=None if trs is None else np.array([t[1] for t in trs], np.int32),
                   qc_nm_threshold=qc_nm_threshold, qnames=qn, ps_names=psn, contig_names=cn)
    ti.validate()
    return ti


def _reads(n, s, e, hp=0):
    return [(s, e, hp)] * n


def _rng_seq(rng, n):
    return "".join("ACGT"[i] for i in rng.integers(0, 4, n))


def _mutate(rng, s, rate):
    b = list(s)
    for i in range(len(b)):
        if rng.random() < rate:
            b[i] = "ACGT"[rng.integers(0, 4)]
    return "".join(b)


# ----------------------------------------------------------------------------- hand-made cases
def case_resplit_wrap():
    """svlens {60,80,120} -> bins 60,80,120 -> wrap-around merge into one cluster."""
    leads = []
    for i, sl in enumerate([60, 60, 80, 80, 120, 120, 61, 119]):
        leads.append(dict(svtype="DEL", ref_start=10050 + i, svlen=-sl, read=f"a{i}", strand="+-"[i % 2]))
    return mk_task(leads, _reads(20, 0, 30000), 30000)


def case_merge_index_rule():
    """chains of adjacent / one-bin-apart seed clusters with small and large stdev."""
    leads = []
    k = 0
    for b in [10000, 10100, 10200, 10400, 10500, 10700, 10900, 11200, 11300, 11400, 11600]:
        for j in range(3):
            leads.append(dict(svtype="DEL", ref_start=b + [1, 50, 98][j], svlen=-(200 + (k % 3)), read=f"m{k}",
                              strand="+-"[k % 2]))
            k += 1
    return mk_task(leads, _reads(25, 0, 40000), 40000)


def case_bnd_first_error():
    """a task whose only candidates are BNDs raises UnboundLocalError in coverage()."""
    leads = [dict(svtype="BND", ref_start=5000 + i % 2, read=f"b{i}", strand="+-"[i % 2],
                  mate=("chr2", 70000 + i, True, False)) for i in range(6)]
    return mk_task(leads, _reads(20, 0, 20000), 20000)


def case_bnd_stale_end():
    """BND inherits `end` of the last non-BND candidate; mixed mates exercise resplit_bnd."""
    leads = [dict(svtype="DEL", ref_start=3000 + i % 3, svlen=-400, read=f"d{i}", strand="+-"[i % 2]) for i in range(6)]
    leads += [dict(svtype="INV", ref_start=8000 + i % 2, svlen=900, read=f"v{i}", strand="+-"[i % 2], source="SPLIT_SUP")
              for i in range(5)]
    mates = [("chr2", 70000, True, False), ("chr2", 70010, True, False), ("chr2", 71200, True, False),
             ("chr2", 74000, True, False), ("chr10", 500, False, True), ("chr10", 650, False, True),
             ("chr2", 70020, False, False), ("chr2", 70030, True, True), ("chr2", 69990, True, False)]
    leads += [dict(svtype="BND", ref_start=12000 + (i % 3), read=f"b{i}", strand="+-"[i % 2], mate=m)
              for i, m in enumerate(mates)]
    leads += [dict(svtype="BND", ref_start=40, read=f"e{i}", strand="+-"[i % 2], mate=("chr1", 900 + i, True, False))
              for i in range(4)]  # start-100 < 0 -> numpy negative-index wrap
    leads += [dict(svtype="SINGLE_LEFT", ref_start=15000, svlen=0, read=f"s{i}") for i in range(3)]
    return mk_task(leads, _reads(18, 0, 19950) + _reads(5, 100, 9000, 1), 20000)


def case_merge_inner():
    """merge_inner (cluster.py:85-122): same-read fusion within cluster_merge_pos, strand guard,
    None-propagating seq, and a repeat cluster (threshold -1) that fuses everything per read."""
    rng = np.random.default_rng(11)
    allele = _rng_seq(rng, 300)
    leads = []
    for i in range(8):
        q = 2000 + i
        if i < 5:  # two pieces on one read, close on ref and query
            leads.append(dict(svtype="INS", ref_start=5000, svlen=150, read=f"r{i}", qry_start=q, qry_end=q + 150,
                              seq=_mutate(rng, allele[:150], 0.03), strand="+-"[i % 2]))
            leads.append(dict(svtype="INS", ref_start=5060, svlen=150, read=f"r{i}", qry_start=q + 210, qry_end=q + 360,
                              seq=_mutate(rng, allele[150:], 0.03) if i != 2 else None, strand="+-"[i % 2]))
        elif i == 5:  # far on the query: no fusion
            leads.append(dict(svtype="INS", ref_start=5001, svlen=150, read=f"r{i}", qry_start=q, qry_end=q + 150, seq=allele[:150]))
            leads.append(dict(svtype="INS", ref_start=5061, svlen=150, read=f"r{i}", qry_start=q + 2000, qry_end=q + 2150, seq=allele[150:]))
        elif i == 6:  # strand differs: no fusion
            leads.append(dict(svtype="INS", ref_start=5002, svlen=150, read=f"r{i}", qry_start=q, qry_end=q + 150, seq=allele[:150], strand="+"))
            leads.append(dict(svtype="INS", ref_start=5062, svlen=150, read=f"r{i}", qry_start=q + 200, qry_end=q + 350, seq=allele[150:], strand="-"))
        else:
            leads.append(dict(svtype="INS", ref_start=5003, svlen=300, read=f"r{i}", qry_start=q, qry_end=q + 300, seq=_mutate(rng, allele, 0.03)))
    # repeat region: DEL pieces on the same reads far apart still fuse (threshold -1)
    for i in range(6):
        leads.append(dict(svtype="DEL", ref_start=9100 + i, svlen=-70, read=f"t{i}", qry_start=500, strand="+-"[i % 2]))
        leads.append(dict(svtype="DEL", ref_start=9190 - i, svlen=-90, read=f"t{i}", qry_start=9000, strand="-+"[i % 2]))
    return mk_task(leads, _reads(22, 0, 15000), 15000, trs=[(8500, 9800)])


def case_long_ins():
    """INS >= long_ins_length: leads_long support union, SUPPORT_LONG, rescale_support, z-score exemption."""
    rng = np.random.default_rng(5)
    allele = _rng_seq(rng, 2700)
    leads = []
    for i in range(4):
        leads.append(dict(svtype="INS", ref_start=20010 + i, svlen=2700 + i, read=f"f{i}", seq=_mutate(rng, allele, 0.05) + "A" * i,
                          strand="+-"[i % 2]))
    for i in range(7):
        leads.append(dict(svtype="INS", ref_start=20020 + i, svlen=None, read=f"c{i}" if i < 5 else f"f{i - 5}", strand="+-"[i % 2]))
    # a second bin with only clipped leads (dropped) and one with 1 normal + clips (dropped: <2 normal)
    for i in range(3):
        leads.append(dict(svtype="INS", ref_start=30010, svlen=None, read=f"x{i}"))
    leads.append(dict(svtype="INS", ref_start=31010, svlen=500, read="y0", seq=_rng_seq(rng, 500)))
    leads.append(dict(svtype="INS", ref_start=31011, svlen=None, read="y1"))
    return mk_task(leads, _reads(30, 0, 50000), 50000)


def case_gt_failed_and_edges():
    """Zero coverage -> GT_FAILED; calls next to the contig end -> IndexError samples keep 0."""
    leads = [dict(svtype="DEL", ref_start=1000 + i, svlen=-300, read=f"z{i}", strand="+-"[i % 2]) for i in range(5)]
    leads += [dict(svtype="DEL", ref_start=9990 - i, svlen=-120, read=f"w{i}", strand="+-"[i % 2]) for i in range(5)]
    leads += [dict(svtype="DUP", ref_start=9800 + i, svlen=700, read=f"u{i}", strand="+-"[i % 2], source="SPLIT_SUP") for i in range(4)]
    return mk_task(leads, _reads(12, 5000, 10000), 10000)


def case_compute_metrics_big():
    """compute_metrics with > 100 leads: 150 and 250 leads per bin; seq cap at 10 per bin."""
    rng = np.random.default_rng(3)
    allele = _rng_seq(rng, 100)
    leads = []
    for i in range(150):
        leads.append(dict(svtype="DEL", ref_start=7000 + (i * 7) % 100, svlen=-(100 + i % 5), read=f"p{i}", strand="+-"[i % 2]))
    for i in range(250):
        leads.append(dict(svtype="INS", ref_start=7200 + (i * 3) % 100, svlen=100, read=f"q{i}", seq=_mutate(rng, allele, 0.05),
                          strand="+-"[i % 2]))
    for i in range(130):
        leads.append(dict(svtype="INS", ref_start=7300 + (i * 11) % 100, svlen=100 + (i % 3), read=f"q{i + 300}",
                          seq=_mutate(rng, allele, 0.05) + "C" * (i % 3), strand="+-"[i % 2]))
    return mk_task(leads, _reads(260, 0, 20000), 20000)


def case_phase_rescue():
    """phase_sv majorities, hom-alt phase override, rescue_phasing of a MOSAIC_VAF call via hap counts."""
    leads = []
    # het on hap 1, low VAF (reads of hap 1 only, few) -> MOSAIC_VAF, rescued (sv_reads/all_reads >= .75)
    for i in range(5):
        leads.append(dict(svtype="DEL", ref_start=4000 + i % 2, svlen=-250, read=f"h{i}", hap=1, ps="4001", strand="+-"[i % 2],
                          nm=0.01 + i * 0.003))
    # hom-alt with phased reads -> phase tuple forced from PHASE info
    for i in range(12):
        leads.append(dict(svtype="DEL", ref_start=12000 + i % 3, svlen=-500, read=f"g{i}", hap=2 if i < 9 else 1,
                          ps="12001" if i < 9 else "NULL", strand="+-"[i % 2]))
    # conflicting phases -> FAIL
    for i in range(8):
        leads.append(dict(svtype="INS", ref_start=16000 + i % 2, svlen=200, read=f"k{i}", hap=1 + i % 2, ps=["9", "10"][i % 2],
                          strand="+-"[i % 2], seq="ACGT" * 50))
    reads = _reads(6, 3000, 5000, 1) + _reads(24, 3000, 5000, 0) + _reads(2, 3000, 5000, 2) + _reads(12, 11000, 17500, 2) + \
        _reads(1, 11000, 17500, 1)
    return mk_task(leads, reads, 20000)


def case_consensus_quirks():
    """novel_from_reads quirks: j==0 first anchor, unequal advances, short reads, N/lowercase,
    repeated k-mers (taboo), reads longer/shorter than best, exactly 4 others (threshold)."""
    rng = np.random.default_rng(17)
    allele = _rng_seq(rng, 420)
    rep = ("ACGTTG" * 40)[:230]
    leads = []
    def add(pos, seq, name, svlen=None):
        leads.append(dict(svtype="INS", ref_start=pos, svlen=len(seq) if svlen is None else svlen, read=name, seq=seq,
                          strand="+-"[len(leads) % 2]))
    # site A: indels in the others -> unequal advances
    add(2000, allele, "a0")
    add(2001, allele[:100] + "GG" + allele[100:], "a1")
    add(2002, allele[:200] + allele[203:], "a2")
    add(2003, _mutate(rng, allele, 0.1), "a3")
    add(2004, allele[3:], "a4")
    add(2005, "TT" + allele, "a5")
    add(2006, allele[:50].lower() + allele[50:300] + "N" * 5 + allele[305:], "a6")
    add(2007, "ACG", "a7", svlen=420)          # shorter than k
    # site B: low complexity -> taboo k-mers, exactly 4 others
    for i in range(5):
        add(6000 + i, _mutate(rng, rep, 0.02), f"b{i}")
    # site C: 3 others only -> no consensus, best lead verbatim
    for i in range(4):
        add(9000 + i, _mutate(rng, allele[:130], 0.08), f"c{i}")
    # site D: systematic error in the best read corrected by voters
    base = _rng_seq(rng, 360)
    wrong = base[:180] + ("A" if base[180] != "A" else "C") + base[181:]
    add(12000, wrong, "d0")
    for i in range(7):
        add(12001 + i % 3, base, f"d{i + 1}")
    return mk_task(leads, _reads(30, 0, 20000), 20000)


def case_long_del_dup_cov():
    """COV_CHANGE_DEL / COV_CHANGE_DUP / COV_MIN / SVLEN_MIN branches on >= 50 kb events."""
    leads = []
    for i in range(6):
        leads.append(dict(svtype="DEL", ref_start=160000 + i % 2, svlen=-60000, read=f"l{i}", strand="+-"[i % 2], source="SPLIT_SUP"))
    for i in range(6):
        leads.append(dict(svtype="DUP", ref_start=300000 + i % 2, svlen=70000, read=f"m{i}", strand="+-"[i % 2], source="SPLIT_SUP"))
    for i in range(12):
        leads.append(dict(svtype="DEL", ref_start=500000 + i % 3, svlen=-47 - (i % 4), read=f"n{i}", strand="+-"[i % 2]))
    for i in range(5):
        leads.append(dict(svtype="INV", ref_start=650000 + i % 2, svlen=15000, read=f"o{i}", strand="+-"[i % 2], source="SPLIT_SUP"))
    reads = _reads(30, 0, 125000) + _reads(12, 90000, 200000) + _reads(20, 150000, 420000) + _reads(25, 330000, 360000) + \
        _reads(20, 400000, 640000) + _reads(9, 670000, 700000)
    return mk_task(leads, reads, 700000)


def case_tr_sweep():
    """Tandem-repeat sweep semantics (cluster.py:240-246) with nested/overlapping TRs and repeat merging."""
    leads = []
    k = 0
    for b, n, sl in [(3050, 3, 300), (3450, 3, 310), (4250, 2, 100), (5050, 3, 90), (5650, 3, 95), (7050, 4, 80), (9050, 2, 60)]:
        for j in range(n):
            leads.append(dict(svtype="INS", ref_start=b + j * 9, svlen=sl + j, read=f"t{k}", seq="ACGTTGCA" * ((sl + j) // 8) + "A" * ((sl + j) % 8),
                              strand="+-"[k % 2]))
            k += 1
    trs = [(2000, 8000), (2500, 3100), (3000, 3600), (5000, 5200), (6900, 7200), (9000, 9040)]
    return mk_task(leads, _reads(20, 0, 12000), 12000, trs=trs)


def case_empty():
    return mk_task([], _reads(3, 0, 5000), 5000)


def case_single_leads_only():
    """bins with a single lead never seed a cluster (cluster.py:262) -> no candidates."""
    leads = [dict(svtype="DEL", ref_start=1000 + 300 * i, svlen=-100, read=f"s{i}") for i in range(8)]
    return mk_task(leads, _reads(10, 0, 5000), 5000)


HAND = {
    "resplit_wrap": (case_resplit_wrap, {}, ()),
    "merge_index_rule": (case_merge_index_rule, {}, ()),
    "bnd_first_error": (case_bnd_first_error, {}, ()),
    "bnd_stale_end": (case_bnd_stale_end, {}, ()),
    "merge_inner": (case_merge_inner, {}, ()),
    "long_ins": (case_long_ins, {}, ()),
    "gt_failed_edges": (case_gt_failed_and_edges, {}, ()),
    "compute_metrics_big": (case_compute_metrics_big, {}, ()),
    "phase_rescue": (case_phase_rescue, {}, ()),
    "consensus_quirks": (case_consensus_quirks, {}, ()),
    "long_del_dup_cov": (case_long_del_dup_cov, {}, ()),
    "long_del_dup_cov_mosaic": (case_long_del_dup_cov, dict(mosaic=True), ("--mosaic",)),
    "tr_sweep": (case_tr_sweep, {}, ()),
    "tr_sweep_repeat": (case_tr_sweep, dict(repeat=True), ("--repeat",)),
    "empty": (case_empty, {}, ()),
    "single_leads_only": (case_single_leads_only, {}, ()),
    "single_leads_noqc": (case_single_leads_only, dict(no_qc=True), ("--no-qc",)),
    "phase_off_symbolic": (case_consensus_quirks, dict(symbolic=True), ("--symbolic",)),
    "no_consensus": (case_consensus_quirks, dict(no_consensus=True), ("--no-consensus",)),
    # --dev-no-resplit keeps BND clusters with mixed mate contigs whole: SUPPORT counts the reads of the selected mate contig,
    # RNAMES the reads of the whole cluster (sv.py:555 vs 636)
    "bnd_mixed_mates_no_resplit": (case_bnd_stale_end, dict(dev_no_resplit=True), ("--dev-no-resplit",)),
}

# seeded synthetic (SURVEY.md 8d shapes, shrunk contigs) --------------------------------------------
SYNTH = {
    "chr20_30x_ont": (lambda: synth.gen_task(0, "chr20", 3_000_000, 30, 1), {}, ()),
    "chr21_30x_mosaic": (lambda: synth.gen_task(1, "chr21", 2_500_000, 30, 3, mosaic_frac=0.3), dict(mosaic=True), ("--mosaic",)),
    "chr22_60x_hifi": (lambda: synth.gen_task(2, "chr22", 2_000_000, 60, 2, err=0.005), {}, ()),
    "chr19_30x_no_tr": (lambda: _no_tr(synth.gen_task(3, "chr19", 2_000_000, 30, 5)), {}, ()),
    "chr18_20x_auto_nm": (lambda: synth.gen_task(4, "chr18", 2_000_000, 20, 6), dict(minsupport="auto", qc_nm=True),
                          ("--minsupport", "auto", "--qc-nm")),
    "chr17_15x_noqc": (lambda: synth.gen_task(5, "chr17", 1_500_000, 15, 7), dict(no_qc=True), ("--no-qc",)),
    # one reference-run fixture per BASELINE.json config at FULL contig size (chr21, 46.7 Mb; bench.py's own generator
    # settings for configs[1] / [2] / [3]), so the oracle is pinned at the scale the bench runs it at
    "chr21_full_30x_ont": (lambda: synth.gen_task(20, "chr21", synth.GRCH38["chr21"], 30, 1), {}, ()),
    "chr21_full_60x_hifi": (lambda: synth.gen_task(20, "chr21", synth.GRCH38["chr21"], 60, 1, err=0.005, read_len_mean=15000.0), {}, ()),
    "chr21_full_30x_mosaic": (lambda: synth.gen_task(20, "chr21", synth.GRCH38["chr21"], 30, 1, mosaic_frac=0.3), dict(mosaic=True), ("--mosaic",)),
}

_FUZZ_CFG = [
    ({}, ()),
    (dict(mosaic=True), ("--mosaic",)),
    (dict(minsupport="auto", qc_nm=True), ("--minsupport", "auto", "--qc-nm")),
    (dict(no_qc=True), ("--no-qc",)),
    (dict(qc_strand=True, minsvlen="50", cluster_merge_pos=50), ("--qc-strand", "True", "--minsvlen", "50", "--cluster-merge-pos", "50")),
    (dict(repeat=True, mosaic=True, mosaic_include_germline=True), ("--repeat", "--mosaic", "--mosaic-include-germline")),
]
FUZZ = {f"fuzz_{s}_{ci}": ((lambda s=s: synth.gen_fuzz(s, n_leads=400 + 90 * s, contig_len=30000 + 4000 * s)), kw, args)
        for s in range(12) for ci, (kw, args) in enumerate(_FUZZ_CFG) if (s + ci) % 3 == 0}


def _no_tr(ti):
    ti.tr_start = None
    ti.tr_end = None
    return ti


# the developer switches of the refinement stage on adversarial tasks (no resplit at all / none for the length bins only)
FUZZ_DEV = {
    "fuzz_5_no_resplit": (lambda: synth.gen_fuzz(5000, task_id=0), dict(dev_no_resplit=True), ("--dev-no-resplit",)),
    "fuzz_7_no_resplit_mosaic": (lambda: synth.gen_fuzz(5051, task_id=3), dict(dev_no_resplit=True, mosaic=True), ("--dev-no-resplit", "--mosaic")),
    "fuzz_9_no_resplit_repeat": (lambda: synth.gen_fuzz(5017, task_id=1), dict(dev_no_resplit_repeat=True), ("--dev-no-resplit-repeat",)),
}

def case_nmask_cov():
    """LeadProvider._mask_N_coverage (leadprov.py:420-443): runs of 'N' in the reference zero the coverage vector - under a
    DEL's centre sample, around an INS, across the start of the contig (the negative-index wrap of `cv[start-100]` lands in a
    masked stretch at the contig's end) and over a whole call; coverage.mean() drops accordingly."""
    rng = np.random.default_rng(17)
    leads = []
    for i in range(9):
        leads.append(dict(svtype="DEL", ref_start=20_600 + i % 4, svlen=-1200, read=f"d{i}", strand="+-"[i % 2]))
    allele = _rng_seq(rng, 220)
    for i in range(8):
        leads.append(dict(svtype="INS", ref_start=60_010 + i % 3, svlen=220, seq=_mutate(rng, allele, 0.03), read=f"i{i}", strand="+-"[i % 2]))
    for i in range(7):
        leads.append(dict(svtype="DEL", ref_start=90 + i % 2, svlen=-60, read=f"e{i}", strand="+-"[i % 2]))       # start - 100 < 0
    for i in range(6):
        leads.append(dict(svtype="DUP", ref_start=100_000 + i % 3, svlen=3000, read=f"u{i}", strand="+-"[i % 2], source="SPLIT_SUP"))
    ti = mk_task(leads, _reads(24, 0, 70_000) + _reads(11, 30_000, 139_000) + _reads(9, 95_000, 140_000, 1), 140_000)
    ti.nmask_start = np.array([19_900, 59_850, 60_090, 99_000, 139_700], np.int32)
    ti.nmask_end = np.array([20_050, 59_950, 60_300, 104_500, 140_000], np.int32)
    return ti


def case_deep_wrap_cov():
    """The coverage vector is uint16 (leadprov.py:451): 65 536 + k reads over a position read as k - in the five samples of a
    call, in coverage.mean() and (test_snf.py) in the per-bin means of the SNF writer.  A 3-kb amplicon at 66 000x next to
    an ordinary stretch."""
    leads = []
    for i in range(12):
        leads.append(dict(svtype="DEL", ref_start=1500 + i % 3, svlen=-300, read=f"a{i}", strand="+-"[i % 2]))
    for i in range(8):
        leads.append(dict(svtype="DEL", ref_start=6200 + i % 2, svlen=-80, read=f"b{i}", strand="+-"[i % 2]))
    reads = _reads(66_000, 400, 2_900) + _reads(200, 1_000, 2_000, 1) + _reads(30, 3_000, 9_000) + _reads(65_535, 2_950, 2_960)
    return mk_task(leads, reads, 10_000)


HAND_COV = {
    "nmask_cov": (case_nmask_cov, {}, ()),
    "deep_wrap_cov": (case_deep_wrap_cov, {}, ()),
}
ALL = {**HAND, **SYNTH, **FUZZ, **FUZZ_DEV, **HAND_COV}


# multi-sample combine (BASELINE.json configs[4] shape, shrunk): samples share sites, not reads ------------------
def population(n_samples, contig="chr21", length=1_500_000, cov=20, site_seed=501, task_id=0):
    return [synth.gen_task(task_id, contig, length, cov, seed=100 + s, site_seed=site_seed) for s in range(n_samples)]


COMBINE = {
    "combine_4samples": (lambda: population(4), ()),
    "combine_7samples_lowcov": (lambda: population(7, contig="chr22", length=1_000_000, cov=12, site_seed=777, task_id=3), ()),
    "combine_3samples_no_align": (lambda: population(3, length=1_000_000, site_seed=9), ("--combine-pctseq", "0")),
}


# the reference's whole CombineTask.execute (block / bin / flush-window driver) on in-memory SNF blocks
COMBINE_TASK = {
    "combine_task_6samples": (lambda: population(6, contig="chr21", length=3_000_000, cov=20, site_seed=4242, task_id=2), ()),
    "combine_task_3samples_lowcov": (lambda: population(3, contig="chr22", length=1_500_000, cov=10, site_seed=99, task_id=5), ()),
    # dense sites: several flush windows per 100-kb block, kept groups between windows and across blocks
    "combine_task_8samples_dense": (lambda: [synth.gen_task(4, "chr20", 600_000, 15, seed=300 + s, site_seed=31337, site_density=40 * 27000 / 3.1e9)
                                             for s in range(8)], ()),
    # the decision rules of SVGroup.call away from their defaults (confidence thresholds, NULL coverage, pair relabelling, filtered output)
    "combine_task_10samples_options": (lambda: [synth.gen_task(1, "chr19", 1_200_000, 15, seed=400 + s, site_seed=606, site_density=30 * 27000 / 3.1e9)
                                                for s in range(10)],
                                       ("--combine-high-confidence", "0.3", "--combine-low-confidence", "0.4", "--combine-low-confidence-abs", "3",
                                        "--combine-null-min-coverage", "8", "--combine-pair-relabel", "--combine-pair-relabel-threshold", "15",
                                        "--combine-output-filtered")),
    # medians instead of the first candidate's coordinates, a higher support threshold and minimum length
    "combine_task_5samples_medians": (lambda: [synth.gen_task(6, "chr18", 800_000, 25, seed=500 + s, site_seed=77, site_density=30 * 27000 / 3.1e9)
                                               for s in range(5)],
                                      ("--dev-combine-medians", "--combine-support-threshold", "4", "--minsvlen", "80")),
}

# real .snf files written by the reference (one per sample; per-sample reference command line); the samples are those of
# COMBINE_TASK["combine_task_3samples_lowcov"], so the combined calls of that golden also pin the merge over the files
SNF_FILES = {
    "snf_3samples_lowcov": (COMBINE_TASK["combine_task_3samples_lowcov"][0], ((), ("--output-rnames",), ())),
}


# ---------------------------------------------------------------------------------------------- signature extraction
# name -> dict(gen=kwargs of synth_bam.gen_records | fixture=file under tests/golden, contig, region, read_id_offset,
#              args=reference command line, overrides=attributes set on the reference config afterwards,
#              cfg=the same settings as sniffles_amd / oracle extraction config keywords)
EXTRACT = {
    "extract_fuzz_a": dict(gen=dict(seed=11, n_reads=400, sa_frac=0.4), contig="chrA", region=(0, 400000),
                           read_id_offset=0, args=(), overrides={}, cfg={}),
    "extract_fuzz_window": dict(gen=dict(seed=12, n_reads=400, sa_frac=0.5), contig="chrA", region=(60000, 310000),
                                read_id_offset=5000, args=(), overrides={}, cfg={}),
    "extract_no_tags": dict(gen=dict(seed=13, n_reads=250, with_tags=False), contig="chrA", region=(0, 400000),
                            read_id_offset=0, args=(), overrides={}, cfg={}),
    "extract_lowq_short": dict(gen=dict(seed=14, n_reads=400, sa_frac=0.5, read_len_mean=1500), contig="chrA",
                               region=(0, 400000), read_id_offset=7,
                               args=("--mapq", "0", "--min-alignment-length", "100", "--minsvlen", "30",
                                     "--long-ins-length", "1000", "--max-splits-base", "1", "--max-splits-kb", "0.7",
                                     "--dev-keep-lowqual-splits", "--exclude-flags", "1536", "--dev-seq-cache-maxlen", "400"),
                               overrides={},
                               cfg=dict(mapq=0, min_alignment_length=100, minsvlen_screen=27, long_ins_length=1000,
                                        max_splits_base=1, max_splits_kb=0.7, dev_keep_lowqual_splits=True,
                                        exclude_flags=1536, dev_seq_cache_maxlen=400)),
    "extract_no_advanced_tags": dict(gen=dict(seed=15, n_reads=300, sa_frac=0.4), contig="chrA", region=(0, 400000),
                                     read_id_offset=0, args=("--detect-large-ins", "False"), overrides=dict(phase=False),
                                     cfg=dict(advanced_tags=False, detect_large_ins=False)),
    "extract_other_contig": dict(gen=dict(seed=16, n_reads=300, sa_frac=0.5, contig_index=2), contig="chr2",
                                 region=(0, 300000), read_id_offset=0, args=(), overrides={}, cfg={}),
    "extract_ont_long": dict(gen=dict(seed=17, n_reads=60, style="ont", read_len_mean=20000, sa_frac=0.3), contig="chrA",
                             region=(0, 400000), read_id_offset=0, args=(), overrides={}, cfg={}),
    "extract_hg008_chr1": dict(fixture="bam_hg008.bam.gz", contig="chr1", region=(0, 248956422), read_id_offset=0,
                               args=(), overrides={}, cfg={}),
    "extract_hg008_chr18": dict(fixture="bam_hg008.bam.gz", contig="chr18", region=(0, 80373285), read_id_offset=0,
                                args=(), overrides={}, cfg={}),
    "extract_hg008_chrX": dict(fixture="bam_hg008.bam.gz", contig="chrX", region=(0, 156040895), read_id_offset=0,
                               args=(), overrides={}, cfg={}),
    "extract_hg002_chr1": dict(fixture="bam_hg002.bam.gz", contig="chr1", region=(72000000, 73000000), read_id_offset=0,
                               args=(), overrides={}, cfg={}),
}


def extract_records(case):
    """The record table of an extraction case (sniffles_amd.bam.BamRecords)."""
    import gzip
    import os
    from sniffles_amd import bam, synth_bam
    if "fixture" in case:
        with gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", case["fixture"]), "rb") as f:
            return bam.parse_bam(f.read())
    names, lens, recs = synth_bam.gen_records(**case["gen"])
    return bam.records_from_list(names, lens, recs)


# ---------------------------------------------------------------------------------------------- whole samples (BAM -> VCF / SNF)
def _sample(seed, **kw):
    from sniffles_amd import bam, synth_bam
    out = synth_bam.gen_sample(seed, **kw)
    recs = bam.records_from_list(out[0], out[1], out[2])
    recs.tandem_repeats = out[3] if len(out) > 3 else None      # {contig: [(start, end)]}: handed to both pipelines
    return recs


SAMPLES = {
    # two contigs above the 1-Mb cut of should_process_contig and one below it (skipped by the reference)
    "sample_two_contigs_12x": (lambda: _sample(5, ref_lens=(1_200_000, 16_000, 1_050_000), cov=12.0), ()),
    "sample_mosaic_20x": (lambda: _sample(6, ref_names=("chr7",), ref_lens=(1_000_001,), cov=20.0), ("--mosaic",)),
    # split alignments: large deletions, tandem duplications, inversion breakpoints, translocations between the contigs
    # tandem-repeat annotation (the `repeat` merge rules of cluster.resolve) and every contig processed (--all-contigs)
    "sample_tandem_repeats_15x": (lambda: _sample(8, ref_names=("chr9", "chrUn_small"), ref_lens=(1_100_000, 120_000), cov=15.0,
                                                  tr_frac=0.35, site_spacing=9000), ("--all-contigs",)),
    "sample_splits_14x": (lambda: _sample(7, ref_lens=(1_200_000, 16_000, 1_050_000), cov=14.0, split_spacing=90000), ()),
}

# run through the host emulation only (the GPU tier keeps to SAMPLES, whose device runs have been checked on the MI355X)
SAMPLES_EMU = {
    "sample_hifi_40x": (lambda: _sample(9, ref_names=("chr3",), ref_lens=(1_000_500,), cov=40.0, read_len_mean=15000, err=0.005), ()),
    "sample_noqc_10x": (lambda: _sample(10, ref_names=("chr4", "chr5"), ref_lens=(1_000_100, 1_000_200), cov=10.0, split_spacing=150000),
                        ("--no-qc",)),
    "sample_qcnm_auto_18x": (lambda: _sample(11, ref_names=("chr6",), ref_lens=(1_050_000,), cov=18.0, tr_frac=0.2),
                             ("--qc-nm", "--minsupport", "auto")),
}


# populations: several samples sharing SV sites (site_seed), each BAM -> .snf, then the multi-sample merge (config 5 shape)
POPULATIONS = {
    "population_4samples_12x": (lambda: [_sample(30 + s, ref_names=("chr8", "chr9"), ref_lens=(1_000_000, 1_000_050), cov=12.0,
                                                 site_seed=77, site_spacing=12000) for s in range(4)], ()),
    "population_6samples_mixed": (lambda: [_sample(50 + s, ref_names=("chr11",), ref_lens=(1_000_000,), cov=[8.0, 15.0, 25.0][s % 3],
                                                   site_seed=78, site_spacing=9000, split_spacing=120000) for s in range(6)], ()),
}
