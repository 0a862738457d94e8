"""Seeded synthetic HG002-shaped SV-signature sets (SURVEY.md §8d recipe).

No BAM exists in the build/bench environment and pysam is absent, so the
benchmark inputs are generated at the boundary where the hot path starts:
per-contig lead tables in arrival (BAM) order plus read intervals.  Everything
is vectorised numpy so a 30x whole-genome set (~3 M leads, ~0.2 GB of INS
sequence) is produced in tens of seconds.

Shapes follow the reference extraction code so every field the hot path reads
is populated the way `leadprov.read_iterindels` (`leadprov.py:583-670`),
`read_itersplits` (`:227-355`) and `Lead.for_bnd` (`:57-132`) would populate it.
"""
from __future__ import annotations

import numpy as np

from .soa import (TaskInput, SVT, SRC, SVLEN_NONE, SEQ_NONE, PS_NONE, empty_leads, concat_leads)

# GRCh38 primary assembly lengths
GRCH38 = {
    "chr1": 248956422, "chr2": 242193529, "chr3": 198295559, "chr4": 190214555, "chr5": 181538259,
    "chr6": 170805979, "chr7": 159345973, "chr8": 145138636, "chr9": 138394717, "chr10": 133797422,
    "chr11": 135086622, "chr12": 133275309, "chr13": 114364328, "chr14": 107043718, "chr15": 101991189,
    "chr16": 90338345, "chr17": 83257441, "chr18": 80373285, "chr19": 58617616, "chr20": 64444167,
    "chr21": 46709983, "chr22": 50818468, "chrX": 156040895, "chrY": 57227415,
}
CONTIGS = list(GRCH38)
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def _ranges(counts: np.ndarray) -> tuple:
    """For counts c_i return (owner index, position within owner) of length sum(c)."""
    counts = counts.astype(np.int64)
    total = int(counts.sum())
    owner = np.repeat(np.arange(counts.shape[0], dtype=np.int64), counts)
    first = np.cumsum(counts) - counts
    within = np.arange(total, dtype=np.int64) - np.repeat(first, counts)
    return owner, within


def gen_task(task_id: int, contig: str, contig_len: int, coverage: float, seed: int,
             err: float = 0.04, mosaic_frac: float = 0.0, site_density: float = 27000 / 3.1e9,
             contig_names=None, read_len_mean: float = 20000.0, site_seed: int = None) -> TaskInput:
    """One contig's lead table.  `err` = per-base substitution error of INS read sequences
    (0.04 ONT-like, 0.005 HiFi-like); `mosaic_frac` = fraction of sites planted at VAF 0.05-0.2."""
    rng = np.random.default_rng([seed, task_id, 7919])
    L = int(contig_len)
    contig_names = sorted(contig_names or CONTIGS)
    ctg_rank = {c: i for i, c in enumerate(contig_names)}

    # ---------------- reads (sorted by start = BAM order) ----------------
    n_reads = max(8, int(coverage * L / read_len_mean))
    rlen = np.clip(rng.exponential(read_len_mean, n_reads), 2000, 200000).astype(np.int64)
    rstart = np.sort(rng.integers(0, max(1, L - 2000), n_reads))
    rend = np.minimum(rstart + rlen, L)
    rstrand = rng.integers(0, 2, n_reads).astype(np.uint8)
    rhp = rng.choice(np.array([0, 1, 2], np.uint8), n_reads, p=[0.3, 0.35, 0.35])
    # phase set: block of 1 Mbp when phased, "NULL" otherwise; rank order must equal str order
    ps_block = (rstart // 1_000_000) * 1_000_000 + 1
    ps_strs = sorted(set(str(int(b)) for b in np.unique(ps_block)) | {"NULL"})
    ps_rank_of = {s: i for i, s in enumerate(ps_strs)}
    blk_rank = np.array([ps_rank_of[str(int(b))] for b in np.unique(ps_block)], np.int32)
    rps = np.where(rhp > 0, blk_rank[np.searchsorted(np.unique(ps_block), ps_block)], ps_rank_of["NULL"]).astype(np.int32)
    rmapq = np.where(rng.random(n_reads) < 0.9, 60, rng.integers(20, 60, n_reads)).astype(np.uint8)
    rnm = np.clip(rng.normal(0.02, 0.006, n_reads), 0.0, None)
    ris_sa = (rng.random(n_reads) < 0.04).astype(np.uint8)
    ralen = (rend - rstart).astype(np.int64)

    # ---------------- SV sites ----------------
    # `site_seed`: several samples of one population share the sites (and alleles) but not the reads
    rng_reads = rng
    if site_seed is not None:
        rng = np.random.default_rng([site_seed, task_id, 104723])
    n_sites = max(1, int(round(site_density * L)))
    spos = np.sort(rng.integers(6000, max(6001, L - 60000), n_sites))
    u = rng.random(n_sites)
    stype = np.where(u < 0.47, SVT["INS"], np.where(u < 0.94, SVT["DEL"],
             np.where(u < 0.97, SVT["DUP"], np.where(u < 0.99, SVT["INV"], SVT["BND"])))).astype(np.uint8)
    v = rng.random(n_sites)
    slen = np.where(v < 0.7, 50 + rng.exponential(150, n_sites),
                    np.where(v < 0.9, rng.normal(320, 15, n_sites), rng.normal(6000, 100, n_sites)))
    slen = np.maximum(50, slen).astype(np.int64)
    shom = rng.random(n_sites) < 0.4
    shap = rng.integers(1, 3, n_sites).astype(np.uint8)
    str_like = rng.random(n_sites) < 0.4
    sjit = np.where(str_like, 40.0, 2.0)
    svaf = np.where(rng.random(n_sites) < mosaic_frac, rng.uniform(0.05, 0.2, n_sites), -1.0)
    smate_ctg = rng.integers(0, len(contig_names), n_sites).astype(np.int32)
    smate_pos = rng.integers(10000, 40_000_000, n_sites).astype(np.int32)
    sfirst = rng.integers(0, 2, n_sites).astype(np.uint8)
    srev = rng.integers(0, 2, n_sites).astype(np.uint8)

    tr_start = np.maximum(0, spos[str_like] - 500).astype(np.int32)
    tr_end = (spos[str_like] + 500).astype(np.int32)
    if site_seed is not None:  # alleles belong to the sites, everything below to the sample
        n_ins_bases = int(np.where(stype == SVT["INS"], slen, 0).sum())
        shared_alleles = ACGT[rng.integers(0, 4, n_ins_bases)]
        if mosaic_frac == 0.0:  # population sample: not every sample carries every site
            present = np.random.default_rng([seed, task_id, 31]).random(n_sites) < 0.7
            svaf = np.where(present, svaf, 0.0)
    else:
        shared_alleles = None
    rng = rng_reads

    # ---------------- site x covering-read pairs ----------------
    extent = np.where((stype == SVT["INS"]) | (stype == SVT["BND"]), 0, slen)
    lo = np.searchsorted(rstart, spos - 200000, side="left")
    hi = np.searchsorted(rstart, spos - 200, side="right")
    s_i, w = _ranges(hi - lo)
    r_i = lo[s_i] + w
    keep = rend[r_i] >= spos[s_i] + extent[s_i] + 200
    s_i, r_i = s_i[keep], r_i[keep]
    # carrier model: hom -> all; het -> matching haplotype, unphased 50 %; mosaic -> VAF
    pu = rng.random(s_i.shape[0])
    carr = np.where(svaf[s_i] >= 0, pu < svaf[s_i],
                    np.where(shom[s_i], True,
                             np.where(rhp[r_i] == 0, pu < 0.5, rhp[r_i] == shap[s_i])))
    s_i, r_i = s_i[carr], r_i[carr]
    m = s_i.shape[0]

    jpos = (spos[s_i] + np.rint(rng.normal(0, 1, m) * sjit[s_i])).astype(np.int64)
    jlen = np.maximum(45, np.rint(slen[s_i] * (1 + rng.normal(0, 0.015, m)))).astype(np.int64)
    t = stype[s_i]

    site = empty_leads(m)
    site["svtype"][:] = t
    is_ins, is_del, is_bnd = t == SVT["INS"], t == SVT["DEL"], t == SVT["BND"]
    is_split = (t == SVT["DUP"]) | (t == SVT["INV"])
    site["ref_start"][:] = np.where(is_del, jpos + jlen, jpos)
    site["ref_end"][:] = np.where(is_del, jpos, np.where(is_split, jpos + jlen, jpos))
    q0 = np.clip(jpos - rstart[r_i], 0, None)
    site["qry_start"][:] = q0
    site["qry_end"][:] = np.where(is_ins, q0 + jlen, q0)
    site["svlen"][:] = np.where(is_del, -jlen, np.where(is_bnd, 0, jlen))
    site["source"][:] = np.where(is_bnd, SRC["BND_SA"], np.where(is_split, SRC["SPLIT_SUP"], SRC["INLINE"]))
    site["mate_contig"][:] = np.where(is_bnd, smate_ctg[s_i], 0)
    site["mate_ref_start"][:] = np.where(is_bnd, smate_pos[s_i] + rng.integers(-3, 4, m), 0)
    site["bnd_is_first"][:] = np.where(is_bnd, sfirst[s_i], 0)
    site["bnd_is_reverse"][:] = np.where(is_bnd, srev[s_i], 0)
    # a minority of BND reads point somewhere else (exercises resplit_bnd / resolve_bnd majority)
    stray = is_bnd & (rng.random(m) < 0.1)
    site["mate_ref_start"][stray] += 50000
    # long INS: some reads only show a clip (svlen None, seq None; leadprov.py:639-653)
    clip = is_ins & (slen[s_i] >= 1250) & (rng.random(m) < 0.25)
    site["svlen"][clip] = SVLEN_NONE
    site_read = r_i
    # TR-like INS sites: 15 % of reads report the insertion as two nearby pieces (merge_inner food)
    two = is_ins & ~clip & str_like[s_i] & (rng.random(m) < 0.15) & (jlen >= 120)
    # INS sequences: site allele stretched to the read's length + substitution errors
    ins_mask = is_ins & ~clip
    ins_idx = np.nonzero(ins_mask)[0]
    a_len = slen
    a_off = np.cumsum(np.where(stype == SVT["INS"], a_len, 0)) - np.where(stype == SVT["INS"], a_len, 0)
    allele_pool = shared_alleles if shared_alleles is not None else \
        ACGT[rng.integers(0, 4, int(np.where(stype == SVT["INS"], a_len, 0).sum()))]
    out_len = jlen[ins_idx]
    o_i, p = _ranges(out_len)
    src_pos = (p * a_len[s_i[ins_idx]][o_i]) // np.maximum(1, out_len[o_i])
    seq_bytes = allele_pool[a_off[s_i[ins_idx]][o_i] + src_pos]
    e = rng.random(seq_bytes.shape[0]) < err
    seq_bytes = np.where(e, ACGT[rng.integers(0, 4, seq_bytes.shape[0])], seq_bytes).astype(np.uint8)
    seq_off = np.cumsum(out_len) - out_len
    site["seq_len"][ins_idx] = out_len
    site["seq_off"][ins_idx] = seq_off
    pool_parts = [seq_bytes]
    pool_size = int(out_len.sum())

    # split "two-piece" INS leads: piece A keeps [0,h), piece B is a new lead with [h,len) 60 bp downstream
    two_idx = np.nonzero(two)[0]
    extra = empty_leads(two_idx.shape[0])
    if two_idx.shape[0]:
        h = (jlen[two_idx] // 2).astype(np.int64)
        for name in site:
            extra[name][:] = site[name][two_idx]
        extra["ref_start"][:] = site["ref_start"][two_idx] + 60
        extra["ref_end"][:] = extra["ref_start"]
        extra["qry_start"][:] = site["qry_start"][two_idx] + h + 60
        extra["qry_end"][:] = extra["qry_start"] + (jlen[two_idx] - h)
        extra["svlen"][:] = jlen[two_idx] - h
        extra["seq_off"][:] = site["seq_off"][two_idx] + h
        extra["seq_len"][:] = jlen[two_idx] - h
        site["svlen"][two_idx] = h
        site["seq_len"][two_idx] = h
        site["qry_end"][two_idx] = site["qry_start"][two_idx] + h
    extra_read = site_read[two_idx]

    # ---------------- noise leads (0.3 / read), INS or DEL 45-89 bp ----------------
    n_noise = rng.poisson(0.3 * n_reads)
    nr = rng.integers(0, n_reads, n_noise)
    noff = (rng.random(n_noise) * np.maximum(1, ralen[nr] - 200)).astype(np.int64) + 100
    nlen = rng.integers(45, 90, n_noise).astype(np.int64)
    nins = rng.random(n_noise) < 0.5
    noise = empty_leads(n_noise)
    npos = rstart[nr] + noff
    noise["svtype"][:] = np.where(nins, SVT["INS"], SVT["DEL"])
    noise["ref_start"][:] = np.where(nins, npos, npos + nlen)
    noise["ref_end"][:] = npos
    noise["qry_start"][:] = noff
    noise["qry_end"][:] = np.where(nins, noff + nlen, noff)
    noise["svlen"][:] = np.where(nins, nlen, -nlen)
    noise["source"][:] = SRC["INLINE"]
    nl = np.where(nins, nlen, 0)
    noise["seq_len"][:] = np.where(nins, nlen, SEQ_NONE)
    noise["seq_off"][:] = pool_size + np.cumsum(nl) - nl
    pool_parts.append(ACGT[rng.integers(0, 4, int(nl.sum()))])
    pool_size += int(nl.sum())

    # ---------------- single-break leads (0.2 / read) ----------------
    n_sb = rng.poisson(0.2 * n_reads)
    sr = rng.integers(0, n_reads, n_sb)
    left = rng.random(n_sb) < 0.5
    sb = empty_leads(n_sb)
    sb["svtype"][:] = np.where(left, SVT["SINGLE_LEFT"], SVT["SINGLE_RIGHT"])
    sb["ref_start"][:] = np.where(left, rstart[sr], rend[sr])
    sb["ref_end"][:] = sb["ref_start"]
    cl = rng.integers(45, 1200, n_sb)
    sb["qry_start"][:] = np.where(left, 0, ralen[sr])
    sb["qry_end"][:] = sb["qry_start"] + cl
    sb["svlen"][:] = 0
    sb["source"][:] = SRC["INLINE"]

    # ---------------- assemble in BAM order: by read, then by ref position along the read ----------------
    parts = [site, extra, noise, sb]
    reads_of = [site_read, extra_read, nr, sr]
    leads = concat_leads(parts)
    lread = np.concatenate(reads_of).astype(np.int64)
    ref_lo = np.minimum(leads["ref_start"], leads["ref_end"]).astype(np.int64)
    order = np.lexsort((ref_lo, lread))
    for name in leads:
        leads[name] = np.ascontiguousarray(leads[name][order])
    lread = lread[order]
    # per-read attributes
    bnd = leads["svtype"] == SVT["BND"]
    leads["qname_id"][:] = lread.astype(np.uint32)
    leads["read_id"][:] = (lread + 1).astype(np.uint32)
    leads["strand"][:] = rstrand[lread]
    leads["mapq"][:] = rmapq[lread]
    leads["nm"][:] = rnm[lread]
    leads["read_len"][:] = np.where(bnd | (leads["source"] != SRC["INLINE"]), 0, ralen[lread])
    # Lead.for_bnd leaves hap="0", phase_set=None, is_sa=False (leadprov.py:113-131)
    leads["hap"][:] = np.where(bnd, 0, rhp[lread])
    leads["ps_rank"][:] = np.where(bnd, PS_NONE, rps[lread])
    leads["is_sa"][:] = np.where(bnd, 0, ris_sa[lread])

    ti = TaskInput(task_id=task_id, contig=contig, contig_len=L, leads=leads,
                   seq_pool=np.ascontiguousarray(np.concatenate(pool_parts)) if pool_size else np.zeros(0, np.uint8),
                   read_start=rstart.astype(np.int32), read_end=rend.astype(np.int32), read_hp=rhp,
                   tr_start=tr_start, tr_end=tr_end,
                   qc_nm_threshold=float(rnm.sum() / max(1, n_reads)),
                   qnames=None, ps_names=ps_strs, contig_names=contig_names)
    ti.validate()
    return ti


def gen_genome(coverage: float = 30.0, seed: int = 1, contigs=None, err: float = 0.04,
               mosaic_frac: float = 0.0, scale: float = 1.0) -> list:
    """Whole-genome (24 contigs) task list; `scale` shrinks every contig (tests)."""
    contigs = contigs or CONTIGS
    return [gen_task(i, c, max(200000, int(GRCH38[c] * scale)), coverage, seed, err=err,
                     mosaic_frac=mosaic_frac) for i, c in enumerate(contigs)]


def gen_fuzz(seed: int, task_id: int = 0, n_leads: int = None, contig_len: int = None) -> TaskInput:
    """Adversarial small task: dense hot-spots, shared reads, nested tandem repeats, zero-coverage
    zones, thresholds-straddling lengths, non-ACGT bases.  Exercises the quirks of SURVEY.md Appendix A
    (resplit wrap-around, merge index rule, stale BND end, seq cap, negative-index coverage)."""
    rng = np.random.default_rng([seed, 104729])
    L = int(contig_len or rng.integers(20_000, 120_000))
    n = int(n_leads or rng.integers(50, 2500))
    n_reads = int(rng.integers(30, 600))
    rstart = np.sort(rng.integers(0, L - 10, n_reads))
    rend = np.minimum(L + rng.integers(-5, 50, n_reads), rstart + rng.integers(200, L, n_reads))
    rend = np.maximum(rend, rstart + 1)
    if rng.random() < 0.5:  # carve a zero-coverage hole
        h0 = int(rng.integers(0, L)); h1 = h0 + int(rng.integers(100, 5000))
        kill = (rstart < h1) & (rend > h0)
        rend = np.where(kill, np.maximum(rstart + 1, np.minimum(rend, h0)), rend)
        rstart = np.where(kill & (rstart >= h0), np.minimum(L - 2, h1), rstart)
        o = np.argsort(rstart, kind="stable"); rstart, rend = rstart[o], np.maximum(rend[o], rstart[o] + 1)
    rhp = rng.integers(0, 3, n_reads).astype(np.uint8)
    n_hot = int(rng.integers(1, 12))
    hot = rng.integers(0, L, n_hot)
    hot_sd = rng.choice([1, 5, 30, 120, 400], n_hot)
    hot_len = rng.choice([48, 52, 60, 100, 300, 700, 2600, 3000], n_hot)
    hot_type = rng.choice([0, 0, 0, 1, 1, 1, 2, 3, 4, 5, 6], n_hot)
    which = rng.integers(0, n_hot, n)
    leads = empty_leads(n)
    uniform = rng.random(n) < 0.15
    pos = np.where(uniform, rng.integers(-50, L + 50, n), hot[which] + np.rint(rng.normal(0, 1, n) * hot_sd[which])).astype(np.int64)
    pos = np.clip(pos, -20, L + 20)
    t = np.where(uniform | (rng.random(n) < 0.1), rng.integers(0, 7, n), hot_type[which]).astype(np.uint8)
    ln = np.maximum(1, np.rint(hot_len[which] * (1 + rng.normal(0, 0.08, n)) + rng.integers(-25, 26, n))).astype(np.int64)
    ln = np.where(rng.random(n) < 0.1, rng.integers(30, 140, n), ln)
    is_ins, is_del, is_bnd = t == SVT["INS"], t == SVT["DEL"], t == SVT["BND"]
    single = t >= SVT["SINGLE_LEFT"]
    leads["svtype"][:] = t
    leads["ref_start"][:] = pos
    leads["ref_end"][:] = np.where(is_del, pos - ln, np.where(is_ins | is_bnd | single, pos, pos + ln))
    leads["svlen"][:] = np.where(is_del, -ln, np.where(is_bnd | single, 0, ln))
    long_clip = is_ins & (rng.random(n) < 0.12)
    leads["svlen"][long_clip] = SVLEN_NONE
    n_q = max(2, int(n * rng.choice([0.2, 0.5, 0.9])))
    q = rng.integers(0, n_q, n)
    leads["qname_id"][:] = q
    leads["read_id"][:] = q + 1 + (rng.random(n) < 0.1) * n_q  # supplementary record of the same read
    qstrand = rng.integers(0, 2, n_q).astype(np.uint8)
    leads["strand"][:] = np.where(rng.random(n) < 0.9, qstrand[q], 1 - qstrand[q])
    leads["mapq"][:] = rng.integers(0, 61, n)
    leads["nm"][:] = np.round(rng.uniform(0, 0.09, n_q), 4)[q]
    leads["source"][:] = np.where(is_bnd, SRC["BND_SA"], rng.choice([0, 0, 0, 1, 2], n))
    qhap = rng.integers(0, 3, n_q).astype(np.uint8)
    leads["hap"][:] = np.where(is_bnd, 0, qhap[q])
    ps_strs = sorted({"NULL", "1", "10001", "9", "250001"})
    ps_of_q = rng.integers(0, len(ps_strs), n_q).astype(np.int32)
    leads["ps_rank"][:] = np.where(is_bnd, PS_NONE, np.where(qhap[q] > 0, ps_of_q[q], ps_strs.index("NULL")))
    leads["is_sa"][:] = rng.random(n) < rng.choice([0.0, 0.1, 0.9])
    qs = rng.integers(0, 30000, n)
    leads["qry_start"][:] = qs
    leads["qry_end"][:] = np.where(is_ins, qs + ln, qs)
    leads["read_len"][:] = qs + rng.integers(0, 30000, n)
    same_read_near = rng.random(n) < 0.3  # make same-read neighbours plausible for merge_inner
    leads["qry_start"][same_read_near] = (pos[same_read_near] % 997) * 3
    leads["qry_end"][same_read_near] = leads["qry_start"][same_read_near] + np.where(is_ins[same_read_near], ln[same_read_near], 0)
    contig_names = sorted(["chr1", "chr10", "chr2", "chrX"])
    leads["mate_contig"][:] = np.where(is_bnd, rng.choice([0, 0, 0, 1, 2, 3], n), 0)
    leads["mate_ref_start"][:] = np.where(is_bnd, rng.choice([5000, 5400, 6100, 9000, 250000], n) + rng.integers(-600, 600, n), 0)
    leads["bnd_is_first"][:] = np.where(is_bnd, rng.random(n) < 0.7, 0)
    leads["bnd_is_reverse"][:] = np.where(is_bnd, rng.random(n) < 0.3, 0)
    # INS sequences: per hot-spot allele, stretched, with errors and a few odd characters
    alpha = np.frombuffer(b"ACGTNacgtR", dtype=np.uint8)
    has_seq = is_ins & ~long_clip & (rng.random(n) < 0.92)
    idx = np.nonzero(has_seq)[0]
    slen = np.where(rng.random(idx.shape[0]) < 0.85, ln[idx], np.maximum(1, ln[idx] + rng.integers(-30, 30, idx.shape[0])))
    a_len = np.maximum(8, hot_len)
    a_off = np.cumsum(a_len) - a_len
    allele = ACGT[rng.integers(0, 4, int(a_len.sum()))]
    if rng.random() < 0.3:  # low-complexity allele -> repeated k-mers (taboo anchors)
        allele[:] = np.tile(ACGT[rng.integers(0, 4, 7)], allele.shape[0] // 7 + 1)[:allele.shape[0]]
    o_i, p = _ranges(slen)
    w = which[idx][o_i]
    src = (p * a_len[w]) // np.maximum(1, slen[o_i])
    sb = allele[a_off[w] + src]
    e = rng.random(sb.shape[0]) < rng.choice([0.0, 0.03, 0.12])
    sb = np.where(e, ACGT[rng.integers(0, 4, sb.shape[0])], sb)
    odd = rng.random(sb.shape[0]) < 0.002
    sb = np.where(odd, alpha[rng.integers(0, alpha.shape[0], sb.shape[0])], sb).astype(np.uint8)
    leads["seq_len"][idx] = slen
    leads["seq_off"][idx] = np.cumsum(slen) - slen
    # arrival order: by a synthetic read order
    order = np.lexsort((pos, q))
    for name in leads:
        leads[name] = np.ascontiguousarray(leads[name][order])
    # tandem repeats: random, sorted by start, may nest / overlap
    tr_s = tr_e = None
    if rng.random() < 0.7:
        k = int(rng.integers(0, 8))
        s0 = np.sort(np.concatenate([hot[rng.integers(0, n_hot, k)] - rng.integers(0, 800, k), rng.integers(0, L, 2)]))
        tr_s = np.maximum(0, s0).astype(np.int32)
        tr_e = (tr_s + rng.integers(10, 2500, tr_s.shape[0])).astype(np.int32)
    ti = TaskInput(task_id=task_id, contig="chrF", contig_len=L, sv_id_start=int(rng.integers(0, 3)) * 17, leads=leads,
                   seq_pool=np.ascontiguousarray(sb), read_start=rstart.astype(np.int32), read_end=rend.astype(np.int32),
                   read_hp=rhp, tr_start=tr_s, tr_end=tr_e, qc_nm_threshold=float(rng.choice([0.02, 0.045])),
                   ps_names=ps_strs, contig_names=contig_names)
    ti.validate()
    return ti
