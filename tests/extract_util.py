"""Helpers of the extraction parity tests: canonical lead rows and the quantities the reference derives from reads."""
import hashlib

import numpy as np

from sniffles_amd.soa import SVTYPES, SOURCES, SVLEN_NONE


def records_sha(recs) -> str:
    h = hashlib.sha256()
    h.update(recs.blob.tobytes())
    h.update(recs.rec_off.tobytes())
    h.update(repr((recs.ref_names, recs.ref_lens)).encode())
    return h.hexdigest()


def canon_leads(ti) -> list:
    """TaskInput -> rows in the field order of oracle/ref_harness.lead_record (golden "leads")."""
    L = ti.leads
    pool = ti.seq_pool.tobytes()
    rows = []
    for i in range(ti.n_leads):
        svt = SVTYPES[L["svtype"][i]]
        sl, so, svlen, nm = int(L["seq_len"][i]), int(L["seq_off"][i]), int(L["svlen"][i]), float(L["nm"][i])
        bnd = None
        if svt == "BND":
            bnd = [ti.contig_name(int(L["mate_contig"][i])), int(L["mate_ref_start"][i]), bool(L["bnd_is_first"][i]),
                   bool(L["bnd_is_reverse"][i])]
        rows.append([int(L["read_id"][i]), ti.qname(int(L["qname_id"][i])), ti.contig, int(L["ref_start"][i]),
                     int(L["ref_end"][i]), int(L["qry_start"][i]), int(L["qry_end"][i]), "-" if L["strand"][i] else "+",
                     int(L["mapq"][i]), None if nm != nm else nm.hex(), SOURCES[L["source"][i]], svt,
                     None if svlen == int(SVLEN_NONE) else svlen, None if sl < 0 else pool[so:so + sl].decode("latin-1"),
                     str(int(L["hap"][i])), ti.ps_name(int(L["ps_rank"][i])), bool(L["is_sa"][i]), int(L["read_len"][i]), bnd])
    return rows


def coverage_diff(reads, contig_len):
    """Sparse difference array of `coverage[s:e] += 1` over all reads (leadprov.py:510)."""
    d = np.zeros(contig_len + 1, np.int64)
    for s, e, _ in reads:
        s, e = max(0, min(s, contig_len)), max(0, min(e, contig_len))
        if e > s:
            d[s] += 1
            d[e] -= 1
    d = d[:contig_len]
    nz = np.nonzero(d)[0]
    return nz.tolist(), d[nz].tolist()


def hapref(reads, binsize=100):
    """leadhapcount["REF"] after record_hap_ref for every read (leadprov.py:387-398, 567-571)."""
    tab = {}
    for s, e, hp in reads:
        for b in range(int(s / binsize) * binsize, int(e / binsize) * binsize, binsize):
            tab.setdefault(b, [0, 0, 0])[hp] += 1
    return sorted([b] + c for b, c in tab.items())


def check_against_golden(exp, rows, reads, qc_nm_threshold_hex, read_id, contig_len):
    assert rows == exp["leads"]
    assert read_id == exp["read_id"] and len(reads) == exp["read_count"]
    assert qc_nm_threshold_hex == exp["qc_nm_threshold"]
    assert contig_len == exp["contig_len"]
    pos, delta = coverage_diff(reads, contig_len)
    assert pos == exp["cov_pos"] and delta == exp["cov_delta"]
    assert hapref(reads) == exp["hapref"]
