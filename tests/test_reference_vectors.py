"""The reference's OWN known-answer vectors for this path: `src/tests/test_bnd.py`
  * TestBND.test_resolve_bnd (test_bnd.py:85-136): six BND ALT strings / mate fields (VCF 4.2, chapter 5.4),
  * TestBNDCusterSplit.test_SingleCluster (test_bnd.py:139-196): resplit_bnd keeps a 2-lead cluster intact.
At the reference's HEAD these tests fail on a stale `Cluster(...)` constructor; the vectors themselves are the spec.
Here they run through the whole path: two identical BND leads -> one cluster -> `call_from` + `resolve_bnd`."""
import pytest

import cases
from sniffles_amd import lib, records
from sniffles_amd.config import SnifflesConfig

# (contig, pos, mate_contig, mate_pos, is_first, is_reverse, expected ALT)   test_bnd.py:90-127
BND_VECTORS = [
    ("chr2", 321681, "chr17", 198982, True, True, "N]chr17:198982]"),
    ("chr2", 321682, "chr13", 123456, False, True, "]chr13:123456]N"),
    ("chr13", 123456, "chr2", 321682, True, False, "N[chr2:321682["),
    ("chr13", 123457, "chr17", 198983, False, False, "[chr17:198983[N"),
    ("chr17", 198982, "chr2", 321681, True, True, "N]chr2:321681]"),
    ("chr17", 198983, "chr13", 123457, False, False, "[chr13:123457[N"),
]


def bnd_task(contig, pos, mate_contig, mate_pos, is_first, is_reverse):
    leads = [dict(svtype="BND", ref_start=pos, svlen=0, read=f"read{k}", qry_start=1000 * k, qry_end=1000 * k, strand="+",
                  mapq=60, nm=100, mate=(mate_contig, mate_pos, is_first, is_reverse)) for k in (1, 2)]
    # a task whose only candidates are BNDs raises UnboundLocalError in the reference (postprocessing.py:84-106, preserved
    # here), so a plain deletion precedes the breakend, as in any real contig
    leads += [dict(svtype="DEL", ref_start=5000 + k, svlen=-300, read=f"del{k}", strand="+", mapq=60) for k in range(4)]
    reads = [(0, pos + 5000, 0)] * 12
    return cases.mk_task(leads, reads, pos + 100_000, contig=contig)


def run(ti, _lib=None):
    with lib.Batch(SnifflesConfig(), [ti]) as b:
        b.call_candidates(); b.finalize()
        return records.records(b.fetch(1), [ti], "final")[0]


def check(_lib):
    for contig, pos, mc, mp, first, rev, alt in BND_VECTORS:
        recs = run(bnd_task(contig, pos, mc, mp, first, rev), _lib)
        assert [x["svtype"] for x in recs] == ["DEL", "BND"]
        r = recs[1]
        assert (r["svtype"], r["pos"], r["alt"]) == ("BND", pos, alt)
        assert r["bnd"][:2] == [mc, mp] and r["bnd"][2:] == [first, rev]      # INFO CHR2 / mate position / orientation
    # test_SingleCluster: two exact leads with the same mate stay one cluster of two leads
    r = run(bnd_task("chr1", 10_000, "chr2", 20_000, True, False), _lib)
    assert len(r) == 2 and r[1]["support"] == 2 and r[1]["bnd"] == ["chr2", 20_000, True, False]


def test_reference_bnd_vectors_emulated():
    import emu.emu as E
    check(E.lib())


@pytest.mark.gpu
def test_reference_bnd_vectors_gpu():
    check(None)
