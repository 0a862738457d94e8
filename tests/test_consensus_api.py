"""Seam B4: sniffles_amd.consensus.novel_from_reads against vectors produced by the UNMODIFIED reference
function (tests/golden/consensus_novel_from_reads.json.gz, oracle/make_golden.py::main_consensus)."""
import pytest

import golden_util as gu


def load():
    return gu.load("consensus_novel_from_reads")["problems"]


def test_golden_vectors_are_meaningful():
    probs = load()
    assert len(probs) >= 100
    assert sum(p["expected"] != p["best"] for p in probs) >= 20          # the consensus really edits the best read
    assert all(len(p["expected"]) == len(p["best"]) for p in probs)      # substitutions only (consensus.py:365-380)


@pytest.mark.gpu
def test_novel_from_reads_batch_matches_reference():
    from sniffles_amd import consensus
    probs = load()
    got = consensus.novel_from_reads_batch([(p["best"], p["others"], p["skip"]) for p in probs], klen=6)
    bad = [i for i, (g, p) in enumerate(zip(got, probs)) if g != p["expected"]]
    assert bad == []


@pytest.mark.gpu
def test_long_copied_segments_match_reference():
    """Segments of hundreds to thousands of bases between two anchors (processed by the whole wave), shift excursions, odd
    characters inside them: tests/golden/consensus_long_segments.json.gz, from the unmodified reference function."""
    from sniffles_amd import consensus
    probs = gu.load("consensus_long_segments")["problems"]
    got = consensus.novel_from_reads_batch([(p["best"], p["others"], p["skip"]) for p in probs], klen=6)
    assert [i for i, (g, p) in enumerate(zip(got, probs)) if g != p["expected"]] == []
    assert sum(p["expected"] != p["best"] for p in probs) >= 20


@pytest.mark.gpu
def test_novel_from_reads_signature_and_errors():
    from sniffles_amd import consensus, lib

    class Lead:
        def __init__(self, seq):
            self.seq = seq

    p = load()[3]
    out = consensus.novel_from_reads(Lead(p["best"]), [Lead(o) for o in p["others"]], klen=p["klen"], skip=p["skip"],
                                     skip_repetitive=p["skip"])
    assert out == p["expected"]
    with pytest.raises(ValueError):
        consensus.novel_from_reads(Lead("ACGT"), [], klen=6, skip=3, skip_repetitive=4)
    with pytest.raises(lib.SnifflesAmdError):      # '-' is the reference's gap symbol
        consensus.novel_from_reads(Lead("ACGTACGTAC-T"), [Lead("ACGTACGTACGT")], klen=6, skip=3, skip_repetitive=3)
    with pytest.raises(lib.SnifflesAmdError):      # beyond the workgroup kernels' limits: refused, never approximated
        consensus.novel_from_reads(Lead("ACGT" * 20000), [Lead("ACGT" * 20000)], klen=6, skip=3, skip_repetitive=3)
