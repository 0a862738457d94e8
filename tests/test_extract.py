"""Signature extraction: raw BAM records -> task input, against the reference.

CPU tier: the oracle restatement (oracle/extract_oracle.py) and the kernel bodies (host emulation, thread form) against
goldens produced by the UNMODIFIED reference (`leadprov.build_leadtab` over oracle/pysam_stub objects), the reference's
own known-answer reads (src/tests/test_bnd_leads.py), error behaviour, and the hand-over into the clustering path.
GPU tier (tests/test_extract_gpu.py) runs the same comparisons through libsniffles_amd.so on the MI355X.
"""
import struct

import numpy as np
import pytest

import cases
import extract_util as xu
import golden_util as gu


@pytest.fixture
def emu_lib():
    from emu import emu
    return emu.lib()


def oracle_cfg(case):
    import extract_oracle as eo
    return eo.Cfg(**case["cfg"])


class DevCfg:
    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)


@pytest.mark.parametrize("name", sorted(cases.EXTRACT))
def test_oracle_matches_reference(name):
    import extract_oracle as eo
    case = cases.EXTRACT[name]
    recs = cases.extract_records(case)
    doc = gu.load(name)
    assert xu.records_sha(recs) == doc["input_sha"], "seeded record generator drifted: regenerate the goldens"
    st, en = case["region"]
    out = eo.extract_region(recs.blob, recs.rec_off, recs.ref_names, case["contig"], st, en, oracle_cfg(case),
                            case["read_id_offset"])
    clen = recs.ref_lens[recs.ref_names.index(case["contig"])]
    xu.check_against_golden(doc["expected"], out["rows"], out["reads"], out["qc_nm_threshold"], out["read_id"], clen)


@pytest.mark.parametrize("name", sorted(cases.EXTRACT))
def test_kernel_bodies_match_reference(name, emu_lib):
    from sniffles_amd import extract
    case = cases.EXTRACT[name]
    recs = cases.extract_records(case)
    doc = gu.load(name)
    st, en = case["region"]
    ti, info = extract.extract_region(recs, case["contig"], st, en, DevCfg(**case["cfg"]), case["read_id_offset"])
    reads = list(zip(ti.read_start.tolist(), ti.read_end.tolist(), ti.read_hp.tolist()))
    xu.check_against_golden(doc["expected"], xu.canon_leads(ti), reads, float(ti.qc_nm_threshold).hex(), info.read_id,
                            ti.contig_len)
    assert info.read_count == len(reads)


@pytest.mark.skipif(not __import__("make_ref").ref_root(), reason="needs the reference (its checkout, or the staged build oracle/_ref that make_ref.py compiles)")
def test_extraction_matches_reference_on_random_tables_and_filters(monkeypatch, capsys):
    """oracle/ref_extractfuzz.py: random record tables, regions, read-id offsets and read filters; the unmodified reference's
    build_leadtab against the oracle and against the kernels (the goldens pin eleven cases)."""
    import ref_extractfuzz
    monkeypatch.setattr("sys.argv", ["ref_extractfuzz.py", "10", "9000"])
    ref_extractfuzz.main()
    out = capsys.readouterr().out
    assert "mismatching 0 " in out and "MISMATCH" not in out, out[-2000:]


# src/tests/test_bnd_leads.py: (contig, read name, supplementary, reverse) -> asserted Lead.for_bnd result
# (lead.contig, lead.ref_start, mate_contig, mate_ref_start, is_first, is_reverse).  The "Red" / HG002 classes of that
# file describe same-strand splits, for which the current reference code returns None (leadprov.py:86-87); they are
# covered by the goldens above, not by asserted values.
ORANGE = ("chr1", 23_272_628, "chr5", 52_747_359, True, True)      # test_bnd_leads.py:58-69
GREEN = ("chr18", 21_493_610, "chr20", 25_499_120, False, False)   # test_bnd_leads.py:124-132
KNOWN_BND = [
    ("chr1", "fcdb7746-5405-4548-9d72-3a0c81903e1c", False, False, ORANGE),
    ("chr1", "4c68b01d-b732-49f3-9e4a-6f1594ac5f0a", False, True, ORANGE),
    ("chr1", "5089c480-4eae-4c61-87f8-7278dea0daaa", True, False, ORANGE),
    ("chr1", "5647a0ed-80f2-4c6f-bbe4-937d95ac327b", True, True, ORANGE),
    ("chr18", "7c40fcdd-2d5a-4302-aead-a5ed5bd452a3", False, False, GREEN),
    ("chr18", "7297cbb7-714c-4586-998a-017051004b25", False, True, GREEN),
    ("chr18", "42353033-1bbd-4a0c-84dc-cbd6068295f3", True, False, GREEN),
    ("chr18", "90398957-a526-49ad-be1b-2665c1b8189e", True, True, GREEN),
]


def known_bnd_rows(rows_by_contig):
    for contig, qname, supp, rev, want in KNOWN_BND:
        hits = [r for r in rows_by_contig[contig] if r[1] == qname and r[10] == "BND_SA"]
        assert len(hits) == 1, qname
        r = hits[0]
        assert (r[2], r[3], r[18][0], r[18][1], r[18][2], r[18][3]) == want
        assert r[16] is False and r[7] == ("-" if rev else "+")


def test_reference_known_answer_reads_oracle():
    import extract_oracle as eo
    recs = cases.extract_records(cases.EXTRACT["extract_hg008_chr1"])
    rows = {}
    for contig in ("chr1", "chr18"):
        # the reference test calls Lead.for_bnd on the read directly: no MAPQ / length filter
        out = eo.extract_region(recs.blob, recs.rec_off, recs.ref_names, contig, 0, 2 ** 31 - 1,
                                eo.Cfg(mapq=0, min_alignment_length=0))
        rows[contig] = out["rows"]
    known_bnd_rows(rows)


def test_reference_known_answer_reads_kernels(emu_lib):
    from sniffles_amd import extract
    recs = cases.extract_records(cases.EXTRACT["extract_hg008_chr1"])
    rows = {}
    for contig in ("chr1", "chr18"):
        ti, _ = extract.extract_region(recs, contig, 0, 2 ** 31 - 1, DevCfg(mapq=0, min_alignment_length=0))
        rows[contig] = xu.canon_leads(ti)
    known_bnd_rows(rows)


def _one_read(tags: bytes, ops=((0, 2000),), flag=0, mapq=60, seq=True):
    from sniffles_amd import bam, synth_bam
    qlen = sum(n for op, n in ops if op in (0, 1, 4, 7, 8))
    codes = np.full(qlen if seq else 0, 1, np.uint8)
    rec = synth_bam.make_record(0, 1000, mapq, flag, "r1", list(ops), codes, tags)
    if not seq:   # l_seq = 0 with a CIGAR that consumes query bases
        pass
    return bam.records_from_list(["c1", "c2"], [100000, 50000], [rec])


@pytest.mark.parametrize("tags,match", [
    (b"HPC\x03", "HP tag outside"),
    (b"SAZc2,100,+,50M,60;\0", "6 fields"),
    (b"SAZc9,100,-,50M,60,1;\0", "not in the header"),
    (b"SAZc2,1x0,-,50M,60,1;\0", "plain integer"),
    (b"NMZabc\0", "not an integer"),
    (b"XXq\x01", "malformed auxiliary"),
])
def test_inputs_the_reference_raises_on_fail_the_call(tags, match, emu_lib):
    import extract_oracle as eo
    from sniffles_amd import extract
    recs = _one_read(tags)
    if b"c9" not in tags:   # a contig missing from the header is a restriction of the device table, not of the reference
        with pytest.raises((eo.ExtractError, ValueError)):
            eo.extract_region(recs.blob, recs.rec_off, recs.ref_names, "c1", 0, 100000)
    with pytest.raises(Exception, match=match):
        extract.extract_region(recs, "c1", 0, 100000)


def test_missing_sequence_with_insertion_fails(emu_lib):
    from sniffles_amd import extract
    recs = _one_read(b"", ops=((0, 1000), (1, 60), (0, 1000)), seq=False)
    with pytest.raises(Exception, match="without sequence"):
        extract.extract_region(recs, "c1", 0, 100000)


def test_empty_and_foreign_records(emu_lib):
    from sniffles_amd import bam, extract
    empty = bam.records_from_list(["c1"], [1000], [])
    ti, info = extract.extract_region(empty, "c1", 0, 1000)
    assert ti.n_leads == 0 and ti.n_reads == 0 and info.read_count == 0 and ti.qc_nm_threshold == 0.0
    recs = _one_read(b"NMC\x05")
    ti, info = extract.extract_region(recs, "c2", 0, 50000)      # the record is on c1
    assert ti.n_leads == 0 and ti.n_reads == 0
    ti, info = extract.extract_region(recs, "c1", 0, 100000, read_id_offset=41)
    assert ti.n_reads == 1 and info.read_id == 42 and ti.qc_nm_threshold == 5 / 2001.0


def test_record_table_is_validated(emu_lib):
    from sniffles_amd import extract
    recs = _one_read(b"")
    recs.rec_off[1] -= 4
    with pytest.raises(Exception, match="block_size"):
        extract.extract_region(recs, "c1", 0, 100000)


def test_extracted_task_feeds_the_clustering_path(emu_lib, oracle_mod):
    """Extraction output is a valid task input: clustering + calling on it (kernel bodies) equals the C oracle."""
    from sniffles_amd import extract, lib, records
    from sniffles_amd.config import SnifflesConfig
    tis = []
    for k, name in enumerate(("extract_fuzz_a", "extract_lowq_short")):
        case = cases.EXTRACT[name]
        recs = cases.extract_records(case)
        ti, _ = extract.extract_region(recs, case["contig"], *case["region"], DevCfg(**case["cfg"]), task_id=k)
        tis.append(ti)
    cfg = SnifflesConfig(minsupport=2)
    with lib.Batch(cfg, tis) as b:
        b.call_candidates()
        b.finalize()
        got = records.records(b.fetch(1), tis, "final")
    exp = records.records(oracle_mod.run(cfg, tis, True), tis, "final")
    assert got == exp
    assert sum(len(g) for g in got) > 20


def check_device_handover(_lib, oracle_mod):
    """Extraction results that stay in HBM (snf_batch_add_task_device) in one batch with a task that comes from the host:
    the records equal the all-host batch's and the oracle's; the lazy host copy of the columns equals the eager one."""
    from sniffles_amd import extract, lib, records, soa
    from sniffles_amd.config import SnifflesConfig
    names = ("extract_fuzz_a", "extract_lowq_short", "extract_fuzz_window")
    host, dev, keep = [], [], []
    for k, name in enumerate(names):
        case = cases.EXTRACT[name]
        recs = cases.extract_records(case)
        ti, _ = extract.extract_region(recs, case["contig"], *case["region"], DevCfg(**case["cfg"]), task_id=k)
        host.append(ti)
        if k == 1:
            dev.append(ti)                      # a host task between two device tasks
            continue
        tid, _, x = extract.extract_region_device(recs, case["contig"], *case["region"], DevCfg(**case["cfg"]), task_id=k)
        assert isinstance(tid, soa.DeviceTaskInput) and tid.n_leads == ti.n_leads and tid.n_reads == ti.n_reads
        dev.append(tid)
        keep.append(x)
    cfg = SnifflesConfig(minsupport=2)
    out = []
    for tis in (host, dev):
        with lib.Batch(cfg, tis) as b:
            b.call_candidates()
            b.finalize()
            out.append(records.records(b.fetch(1), host, "final"))
    assert out[0] == out[1] == records.records(oracle_mod.run(cfg, host, True), host, "final")
    for a, d in zip(host, dev):
        for name in a.leads:
            assert np.array_equal(a.leads[name], d.leads[name], equal_nan=a.leads[name].dtype.kind == "f"), name
        assert np.array_equal(a.seq_pool, d.seq_pool) and np.array_equal(a.read_end, d.read_end)
    for x in keep:
        x.close()


def test_device_handover_equals_host_path_emu(emu_lib, oracle_mod):
    check_device_handover(emu_lib, oracle_mod)


@pytest.mark.gpu
def test_device_handover_equals_host_path_gpu(oracle_mod):
    check_device_handover(None, oracle_mod)


# ---- SA strings the wave form cuts into elements by ballots (a lane per element, csrc/snf_extract.hip part A): shapes the random
# tables rarely produce.  Every case: the live reference (when its checkout or staged build is here), the oracle, the wave form
# and the thread form (the reference's loops as they are) must agree on the leads - or all must fail.
_E = "c1,%d,%s,%s,%d,%d"


def _sa(*parts):
    return ("SAZ" + "".join(parts)).encode() + b"\0"


def _el(pos=5000, strand="+", cigar="1000S500M500S", mapq=60, nm=3, contig="c1"):
    return f"{contig},{pos},{strand},{cigar},{mapq},{nm}"


SA_SHAPES = {
    "empty_string": _sa(""),
    "only_semicolons": _sa(";;;"),
    "no_trailing_semicolon": _sa(_el()),
    "leading_and_doubled_semicolons": _sa(";;", _el(pos=9000), ";;", _el(pos=20000, strand="-"), ";;"),
    "three_elements": _sa(_el(pos=3100, cigar="300M1700S"), ";", _el(pos=8000, strand="-", cigar="1500S500M"), ";", _el(contig="c2", pos=77, cigar="900S400M700S"), ";"),
    "second_dropped_third_bad_number": _sa(_el(), ";", _el(cigar="10M5Q"), ";", _el(pos="1x"), ";"),       # CIGAR_analyze fails first: no splits, no error
    "second_bad_number_third_dropped": _sa(_el(), ";", _el(pos="1x"), ";", _el(cigar="10M5Q"), ";"),       # the number fails first: the call fails
    "seven_fields_in_the_second": _sa(_el(), ";", _el() + ",9", ";"),
    "five_fields_in_the_first": _sa("c1,5000,+,100M,60;", _el(), ";"),
    "bad_strand_in_the_third": _sa(_el(), ";", _el(pos=7000), ";", _el(strand="*"), ";"),
    "mapq_out_of_range": _sa(_el(), ";", _el(mapq=300), ";"),
    "empty_fields": _sa("c1,,+,,60,1;"),
    "long_string_beyond_the_lds_copy": _sa(";".join(_el(pos=1000 + 37 * k, cigar=f"{k + 1}S{1500 - k}M{500 - 1}S") for k in range(40)), ";"),
    "more_elements_than_the_table": _sa(";".join(_el(pos=1000 + 10 * k) for k in range(70)), ";"),
}


@pytest.mark.parametrize("shape", sorted(SA_SHAPES))
@pytest.mark.parametrize("splits", ["default", "many"])
def test_sa_string_shapes(shape, splits, emu_lib, monkeypatch):
    import extract_oracle as eo
    from sniffles_amd import extract
    tags = b"NMC\x07" + SA_SHAPES[shape]
    recs = _one_read(tags, ops=((4, 300), (0, 1200), (1, 80), (0, 400), (4, 100)))
    kw = dict(max_splits_base=100) if splits == "many" else {}
    outcomes = {}

    def run(name, fn):
        try:
            outcomes[name] = ("ok", fn())
        except Exception as e:
            outcomes[name] = ("raised", f"{type(e).__name__}: {e}"[:200])

    def kernels():
        ti, info = extract.extract_region(recs, "c1", 0, 100000, DevCfg(**kw))
        return xu.canon_leads(ti)

    run("oracle", lambda: eo.extract_region(recs.blob, recs.rec_off, recs.ref_names, "c1", 0, 100000, eo.Cfg(**kw))["rows"])
    run("wave", kernels)
    monkeypatch.setenv("SNF_EXTRACT_THREAD", "1")
    run("thread", kernels)
    monkeypatch.delenv("SNF_EXTRACT_THREAD")
    import make_ref
    if make_ref.ref_root():
        import ref_harness as rh
        args = ("--max-splits-base", "100") if splits == "many" else ()
        ref = rh.run_reference_extract(recs, "c1", 0, 100000, args, 0, {})
        outcomes["reference"] = ("raised", ref["error"]) if "error" in ref else ("ok", ref["leads"])
    kinds = {k: v[0] for k, v in outcomes.items()}
    if shape == "more_elements_than_the_table" and splits == "many":      # a limit of the device table (64 segments), not of the reference
        assert kinds["wave"] == kinds["thread"] == "raised" and "more split alignments" in outcomes["wave"][1], outcomes
        return
    restricted = {"bad_strand_in_the_third": "SA strand",      # the reference keeps any strand string (and compares strings); the device table holds a bit
                  "mapq_out_of_range": "column range"}         # ... and any integer as MAPQ; the column has eight bits
    if shape in restricted:
        assert kinds["wave"] == kinds["thread"] == "raised" and restricted[shape] in outcomes["wave"][1] and outcomes["wave"][1] == outcomes["thread"][1], outcomes
        return
    assert len(set(kinds.values())) == 1, {k: (v[0], v[1] if v[0] == "raised" else len(v[1])) for k, v in outcomes.items()}
    if kinds["wave"] == "ok":
        for k, v in outcomes.items():
            assert v[1] == outcomes["wave"][1], k
    else:
        assert outcomes["wave"][1] == outcomes["thread"][1], outcomes      # the same first error, in the same words
