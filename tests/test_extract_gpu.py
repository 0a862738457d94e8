"""GPU parity of signature extraction through the C-ABI of libsniffles_amd.so on the MI355X:
the wave-per-record kernels against the reference goldens, the oracle restatement on seeded fuzz records, the
thread form of the same kernels, and the hand-over into the clustering path.  Bit-exact everywhere."""
import numpy as np
import pytest

import cases
import extract_util as xu
import golden_util as gu
from test_extract import DevCfg, known_bnd_rows

pytestmark = pytest.mark.gpu


def dev_extract(recs, contig, st, en, cfg=None, **kw):
    from sniffles_amd import extract
    return extract.extract_region(recs, contig, st, en, cfg, **kw)


def as_tuple(ti, info):
    reads = list(zip(ti.read_start.tolist(), ti.read_end.tolist(), ti.read_hp.tolist()))
    return xu.canon_leads(ti), reads, float(ti.qc_nm_threshold).hex(), info.read_id


@pytest.mark.parametrize("form", ["wave", "thread"])
@pytest.mark.parametrize("name", sorted(cases.EXTRACT))
def test_extraction_matches_reference_golden(name, form, monkeypatch):
    if form == "thread":
        monkeypatch.setenv("SNF_EXTRACT_THREAD", "1")
    else:
        monkeypatch.delenv("SNF_EXTRACT_THREAD", raising=False)
    case = cases.EXTRACT[name]
    recs = cases.extract_records(case)
    doc = gu.load(name)
    assert xu.records_sha(recs) == doc["input_sha"]
    ti, info = dev_extract(recs, case["contig"], *case["region"], DevCfg(**case["cfg"]), read_id_offset=case["read_id_offset"])
    rows, reads, nm_hex, read_id = as_tuple(ti, info)
    xu.check_against_golden(doc["expected"], rows, reads, nm_hex, read_id, ti.contig_len)
    assert info.ms_count > 0 and info.ms_emit > 0


def test_reference_known_answer_reads():
    recs = cases.extract_records(cases.EXTRACT["extract_hg008_chr1"])
    rows = {}
    for contig in ("chr1", "chr18"):
        ti, _ = dev_extract(recs, contig, 0, 2 ** 31 - 1, DevCfg(mapq=0, min_alignment_length=0))
        rows[contig] = xu.canon_leads(ti)
    known_bnd_rows(rows)


def test_fuzz_records_against_oracle():
    import extract_oracle as eo
    from sniffles_amd import bam, synth_bam
    total = 0
    for seed in range(100, 124):
        names, lens, recs = synth_bam.gen_records(seed, 350, sa_frac=0.45, read_len_mean=2500 + 100 * (seed % 7))
        R = bam.records_from_list(names, lens, recs)
        st, en = (0, 400000) if seed % 3 else (40000, 330000)
        kw = {} if seed % 4 else dict(mapq=0, min_alignment_length=200, dev_keep_lowqual_splits=True, max_splits_kb=1.0)
        want = eo.extract_region(R.blob, R.rec_off, R.ref_names, "chrA", st, en, eo.Cfg(**kw), seed)
        ti, info = dev_extract(R, "chrA", st, en, DevCfg(**kw), read_id_offset=seed)
        rows, reads, nm_hex, read_id = as_tuple(ti, info)
        assert rows == want["rows"] and reads == want["reads"]
        assert nm_hex == want["qc_nm_threshold"] and read_id == want["read_id"]
        total += len(rows)
    assert total > 15000


def test_long_reads_wave_equals_thread_and_oracle(monkeypatch):
    """ONT-like records (20-kb reads, thousands of CIGAR operations each): the wave kernels, the thread kernels and,
    the oracle agree; leads come out in record order."""
    import extract_oracle as eo
    from sniffles_amd import bam, synth_bam
    names, lens, recs = synth_bam.gen_records(77, 1200, style="ont", read_len_mean=20000, sa_frac=0.2,
                                              ref_lens=(3000000, 300000, 300000, 100000))
    R = bam.records_from_list(names, lens, recs)
    monkeypatch.delenv("SNF_EXTRACT_THREAD", raising=False)
    ti_w, info_w = dev_extract(R, "chrA", 0, 3000000)
    monkeypatch.setenv("SNF_EXTRACT_THREAD", "1")
    ti_t, info_t = dev_extract(R, "chrA", 0, 3000000)
    for k in ti_w.leads:
        assert np.array_equal(ti_w.leads[k], ti_t.leads[k], equal_nan=True), k
    assert np.array_equal(ti_w.seq_pool, ti_t.seq_pool) and np.array_equal(ti_w.read_end, ti_t.read_end)
    assert ti_w.qc_nm_threshold == ti_t.qc_nm_threshold and info_w.read_id == info_t.read_id
    assert ti_w.n_leads > 1000 and np.all(np.diff(ti_w.leads["read_id"].astype(np.int64)) >= 0)
    want = eo.extract_region(R.blob, R.rec_off, R.ref_names, "chrA", 0, 3000000)
    rows, reads, nm_hex, read_id = as_tuple(ti_w, info_w)
    assert rows == want["rows"] and reads == want["reads"] and nm_hex == want["qc_nm_threshold"]


def test_sa_string_shapes_on_the_device(monkeypatch):
    """The SA shapes of tests/test_extract.py (empty / doubled elements, the first failing element decides, strings beyond the LDS
    copy, more elements than the segment table) in ONE record table: wave form == thread form == oracle on the MI355X."""
    import extract_oracle as eo
    from sniffles_amd import bam, synth_bam
    from test_extract import SA_SHAPES
    ok = [k for k in sorted(SA_SHAPES) if k not in ("second_bad_number_third_dropped", "seven_fields_in_the_second", "five_fields_in_the_first",
                                                     "bad_strand_in_the_third", "mapq_out_of_range", "empty_fields")]
    ops = [(4, 300), (0, 1200), (1, 80), (0, 400), (4, 100)]
    qlen = sum(n for op, n in ops if op in (0, 1, 4))
    recs = [synth_bam.make_record(0, 1000 + 50 * i, 60, 0x10 * (i % 2), f"r{i}", ops, np.full(qlen, 1 + (i % 4), np.uint8), b"NMC\x07" + SA_SHAPES[k])
            for i, k in enumerate(ok * 3)]
    R = bam.records_from_list(["c1", "c2"], [100000, 50000], recs)
    for kw in ({}, dict(max_splits_base=60)):
        want = eo.extract_region(R.blob, R.rec_off, R.ref_names, "c1", 0, 100000, eo.Cfg(**kw))
        monkeypatch.delenv("SNF_EXTRACT_THREAD", raising=False)
        rows_w = as_tuple(*dev_extract(R, "c1", 0, 100000, DevCfg(**kw)))
        monkeypatch.setenv("SNF_EXTRACT_THREAD", "1")
        rows_t = as_tuple(*dev_extract(R, "c1", 0, 100000, DevCfg(**kw)))
        assert rows_w == rows_t
        assert rows_w[0] == want["rows"] and rows_w[1] == want["reads"] and len(want["rows"]) > 3 * len(ok)
    monkeypatch.delenv("SNF_EXTRACT_THREAD", raising=False)
    from sniffles_amd import lib
    from test_extract import _one_read
    for k, match in (("second_bad_number_third_dropped", "plain integer"), ("seven_fields_in_the_second", "6 fields"), ("empty_fields", "plain integer")):
        with pytest.raises(lib.SnifflesAmdError, match=match):
            dev_extract(_one_read(b"NMC\x07" + SA_SHAPES[k], ops=tuple(ops)), "c1", 0, 100000)


def test_errors_fail_loudly():
    from sniffles_amd import lib
    from test_extract import _one_read
    for tags, match in ((b"HPC\x03", "HP tag outside"), (b"SAZc2,100,+,50M,60;\0", "6 fields")):
        with pytest.raises(lib.SnifflesAmdError, match=match):
            dev_extract(_one_read(tags), "c1", 0, 100000)


def test_extracted_tasks_run_through_the_whole_path(oracle_mod):
    """BAM records -> extraction kernels -> clustering / calling / consensus kernels; equals the C oracle fed with the
    same task inputs."""
    from sniffles_amd import lib, records
    from sniffles_amd.config import SnifflesConfig
    tis = []
    for k, name in enumerate(("extract_fuzz_a", "extract_lowq_short", "extract_ont_long")):
        case = cases.EXTRACT[name]
        recs = cases.extract_records(case)
        ti, _ = dev_extract(recs, case["contig"], *case["region"], DevCfg(**case["cfg"]), task_id=k)
        tis.append(ti)
    cfg = SnifflesConfig(minsupport=2)
    with lib.Batch(cfg, tis) as b:
        b.call_candidates()
        b.finalize()
        got = records.records(b.fetch(1), tis, "final")
    assert got == records.records(oracle_mod.run(cfg, tis, True), tis, "final")
    assert sum(len(g) for g in got) > 20
