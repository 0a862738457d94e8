"""End to end (BASELINE.json configs[0] shape: one sample, BAM in, VCF / SNF out): a synthetic coordinate-sorted BAM
through sniffles_amd.pipeline.call_sample - extraction, clustering, calling, QC, genotyping, consensus on the device,
VCF / SNF written by this package - against what the UNMODIFIED reference's call_sample flow produces from the same BAM
(tests/golden/sample_*.json.gz, oracle/ref_harness.py::run_reference_call_sample): the VCF text character by character,
the SNF content field by field.  CPU tier: kernels through the host emulation; GPU tier: the real library."""
import io
import json

import pytest

import cases
import golden_util as gu
import snf_util as su
import vcf_util as vu
from sniffles_amd import bam, pipeline, snf, sv
from sniffles_amd.config import SnifflesConfig
from test_vcf import assert_same_text


def records_sha(recs):
    import hashlib
    h = hashlib.sha256()
    h.update(recs.blob.tobytes())
    h.update(recs.rec_off.tobytes())
    h.update(repr((recs.ref_names, recs.ref_lens)).encode())
    return h.hexdigest()


def config_for(args):
    kw, a = {}, list(args)
    while a:
        k = a.pop(0)[2:].replace("-", "_")
        if a and not a[0].startswith("--"):
            v = a.pop(0)
            for conv in (int, float):
                try:
                    v = conv(v)
                    break
                except ValueError:
                    pass
            kw[k] = v
        else:
            kw[k] = True
    cfg = SnifflesConfig(**kw)
    for k, v in vu.FIXED.items():
        setattr(cfg, k, v)
    return cfg


def run_sample(name, tmp_path, _lib, through_file):
    build, args = {**cases.SAMPLES, **cases.SAMPLES_EMU}[name]
    doc = gu.load(name)
    recs = build()
    assert records_sha(recs) == doc["input_sha"]
    if through_file:      # the container layer too: BGZF blocks on disk -> inflate -> record table
        path = tmp_path / "sample.bam"
        offs = recs.rec_off.tolist()
        raw = bam.bam_stream(recs.ref_names, recs.ref_lens, [recs.blob[a:b].tobytes() for a, b in zip(offs[:-1], offs[1:])])
        path.write_bytes(bam.bgzf_deflate(raw))
        tr_keep = getattr(recs, "tandem_repeats", None)
        recs = bam.read_bam(str(path))
        recs.tandem_repeats = tr_keep
        assert records_sha(recs) == doc["input_sha"]
    # VCF only
    buf = io.StringIO()
    tr = getattr(recs, "tandem_repeats", None)
    res = pipeline.call_sample(recs, config_for(args), vcf_handle=buf, tandem_repeats=tr)
    assert res.read_count == doc["read_count"]
    assert_same_text(buf.getvalue(), doc["vcf"])
    assert res.vcf_records == len(vu.split_text(doc["vcf"])[1])
    # the same text straight from the record table (vcf.VCF.write_records): no SVCall objects
    buf = io.StringIO()
    res2 = pipeline.call_sample(recs, config_for(args), vcf_handle=buf, tandem_repeats=tr, objects=False)
    assert_same_text(buf.getvalue(), doc["vcf"])
    assert res2.vcf_records == res.vcf_records and res2.read_count == res.read_count and not res2.calls
    # VCF + SNF (the candidates are not QC-filtered then)
    buf = io.StringIO()
    cfg = config_for(args)
    snf_path = str(tmp_path / "sample.snf")
    res = pipeline.call_sample(recs, cfg, vcf_handle=buf, snf_path=snf_path, tandem_repeats=tr)
    assert_same_text(buf.getvalue(), doc["vcf_with_snf"])
    assert res.snf_candidates == doc["snf_candidates"]
    f = snf.SNFile.open(snf_path, cfg)
    got = {c: json.loads(json.dumps(su.file_record(f, c, sv.TYPES), sort_keys=True)) for c, _ in res.contig_lengths}
    f.close()
    assert sorted(got) == sorted(doc["snf"])
    for c in got:
        assert sorted(got[c]["blocks"]) == sorted(doc["snf"][c]["blocks"])
        for b in got[c]["blocks"]:
            assert got[c]["blocks"][b] == doc["snf"][c]["blocks"][b], (c, b)
        assert got[c] == doc["snf"][c]


@pytest.mark.parametrize("name", sorted({**cases.SAMPLES, **cases.SAMPLES_EMU}))
def test_bam_to_vcf_and_snf_emu(name, tmp_path):
    import emu.emu as E
    run_sample(name, tmp_path, E.lib(), through_file=(name == "sample_mosaic_20x"))


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(cases.SAMPLES))
def test_bam_to_vcf_and_snf_gpu(name, tmp_path):
    run_sample(name, tmp_path, None, through_file=True)


@pytest.mark.skipif(not __import__("make_ref").ref_root(), reason="needs the reference (its checkout, or the staged build oracle/_ref that make_ref.py compiles)")
def test_bam_to_vcf_matches_reference_under_random_command_lines(monkeypatch, capsys):
    """oracle/ref_samplefuzz.py: random synthetic samples, random command lines (some thirty options), the unmodified
    reference's call_sample flow against pipeline.call_sample, VCF text character by character."""
    import ref_samplefuzz
    monkeypatch.setattr("sys.argv", ["ref_samplefuzz.py", "4", "5000"])
    ref_samplefuzz.main()
    out = capsys.readouterr().out
    assert "mismatching 0 " in out and "MISMATCH" not in out, out[-2000:]


@pytest.mark.skipif(not __import__("make_ref").ref_root(), reason="needs the reference (its checkout, or the staged build oracle/_ref that make_ref.py compiles)")
def test_population_merge_matches_reference_under_random_command_lines(monkeypatch, capsys):
    """oracle/ref_populationfuzz.py: random populations (2-5 samples), random --combine-* command lines; BAM records -> .snf
    files -> merged VCF by the unmodified reference against this package, over its own files and over the reference's."""
    import ref_populationfuzz
    monkeypatch.setattr("sys.argv", ["ref_populationfuzz.py", "2", "7000"])
    ref_populationfuzz.main()
    out = capsys.readouterr().out
    assert "mismatching 0 " in out and "MISMATCH" not in out, out[-2000:]


def test_regions_restrict_extraction_and_coverage():
    """--regions (config.regions_by_contig; sniffles:330-351, leadprov.py:445-472): one whole-contig interval is the plain run;
    a sub-interval leaves only calls whose leads lie inside it; the same interval twice doubles the coverage the calls see."""
    import emu.emu as E
    E.lib()                                              # the host tier becomes the library of this test
    name = "sample_two_contigs_12x"
    recs = cases.SAMPLES[name][0]()
    tr = getattr(recs, "tandem_repeats", None)

    def text(regions):
        cfg = config_for(())
        cfg.regions_by_contig = regions
        buf = io.StringIO()
        res = pipeline.call_sample(recs, cfg, vcf_handle=buf, tandem_repeats=tr)
        return buf.getvalue(), res
    big = [(c, n) for c, n in zip(recs.ref_names, recs.ref_lens) if n >= 1_000_000]
    plain, res0 = text({})
    whole, res1 = text({c: [(c, 0, n)] for c, n in big})
    assert whole == plain and res1.read_count == res0.read_count
    c0, n0 = big[0]
    sub, res2 = text({c0: [(c0, 200_000, 600_000)]})
    rows = [ln.split("\t") for ln in vu.split_text(sub)[1]]
    assert 5 < len(rows) < len(vu.split_text(plain)[1]) and all(r[0] == c0 and 199_000 <= int(r[1]) <= 601_000 for r in rows)
    assert res2.read_count < res0.read_count
    twice, res3 = text({c0: [(c0, 200_000, 600_000), (c0, 200_000, 600_000)]})
    assert res3.read_count == 2 * res2.read_count            # every read is walked once per region, as in the reference


@pytest.mark.skipif(not __import__("make_ref").ref_root(), reason="needs the reference (its checkout, or the staged build oracle/_ref that make_ref.py compiles)")
def test_genotype_vcf_with_regions_matches_reference(tmp_path):
    import ref_harness as rh
    import emu.emu as E
    E.lib()                                              # the host tier becomes the library of this test
    name = "sample_splits_14x"
    doc = gu.load("genotype_vcf")[name]
    recs = cases.SAMPLES[name][0]()
    regs, lines = {}, []
    for c, n in zip(recs.ref_names, recs.ref_lens):
        if n >= 1_000_000:
            regs[c] = [(c, 100_000, 500_000), (c, 450_000, 900_000)]          # overlapping on purpose
            lines += [f"{c}\t100000\t500000", f"{c}\t450000\t900000"]
    bed = tmp_path / "r.bed"
    bed.write_text("\n".join(lines) + "\n")
    ref = rh.run_reference_genotype_vcf(recs, doc["vcf_in"], ("--regions", str(bed)), vu.FIXED)
    cfg = config_for(())
    cfg.regions_by_contig = regs
    buf = io.StringIO()
    pipeline.genotype_vcf(recs, cfg, io.StringIO(doc["vcf_in"]), buf)
    assert buf.getvalue() == (ref["vcf"] if isinstance(ref, dict) else ref) != doc["vcf_out"]


def test_contig_selection_rule():
    cfg = SnifflesConfig()
    assert pipeline.should_process_contig("chr1", 2_000_000, cfg) and not pipeline.should_process_contig("chrUn", 999_999, cfg)
    cfg.all_contigs = True
    assert pipeline.should_process_contig("chrUn", 10, cfg)
    cfg.all_contigs, cfg.contig = False, ["chrUn"]
    assert pipeline.should_process_contig("chrUn", 10, cfg) and not pipeline.should_process_contig("chr1", 2_000_000, cfg)


def test_contig_selection_reference_vectors():
    """The reference's own vectors for `util.should_process_contig` (src/tests/test_params.py:9-25)."""
    cfg = SnifflesConfig()
    assert pipeline.should_process_contig("chr1", 248956422, cfg)                       # normal contig
    assert not pipeline.should_process_contig("fragment", 123456, cfg)                  # short contig excluded
    assert pipeline.should_process_contig("fragment", 123456, SnifflesConfig(contig=["fragment"]))   # ... unless given by -c
    cfg.regions_by_contig = {"fragment": ("fragment", 0, 123456)}                       # ... or by the regions
    assert pipeline.should_process_contig("fragment", 123456, cfg)


def run_population(name, tmp_path, _lib):
    build, args = cases.POPULATIONS[name]
    doc = gu.load(name)
    recs = build()
    assert [records_sha(r) for r in recs] == doc["input_sha"]
    paths = []
    for s, r in enumerate(recs):
        path = str(tmp_path / f"sample{s}.snf")
        pipeline.call_sample(r, config_for(()), snf_path=path, tandem_repeats=getattr(r, "tandem_repeats", None))
        paths.append(path)
    buf = io.StringIO()
    calls = pipeline.combine(paths, config_for(args), vcf_handle=buf)
    assert_same_text(buf.getvalue(), doc["vcf"])
    assert len(calls) >= len(vu.split_text(doc["vcf"])[1]) > 50
    # the same text straight from the group table (vcf.VCF.write_merged): no SVCall objects; also with the options that change the columns
    for extra in ((), ("--phase",), ("--symbolic",), ("--output-rnames",), ("--qc-nm",), ("--mosaic",), ("--minsvlen", "300")):
        with_objects, without = io.StringIO(), io.StringIO()
        if extra:
            pipeline.combine(paths, config_for(tuple(args) + extra), vcf_handle=with_objects)
        else:
            with_objects = buf
        cfg = config_for(tuple(args) + extra)
        assert pipeline.combine(paths, cfg, vcf_handle=without, objects=False) == []
        assert without.getvalue() == with_objects.getvalue(), extra
        assert without.getvalue().count("\n") > 50
    # --no-sort (config.sort = False): a task's records leave in emission order on both paths (result.py:139, :148) - not by position
    unsorted_objects, unsorted_text = io.StringIO(), io.StringIO()
    cfg_a, cfg_b = config_for(args), config_for(args)
    cfg_a.sort = cfg_b.sort = False
    pipeline.combine(paths, cfg_a, vcf_handle=unsorted_objects)
    assert pipeline.combine(paths, cfg_b, vcf_handle=unsorted_text, objects=False) == []
    assert unsorted_text.getvalue() == unsorted_objects.getvalue()
    assert unsorted_text.getvalue() != buf.getvalue() and sorted(unsorted_text.getvalue().splitlines()) == sorted(buf.getvalue().splitlines())
    # into a text file over a binary one (what open(path, "w") hands the writer): the record bytes go to the binary layer as they are,
    # behind whatever the text layer still held
    raw = io.BytesIO()
    handle = io.TextIOWrapper(raw, encoding="utf-8", newline="")
    assert pipeline.combine(paths, config_for(args), vcf_handle=handle, objects=False) == []
    handle.flush()
    assert raw.getvalue().decode("utf-8") == buf.getvalue()
    # the records formatted by several threads outside the interpreter lock (ranges of the emitted groups): the same text
    import os
    try:
        for k in ("1", "3", "7"):
            os.environ["SNF_TEXT_THREADS"] = k
            threaded = io.StringIO()
            assert pipeline.combine(paths, config_for(args), vcf_handle=threaded, objects=False) == []
            assert threaded.getvalue() == buf.getvalue(), k
    finally:
        os.environ.pop("SNF_TEXT_THREADS", None)
    # the merge in runs of contig tasks that overlap host and device work (candstore.execute_many; the default of a big merge): same text,
    # same objects
    from sniffles_amd import candstore
    try:
        for k in ("2", "3"):
            os.environ["SNF_COMBINE_CHUNKS"] = k
            in_runs, in_runs_objects = io.StringIO(), io.StringIO()
            assert pipeline.combine(paths, config_for(args), vcf_handle=in_runs, objects=False) == []
            n_runs = candstore.last_timing.get("chunks", 1)
            assert in_runs.getvalue() == buf.getvalue()
            calls_runs = pipeline.combine(paths, config_for(args), vcf_handle=in_runs_objects)
            assert in_runs_objects.getvalue() == buf.getvalue() and len(calls_runs) == len(calls)
    finally:
        os.environ.pop("SNF_COMBINE_CHUNKS", None)
    return n_runs


@pytest.mark.parametrize("name", sorted(cases.POPULATIONS))
def test_bams_to_merged_vcf_emu(name, tmp_path):
    """BASELINE.json configs[4] shape: every sample BAM -> .snf (this package), then the multi-sample merge over those files
    (this package) - the merged VCF equals the one the unmodified reference produces from the same BAMs through its own
    .snf files, character by character."""
    import emu.emu as E
    assert run_population(name, tmp_path, E.lib()) >= 1      # (runs of contig tasks: tests/test_combine_task.py holds the two-task case)


def test_tandem_repeat_file_loader(tmp_path):
    """sniffles_amd.util.load_tandem_repeats against the reference's function on the same BED (build container), and
    against a hand-checked table everywhere."""
    import os
    import sys
    from sniffles_amd import util
    bed = tmp_path / "tr.bed"
    bed.write_text("chr1\t1000\t1200\tx\nchr1\t300\t400\nchr2\t50\t90\tmotif\t3\nbroken line\nchr2\t700\t900\n")
    got = util.load_tandem_repeats(str(bed), 500)
    assert got == {"chr1": [(0, 900), (500, 1700)], "chr2": [(0, 590), (200, 1400)]}
    # the reference compares a start with the PADDED start of the previous line: 700 after 1000 (padded 500) is "sorted"
    bed2 = tmp_path / "tr2.bed"
    bed2.write_text("chr1\t1000\t1200\nchr1\t700\t800\n")
    assert util.load_tandem_repeats(str(bed2), 500) == {"chr1": [(500, 1700), (200, 1300)]}
    sys.path.insert(0, os.path.join(os.path.dirname(gu.GOLDEN_DIR), "..", "oracle"))
    import ref_harness as rh
    if rh.reference_available():
        ref = rh.load_reference()
        import contextlib
        with contextlib.redirect_stdout(io.StringIO()):
            want = ref.util.load_tandem_repeats(str(bed), 500)
            want2 = ref.util.load_tandem_repeats(str(bed2), 500)
        assert got == want and util.load_tandem_repeats(str(bed2), 500) == want2


# ------------------------------------------------------------------------------------------------ --reference: the N mask end to end
def _fasta_with_N(recs, seed, frac=0.35):
    """{contig: sequence}: random bases with runs of 'N' (200-6000 bp) over about `frac` of every contig, so that many of
    the five coverage samples of a call fall on masked positions."""
    import numpy as np
    rng = np.random.default_rng(seed)
    out = {}
    for name, n in zip(recs.ref_names, recs.ref_lens):
        n = int(n)
        a = rng.choice(np.frombuffer(b"ACGT", np.uint8), n)
        covered = 0
        while covered < frac * n:
            w = int(rng.integers(200, 6000))
            s = int(rng.integers(0, max(1, n - w)))
            a[s:s + w] = ord("N")
            covered += w
        out[name] = a.tobytes().decode("ascii")
    return out


def _with_reference(_lib, name, regions=None, fasta_len_delta=0):
    import ref_harness as rh
    build, args = {**cases.SAMPLES, **cases.SAMPLES_EMU}[name]
    recs = build()
    fasta = _fasta_with_N(recs, 11)
    if fasta_len_delta:      # a FASTA whose contigs are longer / shorter than the BAM header says
        fasta = {k: (v + "N" * fasta_len_delta if fasta_len_delta > 0 else v[:fasta_len_delta]) for k, v in fasta.items()}
    fixed = dict(vu.FIXED)
    cfg = config_for(args)
    cfg.reference = "reference.fa"
    if regions:
        from sniffles_amd.pipeline import regions_of  # noqa: F401
        fixed["regions_by_contig"] = {c: [rh.load_reference().parallel.Region(c, a, b) for a, b in rs] for c, rs in regions.items()}
        cfg.regions_by_contig = {c: [(c, a, b) for a, b in rs] for c, rs in regions.items()}
    exp = rh.run_reference_call_sample(recs, args, fixed=fixed, fasta=fasta)
    plain = rh.run_reference_call_sample(recs, args, fixed=fixed)
    buf = io.StringIO()
    res = pipeline.call_sample(recs, cfg, vcf_handle=buf, tandem_repeats=getattr(recs, "tandem_repeats", None),
                               reference=rh.DictFasta(fasta))
    assert res.read_count == exp["read_count"]
    assert_same_text(buf.getvalue(), exp["vcf"])
    return exp, plain


needs_ref = pytest.mark.skipif(not __import__("make_ref").ref_root(), reason="needs the reference (its checkout, or the staged build oracle/_ref that make_ref.py compiles)")


@needs_ref
@pytest.mark.parametrize("name", ["sample_two_contigs_12x", "sample_splits_14x"])
def test_sample_with_reference_masks_N_coverage_emu(name):
    """`build_leadtab` masks the coverage vector where the reference base is 'N' whenever `--reference` is given
    (leadprov.py:470): genotypes, coverage filters and DR / DV change against the unmasked run, and equal the reference's."""
    import emu.emu as E
    exp, plain = _with_reference(E.lib(), name)
    assert exp["vcf"] != plain["vcf"]           # the mask matters on this sample (otherwise the test shows nothing)


@needs_ref
def test_sample_with_reference_and_regions_emu():
    """--regions: the mask is painted region by region in list order, a later region over an earlier one (leadprov.py:436-440)."""
    import emu.emu as E
    exp, _ = _with_reference(E.lib(), "sample_two_contigs_12x", regions={"chr20": [(400_000, 900_000), (100_000, 500_000)], "chr21": [(0, 1_000_000)]})
    assert len(vu.split_text(exp["vcf"])[1]) > 20


@needs_ref
@pytest.mark.parametrize("delta", [7, -5000])
def test_sample_with_reference_of_other_length_goes_unmasked_emu(delta):
    """A FASTA whose contig is longer than the BAM header's length: the reference's slice assignment raises, it logs and goes on
    unmasked (leadprov.py:441-442) - so does the pipeline.  Shorter: pysam clips the fetch, same outcome."""
    import emu.emu as E
    _with_reference(E.lib(), "sample_two_contigs_12x", fasta_len_delta=delta)


@needs_ref
@pytest.mark.gpu
def test_sample_with_reference_masks_N_coverage_gpu():
    exp, plain = _with_reference(None, "sample_two_contigs_12x")
    assert exp["vcf"] != plain["vcf"]
