"""Force calling (seam B6 + GenotypeTask.execute, reference parallel.py:299-372, postprocessing.py:69-130): target SVs
matched against the sample's candidates and annotated with its coverage on the device, against what the UNMODIFIED
reference GenotypeTask.execute produces for the same task and targets (tests/golden/genotype_targets.json.gz).
CPU tier: kernels through the host emulation; GPU tier: the real library."""
import pytest

import cases
import genotype_util as gutil
import golden_util as gu
from sniffles_amd import leadprov, parallel, postprocessing, sv
from test_dropin_api import leads_of


def make_task(name, specs, _lib):
    build, kw, _ = cases.ALL[name]
    ti = build()
    cfg = gu.make_config(kw, ti)
    lp = leadprov.LeadProvider(cfg, 0, ti.contig, contig_len=ti.contig_len)
    for ld in leads_of(ti):
        lp.record_lead(ld, int(ld.ref_start / cfg.cluster_binsize) * cfg.cluster_binsize)
    for s, e, hp in zip(ti.read_start.tolist(), ti.read_end.tolist(), ti.read_hp.tolist()):
        lp.record_read(s, e, hp)
    targets = gutil.make_targets(specs, sv.SVCall, sv.SVCallBNDInfo, sv.new_call)
    task = parallel.GenotypeTask(id=ti.task_id, sv_id=ti.sv_id_start, contig=ti.contig, start=0, end=ti.contig_len, config=cfg,
                                 lead_provider=lp, genotype_svs=targets)
    task.tandem_repeats = None if ti.tr_start is None else list(zip(ti.tr_start.tolist(), ti.tr_end.tolist()))
    return ti, task


def run_case(name, _lib):
    doc = gu.load("genotype_targets")[name]
    ti, task = make_task(doc.get("case", name), doc["specs"], _lib)
    assert gu.input_sha(ti) == doc["input_sha"]
    exp = doc["expected"]
    if "error" in exp:
        with pytest.raises(UnboundLocalError):
            task.execute()
        return
    got = task.execute()
    task.close()
    want = exp["targets"]
    rec = gutil.result_records(got)
    assert len(rec) == len(want)
    for g, w in zip(rec, want):
        assert g == w, (g["id"], g, w)


NAMES = gutil.CASES + ["bnd_first"]


@pytest.mark.parametrize("name", [n for n in NAMES if n != "chr20_30x_ont"])
def test_genotype_task_emu(name):
    import emu.emu as E
    run_case(name, E.lib())


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_genotype_task_gpu(name):
    run_case(name, None)


@pytest.mark.skipif(not __import__("make_ref").ref_root(), reason="needs the reference (its checkout, or the staged build oracle/_ref that make_ref.py compiles)")
def test_force_calling_matches_reference_on_random_targets_and_options(monkeypatch, capsys):
    """oracle/ref_genotypefuzz.py: random adversarial tasks, target sets derived from the reference's own candidates, random
    options; the unmodified reference's GenotypeTask.execute against this package's."""
    import ref_genotypefuzz
    monkeypatch.setattr("sys.argv", ["ref_genotypefuzz.py", "25", "9000"])
    ref_genotypefuzz.main()
    out = capsys.readouterr().out
    assert "mismatching 0 " in out and "MISMATCH" not in out, out[-2000:]


def test_coverage_needs_the_device_batch():
    import emu.emu as E
    doc = gu.load("genotype_targets")["long_ins"]
    ti, task = make_task("long_ins", doc["specs"], E.lib())
    with pytest.raises(RuntimeError, match="device batch"):
        postprocessing.coverage(task.genotype_svs, task.lead_provider)      # before call_candidates: nothing on the device
    task.execute()
    assert postprocessing.coverage([], task.lead_provider) == task.coverage_average_total
    task.close()


def run_genotype_vcf(name, _lib):
    import io
    from sniffles_amd import pipeline
    from test_pipeline import config_for, records_sha
    doc = gu.load("genotype_vcf")[name]
    recs = cases.SAMPLES[name][0]()
    assert records_sha(recs) == doc["input_sha"]
    buf = io.StringIO()
    n = pipeline.genotype_vcf(recs, config_for(()), io.StringIO(doc["vcf_in"]), buf)
    assert buf.getvalue() == doc["vcf_out"]
    assert n == sum(1 for ln in doc["vcf_out"].split("\n") if ln and not ln.startswith("#")) > 50


@pytest.mark.parametrize("name", ["sample_splits_14x", "sample_two_contigs_12x"])
def test_genotype_vcf_end_to_end_emu(name):
    """`--genotype-vcf` end to end: target VCF + BAM -> the target lines with this sample's genotypes, equal character by
    character to what the unmodified reference writes (its VCF reader, GenotypeTask.execute per contig over the pysam
    stand-in, its rewriter): header additions, missing FORMAT declarations, TRA -> BND, sequence-resolved records, targets
    on unprocessed contigs dropped."""
    import emu.emu as E
    run_genotype_vcf(name, E.lib())


# ---- --reqc: postprocessing.genotype_sv on finalized candidates (CombineTask.execute, parallel.py:507-508)
def run_regenotype(name, _lib=None):
    import cases
    from sniffles_amd import parallel, pipeline, postprocessing, sv
    doc = gu.load("regenotype")[name]
    build, kw, _ = cases.ALL[name]
    ti = build()
    assert gu.input_sha(ti) == doc["input_sha"]
    cfg = gu.make_config(kw, ti)
    task = parallel.CallTask(id=ti.task_id, sv_id=ti.sv_id_start, contig=ti.contig, start=0, end=ti.contig_len, config=cfg)
    task.lead_provider = pipeline._Extracted(ti)
    cands = task.call_candidates(False, cfg)
    task.finalize_candidates(cands, True, cfg)
    task.close()
    cands = [c for c in cands if c.svtype in sv.TYPES]
    postprocessing.genotype_svs(cands, cfg)
    got = []
    for c in cands:
        gt = c.genotypes.get(0)
        got.append(dict(id=c.id, filter=c.filter, qc=bool(c.qc), vaf=c.info.get("VAF"), phase=c.info.get("PHASE"),
                        gt=None if gt is None else [gt[0], gt[1], gt[2], gt[3], gt[4], list(gt[5]) if gt[5] is not None else None]))
    assert got == doc["calls"]


REGENOTYPE = sorted(gu.load("regenotype"))


@pytest.mark.parametrize("name", REGENOTYPE)
def test_regenotype_matches_reference_emu(name):
    import emu.emu as E
    run_regenotype(name, E.lib())


@pytest.mark.gpu
@pytest.mark.parametrize("name", REGENOTYPE)
def test_regenotype_matches_reference_gpu(name):
    run_regenotype(name)
