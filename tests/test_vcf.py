"""VCF lines (BASELINE.json: "bit-identical VCF POS/SVLEN/SVTYPE/GT"): inputs -> this package's hot
path -> this package's writer, compared character by character with the text the UNMODIFIED reference writer produced
from the unmodified reference's calls for the same inputs (tests/golden/vcf_text.json.gz, oracle/make_golden.py::main_vcf),
plus the reference's own known-answer writer tests (src/tests/test_vcf.py:96-258), restated against this writer.
CPU tier: kernels through the host emulation; GPU tier: the real library."""
import io
from types import SimpleNamespace

import pytest

import cases
import golden_util as gu
import vcf_util as vu
from sniffles_amd import leadprov, parallel, sv, vcf
from test_combine import make_cfg
from test_dropin_api import leads_of

GOLD = None


def gold():
    global GOLD
    if GOLD is None:
        GOLD = gu.load("vcf_text")
    return GOLD


def write_text(calls, cfg, contigs_lengths, fasta):
    buf = io.StringIO()
    w = vcf.VCF(cfg, buf)
    w.reference_handle = fasta
    w.write_header(contigs_lengths)
    n = sum(w.write_call(c) for c in calls)
    assert n == w.call_count
    return buf.getvalue()


def single_sample_text(name, variant, _lib):
    build, kw, _ = cases.ALL[name]
    _, overrides, with_fasta = vu.VARIANTS[variant]
    ti = build()
    assert gu.input_sha(ti) == gold()["single"][name]["input_sha"]
    cfg = gu.make_config({**kw, **overrides}, ti)
    for k, v in vu.FIXED.items():
        setattr(cfg, k, v)
    lp = leadprov.LeadProvider(cfg, 0, ti.contig, contig_len=ti.contig_len)
    for ld in leads_of(ti):
        lp.record_lead(ld, int(ld.ref_start / cfg.cluster_binsize) * cfg.cluster_binsize)
    for s, e, hp in zip(ti.read_start.tolist(), ti.read_end.tolist(), ti.read_hp.tolist()):
        lp.record_read(s, e, hp)
    task = parallel.CallTask(id=ti.task_id, sv_id=ti.sv_id_start, contig=ti.contig, start=0, end=ti.contig_len, config=cfg,
                             lead_provider=lp)
    task.tandem_repeats = None if ti.tr_start is None else list(zip(ti.tr_start.tolist(), ti.tr_end.tolist()))
    calls = task.call_svs(cfg)
    task.close()
    fasta = vu.FakeFasta({ti.contig: ti.contig_len}) if with_fasta else None
    return write_text(calls, cfg, [(ti.contig, ti.contig_len)], fasta)


def single_sample_text_from_records(name, variant, _lib):
    """The same case through the record-table writer (vcf.VCF.write_records): no SVCall objects."""
    import numpy as np
    build, kw, _ = cases.ALL[name]
    _, overrides, with_fasta = vu.VARIANTS[variant]
    assert not with_fasta
    ti = build()
    cfg = gu.make_config({**kw, **overrides}, ti)
    for k, v in vu.FIXED.items():
        setattr(cfg, k, v)
    lp = leadprov.LeadProvider(cfg, 0, ti.contig, contig_len=ti.contig_len)
    for ld in leads_of(ti):
        lp.record_lead(ld, int(ld.ref_start / cfg.cluster_binsize) * cfg.cluster_binsize)
    for s, e, hp in zip(ti.read_start.tolist(), ti.read_end.tolist(), ti.read_hp.tolist()):
        lp.record_read(s, e, hp)
    task = parallel.CallTask(id=ti.task_id, sv_id=ti.sv_id_start, contig=ti.contig, start=0, end=ti.contig_len, config=cfg,
                             lead_provider=lp)
    task.tandem_repeats = None if ti.tr_start is None else list(zip(ti.tr_start.tolist(), ti.tr_end.tolist()))
    res, ti_used = task.call_records(cfg)
    buf = io.StringIO()
    w = vcf.VCF(cfg, buf)
    assert w.can_write_records()
    w.write_header([(ti.contig, ti.contig_len)])
    keep = np.arange(len(res.calls)) if cfg.no_qc else np.flatnonzero(res.calls["qc"] != 0)
    keep = keep[np.argsort(res.calls["pos"][keep], kind="stable")]
    n = w.write_records(res, ti_used, keep)
    task.close()
    assert n == w.call_count
    return buf.getvalue()


def assert_same_text(got, want):
    gh, gr = vu.split_text(got)
    wh, wr = vu.split_text(want)
    assert gh == wh
    assert len(gr) == len(wr)
    for g, w in zip(gr, wr):
        assert vu.key_columns(g) == vu.key_columns(w)      # POS / SVTYPE / SVLEN / GT first: the clearer message
        assert g == w
    assert got == want


EMU_CASES = [c for c in vu.CASES if not c.startswith("chr")] + ["chr18_20x_auto_nm"]


@pytest.mark.parametrize("variant", sorted(vu.VARIANTS))
@pytest.mark.parametrize("name", EMU_CASES)
def test_single_sample_vcf_emu(name, variant):
    import emu.emu as E
    assert_same_text(single_sample_text(name, variant, E.lib()), gold()["single"][name]["text"][variant])


@pytest.mark.parametrize("variant", sorted(v for v in vu.VARIANTS if not vu.VARIANTS[v][2]))
@pytest.mark.parametrize("name", EMU_CASES)
def test_single_sample_vcf_from_the_record_table_emu(name, variant):
    """vcf.VCF.write_records (C formatter over the finalized records, no SVCall objects) writes the reference's text too."""
    import emu.emu as E
    assert_same_text(single_sample_text_from_records(name, variant, E.lib()), gold()["single"][name]["text"][variant])


@pytest.mark.gpu
@pytest.mark.parametrize("variant", sorted(vu.VARIANTS))
@pytest.mark.parametrize("name", vu.CASES)
def test_single_sample_vcf_gpu(name, variant):
    assert_same_text(single_sample_text(name, variant, None), gold()["single"][name]["text"][variant])


def combine_text(name, variant, _lib):
    """Multi-sample merge over the per-sample SNF blocks of the CombineTask golden -> sorted calls -> VCF."""
    from test_combine_task import BlocksReader
    doc = gu.load(name)
    exp = doc["expected"]
    cfg = make_cfg(doc["reference_args"], exp["n_samples"])
    for k, v in vu.FIXED.items():
        setattr(cfg, k, v)
    cfg.sample_ids_vcf = [(s, f"S{s}") for s in range(exp["n_samples"])]
    readers = {s: BlocksReader(exp["contig"], exp["samples"][s]) for s in range(exp["n_samples"])}
    task = parallel.CombineTask(id=7, sv_id=0, contig=exp["contig"], start=0, end=exp["contig_len"], config=cfg)
    calls = sorted(task.execute(readers), key=lambda c: c.pos)
    fasta = vu.FakeFasta({exp["contig"]: exp["contig_len"]}) if variant == "fasta" else None
    return write_text(calls, cfg, [(exp["contig"], exp["contig_len"])], fasta)


@pytest.mark.parametrize("variant", ["plain", "fasta"])
@pytest.mark.parametrize("name", ["combine_task_3samples_lowcov", "combine_task_8samples_dense"])
def test_combined_vcf_emu(name, variant):
    import emu.emu as E
    assert_same_text(combine_text(name, variant, E.lib()), gold()["combine"][name]["text"][variant])


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["plain", "fasta"])
@pytest.mark.parametrize("name", ["combine_task_3samples_lowcov", "combine_task_8samples_dense"])
def test_combined_vcf_gpu(name, variant):
    assert_same_text(combine_text(name, variant, None), gold()["combine"][name]["text"][variant])


# ---- the reference's own writer tests (src/tests/test_vcf.py), same vectors, same assertions
def ref_test_config():
    return SimpleNamespace(sample_ids_vcf=[], output_rnames=True, mosaic_af_max=0.3, mosaic=False, id_prefix="Sniffles.",
                           symbolic=False, max_del_seq_len=50000, genotype_format="GT:GQ:DR:DV", minsvlen=1,
                           qc_nm_measure=True, dev_emit_sv_lengths=True, genotype_none=(".", ".", 0, 0, 0, (None, None)),
                           phase=True, mode="call_sample", max_unknown_pct=0.5)


class Fetch:
    def __init__(self, seq, origin=0):
        self.seq, self.origin, self.calls = seq, origin, []

    def fetch(self, contig, start, end):
        self.calls.append((contig, start, end))
        if start < 0:
            raise ValueError("start out of range")      # pysam
        return self.seq[start - self.origin:end - self.origin]


def ref_test_call(**kw):
    d = dict(contig="chr1", id="unittest-1", qual=10, filter="PASS", info={}, genotypes={}, precise=True, support=100,
             rnames=["ut"], postprocess=None, qc=True, nm=-1, fwd=1, rev=1)
    d.update(kw)
    return sv.SVCall(**d)


def written(handle_seq, origin, **call_kw):
    buf = io.StringIO()
    w = vcf.VCF(ref_test_config(), buf)
    w.reference_handle = Fetch(handle_seq, origin)
    assert w.write_call(ref_test_call(**call_kw)) == 1
    f = buf.getvalue().rstrip("\n").split("\t")
    assert len(f) > 8 and f[0] == "chr1" and f[2] == "Sniffles.unittest-1"
    info = dict(kv.split("=") for kv in f[7].split(";") if "=" in kv)
    return int(f[1]), f[3], f[4], info, w.reference_handle


def test_reference_vector_spec_ins():          # test_vcf.py:96-125 (VCF 4.2 spec, 5.2.2)
    pos, ref, alt, _, h = written("atCga", 0, svtype="INS", ref="N", alt="TAG", pos=3, svlen=3, end=3)
    assert (pos, ref, alt) == (3, "C", "CTAG")
    assert h.calls[-1] == ("chr1", 2, 3)


def test_reference_vector_spec_del():          # test_vcf.py:127-157 (VCF 4.2 spec, 5.2.3)
    pos, ref, alt, info, _ = written("aTCGa", 0, svtype="DEL", ref="N", alt="<DEL>", pos=2, svlen=-2, end=4)
    assert (pos, ref, alt, info["SVLEN"], info["END"]) == (2, "TCG", "T", "-2", "4")


def test_reference_vector_del_issue31():       # test_vcf.py:159-196
    seq = ("CAGTGGGGATGTGCTGCGGGGAGGGGGGCGCGGGTCCGCAGTGGGGATGTGCTGCCGGGAGGGGGGCGCGGGTCCGCAGTGGGGATGTGCTGCCGGGAGGGGGGCGCGGGTCC"
           "GCAGTGGGGATGTGCTGCCGGGAGGGGGGCGCGGGTCCGCAGTGGGGATGTGCTGCCGGGAGGGGGGCGCGGGTCCGCAGTGGGGAT")
    pos, ref, alt, _, _ = written(seq, 964600, svtype="DEL", ref="N", alt="<DEL>", pos=964631, svlen=-75, end=964631 - 75)
    assert (pos, alt) == (964631, "C")
    assert ref == "CGGGTCCGCAGTGGGGATGTGCTGCCGGGAGGGGGGCGCGGGTCCGCAGTGGGGATGTGCTGCCGGGAGGGGGGCG"


def test_reference_vector_unresolved_ins():    # test_vcf.py:198-222: the reference's own test FAILS on its writer
    # (SURVEY.md section 4): with a reference handle the IUPAC clean-up table turns "<INS>" into "<INN>".  A drop-in
    # reproduces what the reference writes, not what its test hopes for.
    pos, ref, alt, _, _ = written("T" * 50, 0, svtype="INS", ref="N", alt="<INS>", pos=2, svlen=20, end=22)
    assert (pos, ref, alt) == (2, "T", "<INN>")


def test_reference_vector_del_end_issue580():  # test_vcf.py:224-258
    seq = ("TTAACCCCTAACCCTAACCCTTGACCCTAACCCTTGACCCTAACCCCTGACCCTGACCCTTAACCCTAACCCCTAACCCTTAACCCTTAAACCTTAACCCTCATCCTCACCC"
           "TCACCCTCACCCCTAACCCTAACCCCTAACCCCTAACCCAAACCCTAACCCTAAACCCTAACCCTAAACCCAACCCAAACCCTAACCT")
    pos, ref, alt, info, _ = written(seq, 180400, svtype="DEL", ref="N", alt="<DEL>", pos=180431, svlen=-91, end=180521)
    assert (pos, alt, info["END"]) == (180431, "C", "180522")
    assert ref == "CCCTTGACCCTAACCCCTGACCCTGACCCTTAACCCTAACCCCTAACCCTTAACCCTTAAACCTTAACCCTCATCCTCACCCTCACCCTCAC"


def test_reference_writer_agrees_on_its_own_vectors():
    """In the build container: the unmodified reference writer on the same five calls writes the same lines."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(gu.GOLDEN_DIR), "..", "oracle"))
    import ref_harness as rh
    if not rh.reference_available():
        pytest.skip("reference sources not present")
    ref = rh.load_reference()
    from sniffles import vcf as ref_vcf
    vectors = [("atCga", 0, dict(svtype="INS", ref="N", alt="TAG", pos=3, svlen=3, end=3)),
               ("aTCGa", 0, dict(svtype="DEL", ref="N", alt="<DEL>", pos=2, svlen=-2, end=4)),
               ("T" * 50, 0, dict(svtype="INS", ref="N", alt="<INS>", pos=2, svlen=20, end=22)),
               ("ACGTRYKMNN" * 5, 0, dict(svtype="BND", ref="N", alt="N[chrY:77[", pos=5, svlen=0, end=5,
                                          info={"CHR2": "chrY", "STDEV_POS": 1.25, "FLAG": True, "NONE": None},
                                          genotypes={0: (0, 1, 33, 4, 5, ("1", 777))})),
               ("ACGTRYKMNN" * 5, 0, dict(svtype="DEL", ref="N", alt="<DEL>", pos=0, svlen=-30, end=30, qual=99, precise=False,
                                          svlens=[-30, -31], genotypes={0: (1, 1, 5, 0, 9, (None, None))}))]
    for seq, origin, kw in vectors:
        cfg = ref_test_config()
        cfg.sample_ids_vcf = [(0, "S")] if "genotypes" in kw else []
        out = []
        for cls, module in ((ref.sv.SVCall, ref_vcf), (sv.SVCall, vcf)):
            d = dict(contig="chr1", id="unittest-1", qual=10, filter="PASS", info={}, genotypes={}, precise=True, support=100,
                     rnames=["ut"], postprocess=None, qc=True, nm=-1, fwd=1, rev=1)
            d.update({k: (dict(v) if isinstance(v, dict) else v) for k, v in kw.items()})
            buf = io.StringIO()
            w = module.VCF(cfg, buf)
            w.reference_handle = Fetch(seq, origin)
            w.write_call(cls(**d))
            out.append(buf.getvalue())
        assert out[0] == out[1] and out[0]


def test_formatters_agree_with_the_reference_on_random_values():
    """format_genotype / format_info against the unmodified reference functions on seeded random inputs (build container)."""
    import os
    import random
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(gu.GOLDEN_DIR), "..", "oracle"))
    import ref_harness as rh
    if not rh.reference_available():
        pytest.skip("reference sources not present")
    rh.load_reference()
    from sniffles import vcf as ref_vcf
    rnd = random.Random(7)
    alleles = [0, 1, "."]
    phases = [None, (None, None), ("1", 5), ("2", "NULL"), ("1", None), (None, 17), "1", ("2", "7")]
    for _ in range(400):
        gt = (rnd.choice(alleles), rnd.choice(alleles), rnd.randint(0, 60), rnd.randint(0, 80), rnd.randint(0, 80), rnd.choice(phases))
        if rnd.random() < 0.5:
            gt = gt + (rnd.choice(["NULL", "Sniffles2.INS.1S0", "a,b"]),)
        for phased in (True, False):
            def outcome(fn):
                try:
                    return fn(gt, phased)
                except Exception as e:       # e.g. a one-character string as phase: both raise ValueError
                    return type(e).__name__
            assert outcome(vcf.format_genotype) == outcome(ref_vcf.format_genotype), (gt, phased)
    import numpy as np
    values = [0, 1, -3, 0.5, 1 / 3, 1e-9, 12345.6789, None, True, False, "x", "", ["a", "b"], [], np.float64(2.5), np.int64(4), float("nan")]
    for v in values:
        assert vcf.format_info("K", v) == ref_vcf.format_info("K", v), v


def test_three_decimals_formatter_equals_python_formatting():
    """The C extension prints f"{v:.3f}" values (STDEV_*, VAF, NM) with its own conversion (one multiplication, the rounding error from
    an fma) instead of printf: it has to give the exactly rounded decimal like Python does - ties of the binary value included."""
    import numpy as np
    from sniffles_amd import sv
    fast = sv._load_fast()
    if fast is None:
        pytest.skip("C extension not built")
    rng = np.random.default_rng(7)
    vals = list(rng.random(50000) * 10.0 ** rng.integers(-6, 13, 50000)) + list(-rng.random(5000) * 10.0 ** rng.integers(-6, 6, 5000))
    vals += [k / 8 for k in range(-2000, 2000)] + [k / 2048 for k in range(5000)] + [k / 16 + 0.0005 for k in range(100)]
    vals += [0.0, -0.0, 0.0005, 0.0015, 0.0025, 1e-9, -1e-9, 0.9995, 999999999999.9995, 1e12, 1e12 - 0.0005, 1e15, 1e22, 1.7976931348623157e308,
             5e-324, float("nan"), float("inf"), -float("inf"), 0.5, 1.0005, 2.0005, 1024.0005, 123456.7895, 123456.7885]
    for k in range(1, 20000, 7):          # the neighbours of every kind of decimal tie
        t = k / 1000 + 0.0005
        vals += [t, float(np.nextafter(t, 0)), float(np.nextafter(t, 1e9))]
    a = np.asarray(vals, np.float64)
    got = fast.format_f3(a.tobytes()).decode().split("\n")[:-1]
    assert len(got) == len(a)
    assert [g for v, g in zip(a.tolist(), got) if g != f"{v:.3f}"] == []
