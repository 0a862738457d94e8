"""Parity tests proper: the gfx950 library (through the C-ABI) against the reference goldens, the
oracle on seeded inputs, and size-independent properties at larger sizes.  Run with -m gpu."""
import numpy as np
import pytest

import cases
import golden_util as gu
from sniffles_amd import lib, records, synth
from sniffles_amd.config import SnifflesConfig

pytestmark = pytest.mark.gpu


def run(cfg, tis, fin=True):
    with lib.Batch(cfg, tis) as b:
        b.call_candidates()
        if fin:
            b.finalize()
        return b.fetch(1 if fin else 0)


def test_native_library_loaded_and_device_present():
    assert lib.device_count() >= 1
    from sniffles_amd import abi
    assert lib.load().snf_abi_version() == abi.ABI_VERSION


@pytest.mark.parametrize("name", sorted(cases.ALL))
def test_hip_matches_reference_golden(name):
    """Bit-exact on every integer field, float statistics and the INS consensus sequence (edit-distance
    tolerance 0) against the unmodified reference's outputs."""
    build, kw, _ = cases.ALL[name]
    doc = gu.load(name)
    ti = build()
    assert gu.input_sha(ti) == doc["input_sha"]
    cfg = gu.make_config(kw, ti)
    exp = doc["expected"]
    for stage, key, fin in (("cand", "candidates", False), ("final", "final", True)):
        res = run(cfg, [ti], fin)
        got = records.records(res, [ti], stage)[0]
        if "error" in exp:
            assert got == {"error": exp["error"]}
            continue
        assert gu.diff_records(got, exp[key]) == []
        assert float(res.coverage_average_total[0]) == exp["coverage_average_total"]


KW = [{}, dict(mosaic=True), dict(minsupport="auto", qc_nm=True), dict(no_qc=True),
      dict(qc_strand=True, minsvlen="50", cluster_merge_pos=50), dict(repeat=True, mosaic=True, mosaic_include_germline=True)]


@pytest.mark.parametrize("ci", range(len(KW)))
def test_hip_matches_oracle_on_fuzz_batches(ci, oracle_mod):
    cfg = SnifflesConfig(**KW[ci])
    for seed in range(0, 60, 4):
        tis = [synth.gen_fuzz(seed + k, task_id=k) for k in range(4)]
        exp = oracle_mod.run(cfg, tis, True)
        got = run(cfg, tis, True)
        assert records.records(got, tis, "final") == records.records(exp, tis, "final")
        assert np.array_equal(got.coverage_average_total, exp.coverage_average_total, equal_nan=True)


@pytest.mark.parametrize("gap", ["0", "150", "-1"])
def test_hip_run_cuts_exact(gap, oracle_mod, monkeypatch):
    monkeypatch.setenv("SNF_RUN_GAP", gap)
    cfg = SnifflesConfig(repeat=True)
    tis = [synth.gen_fuzz(500 + k, task_id=k) for k in range(6)]
    assert records.records(run(cfg, tis), tis, "final") == records.records(oracle_mod.run(cfg, tis, True), tis, "final")


def test_hip_wide_sort_keys_exact(oracle_mod, monkeypatch):
    """Batches whose key space exceeds 32 bits sort 64-bit keys; SNF_SORT64 forces that path on a small batch."""
    monkeypatch.setenv("SNF_SORT64", "1")
    cfg = SnifflesConfig()
    tis = [synth.gen_fuzz(700 + k, task_id=k) for k in range(5)]
    exp = oracle_mod.run(cfg, tis, True)
    got = run(cfg, tis, True)
    assert records.records(got, tis, "final") == records.records(exp, tis, "final")
    assert np.array_equal(got.coverage_average_total, exp.coverage_average_total, equal_nan=True)


def test_hip_plain_scan_chains_exact(oracle_mod, monkeypatch):
    """Batches above 2^25 leads use device-wide scans instead of the fused flag/scan/emit kernel pairs; SNF_NO_FUSE
    forces that path on a small batch."""
    monkeypatch.setenv("SNF_NO_FUSE", "1")
    cfg = SnifflesConfig()
    tis = [synth.gen_fuzz(800 + k, task_id=k) for k in range(5)]
    exp = oracle_mod.run(cfg, tis, True)
    got = run(cfg, tis, True)
    assert records.records(got, tis, "final") == records.records(exp, tis, "final")
    assert np.array_equal(got.coverage_average_total, exp.coverage_average_total, equal_nan=True)


def test_hip_ragged_batch_with_empty_tasks(oracle_mod):
    """Tasks without leads, without reads, and an entirely empty batch next to ordinary tasks: every stage has to cope
    with empty ranges (fused scan chains, wave kernels, result block)."""
    cfg = SnifflesConfig()
    empty = cases.mk_task([], [], 50_000, task_id=0, contig="chrE")
    no_leads = cases.mk_task([], [(1000, 30_000, 0), (2000, 60_000, 1)], 80_000, task_id=1, contig="chrR")
    normal = synth.gen_fuzz(950, task_id=2)
    for tis in ([empty], [no_leads], [empty, normal, no_leads], [normal, empty]):
        tis = [t for t in tis]
        exp = oracle_mod.run(cfg, tis, True)
        got = run(cfg, tis, True)
        assert records.records(got, tis, "final") == records.records(exp, tis, "final")
        assert np.array_equal(got.coverage_average_total, exp.coverage_average_total, equal_nan=True)


def test_batches_in_flight_on_host_threads(oracle_mod):
    """bench.py keeps several batch handles in flight from host threads (own streams each): concurrent passes must
    give exactly the results of passes run one after the other."""
    import threading
    cfg = SnifflesConfig()
    sets = [[synth.gen_fuzz(900 + 10 * w + k, task_id=k) for k in range(4)] for w in range(3)]
    exp = [records.records(oracle_mod.run(cfg, tis, True), tis, "final") for tis in sets]
    got = [None] * 3
    errs = []

    def worker(w):
        try:
            with lib.Batch(cfg, sets[w]) as b:
                for _ in range(5):
                    b.call_candidates(); b.finalize(); r = b.fetch(1)
                got[w] = records.records(r, sets[w], "final")
        except BaseException as e:  # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=worker, args=(w,)) for w in range(3)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not errs, errs
    assert got == exp


def genome(scale, cov=30, seed=1, **kw):
    return synth.gen_genome(coverage=cov, seed=seed, scale=scale, **kw)


def test_hip_matches_oracle_on_scaled_genome(oracle_mod):
    """24 contigs at 1 % of GRCh38 lengths, 30x ONT-like: every call of every contig identical."""
    tis = genome(0.01)
    cfg = SnifflesConfig()
    exp = oracle_mod.run(cfg, tis, True)
    got = run(cfg, tis, True)
    assert records.records(got, tis, "final") == records.records(exp, tis, "final")


def test_hip_matches_oracle_full_size_contigs(oracle_mod):
    """chr20-chr22 at their full GRCh38 lengths, 30x (BASELINE configs[0] shape and two more contigs) in one batch:
    every record of every call identical to the oracle, coverage averages bit-equal."""
    tis = [synth.gen_task(i, c, synth.GRCH38[c], 30, 1) for i, c in enumerate(["chr20", "chr21", "chr22"])]
    cfg = SnifflesConfig()
    exp = oracle_mod.run(cfg, tis, True)
    got = run(cfg, tis, True)
    assert records.records(got, tis, "final") == records.records(exp, tis, "final")
    assert np.array_equal(got.coverage_average_total, exp.coverage_average_total, equal_nan=True)


def test_hip_matches_oracle_on_the_bench_workload(oracle_mod):
    """The exact workload of bench.py's headline (BASELINE configs[1]: 24 contigs at full GRCh38 size, 30x, replica 0) in
    ONE batch - the only place where the 31-bit lead keys and the 24-task read-end keys run at full width - against the
    oracle, task by task: every field of every call, ALT bytes, supporting reads, coverage averages."""
    tis = [synth.gen_task(ci, c, synth.GRCH38[c], 30.0, 1) for ci, c in enumerate(synth.CONTIGS)]
    cfg = SnifflesConfig()
    got = run(cfg, tis, True)
    n = 0
    for t, ti in enumerate(tis):
        exp = oracle_mod.run(cfg, [ti], True)
        assert records.diff_results(got, t, exp, 0) == [], ti.contig
        n += len(exp.calls)
    assert n == len(got.calls) and n > 80000


def test_hip_matches_oracle_hifi_60x_and_mosaic(oracle_mod):
    tis = genome(0.004, cov=60, seed=2, err=0.005)
    cfg = SnifflesConfig()
    assert records.records(run(cfg, tis), tis, "final") == records.records(oracle_mod.run(cfg, tis, True), tis, "final")
    tis = genome(0.004, cov=30, seed=3, mosaic_frac=0.05)
    cfg = SnifflesConfig(mosaic=True)
    assert records.records(run(cfg, tis), tis, "final") == records.records(oracle_mod.run(cfg, tis, True), tis, "final")


def test_full_size_properties():
    """At chr20 full size (BASELINE configs[0] shape): tasks are independent, so (a) a batch equals the
    concatenation of single-task runs, (b) repeating a pass is idempotent, (c) calls are in candidate
    order (svtype-major, sv_id consecutive), (d) every INS ALT has the length of a supporting read."""
    t20 = synth.gen_task(19, "chr20", synth.GRCH38["chr20"], 30, 1)
    t21 = synth.gen_task(20, "chr21", synth.GRCH38["chr21"] // 4, 30, 1)
    cfg = SnifflesConfig()
    both = run(cfg, [t20, t21])
    single = [run(cfg, [t20]), run(cfg, [t21])]
    rb = records.records(both, [t20, t21], "final")
    assert rb[0] == records.records(single[0], [t20], "final")[0]
    assert rb[1] == records.records(single[1], [t21], "final")[0]
    with lib.Batch(cfg, [t20, t21]) as b:
        b.call_candidates(); b.finalize(); r1 = b.fetch(1)
        b.call_candidates(); b.finalize(); r2 = b.fetch(1)
    assert r1.calls.tobytes() == r2.calls.tobytes() and r1.alt_pool.tobytes() == r2.alt_pool.tobytes()
    c = both.calls
    for t in range(2):
        lo, hi = int(both.task_call_off[t]), int(both.task_call_off[t + 1])
        assert np.all(np.diff(c["svtype"][lo:hi]) >= 0)
        assert np.array_equal(c["sv_id"][lo:hi], np.arange(hi - lo))
    ins = (c["svtype"] == 0) & (c["alt_len"] >= 0)
    assert ins.sum() > 100 and np.all(c["alt_len"][ins] >= 45)
    assert (c["filter"] == 0).sum() > 400  # ~561 planted sites on chr20


@pytest.mark.parametrize("cov,no_stage", [(150, "0"), (700, "0"), (150, "1")])
def test_hip_matches_oracle_on_very_deep_clusters(oracle_mod, monkeypatch, cov, no_stage):
    """Clusters far beyond one wave (x_big<0/1/2>: the wave runs the serial bodies uniformly): 150x puts them into the LDS rows
    (<= 160 / 512 leads), 700x beyond the rows' capacity (global scratch rows), SNF_NO_BIG_STAGE=1 switches the LDS staging off
    altogether.  HiFi-like reads (low error: many leads per cluster survive), germline and mosaic settings."""
    monkeypatch.setenv("SNF_NO_BIG_STAGE", no_stage)
    tis = [synth.gen_task(0, "chr21", 400_000, cov, 5, err=0.005, read_len_mean=12000, site_density=2e-4),
           synth.gen_task(1, "chr22", 250_000, cov, 6, err=0.005, read_len_mean=12000, mosaic_frac=0.2, site_density=2e-4)]
    for cfg in (SnifflesConfig(), SnifflesConfig(mosaic=True)):
        exp = oracle_mod.run(cfg, tis, True)
        got = run(cfg, tis, True)
        assert max(int(c["n_leads"]) for c in got.calls) > (512 if cov >= 700 else 64)
        for t in range(len(tis)):
            assert records.diff_results(got, t, exp, t) == [], (cov, no_stage, t)
