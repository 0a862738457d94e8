"""The occupancy prefilter (snf_stage_cluster.h a0_*: leads alone in their (task, svtype, bin) cell are dropped in front of the
sort when dev_min_leads_cluster >= 2, cluster.py:262) changes nothing but `cluster_seed_index`, which becomes -1 ("not
provided"); SNF_NO_PREFILTER=1 and snf_batch_fetch_clusters give the reference's seed index."""
import numpy as np
import pytest

from sniffles_amd import cluster, lib, records, synth
from sniffles_amd.config import SnifflesConfig


def tasks():
    return [synth.gen_task(0, "chr20", 1_500_000, 30, 3), synth.gen_fuzz(11, task_id=1), synth.gen_fuzz(12, task_id=2)]


def run(L, cfg, tis):
    with lib.Batch(cfg, tis) as b:
        b.call_candidates()
        b.finalize()
        return b.fetch(1)


def check(L, oracle_mod, monkeypatch):
    tis = tasks()
    for kw in ({}, {"mosaic": True}, {"no_qc": True}):       # no_qc: dev_min_leads_cluster = 1 -> the prefilter stays off
        cfg = SnifflesConfig(**kw)
        exp = oracle_mod.run(cfg, tis, True)
        on = run(L, cfg, tis)
        monkeypatch.setenv("SNF_NO_PREFILTER", "1")
        off = run(L, cfg, tis)
        monkeypatch.delenv("SNF_NO_PREFILTER")
        for t in range(len(tis)):
            assert records.diff_results(on, t, exp, t) == []
            assert records.diff_results(off, t, exp, t) == []
        assert np.array_equal(off.calls["cluster_seed_index"], exp.calls["cluster_seed_index"])
        filtered = cfg.dev_min_leads_cluster >= 2
        assert bool((on.calls["cluster_seed_index"] == -1).all()) == filtered
        for f in on.calls.dtype.names:
            if f != "cluster_seed_index":
                a, b = on.calls[f], off.calls[f]
                assert np.array_equal(a, b) or (a.dtype.kind == "f" and np.array_equal(a, b, equal_nan=True)), f
        assert on.alt_pool.tobytes() == off.alt_pool.tobytes() and on.rnames.tobytes() == off.rnames.tobytes()


def test_prefilter_changes_nothing_but_the_seed_index_emu(oracle_mod, monkeypatch):
    import emu.emu as E
    check(E.lib(), oracle_mod, monkeypatch)


def test_prefilter_changes_nothing_but_the_seed_index_simt(oracle_mod, monkeypatch):
    from emu import simt as S
    check(S.lib(), oracle_mod, monkeypatch)


@pytest.mark.gpu
def test_prefilter_changes_nothing_but_the_seed_index_gpu(oracle_mod, monkeypatch):
    check(None, oracle_mod, monkeypatch)


def test_cluster_views_keep_their_ids_behind_the_prefilter():
    """`cluster.resolve` (seam B3) redoes the candidate stage unfiltered: ids (seed index included) are the same as without
    the prefilter, and the calls fetched afterwards are unchanged."""
    import emu.emu as E
    E.lib()                                              # the host tier becomes the library of this test
    ti = synth.gen_task(0, "chr20", 1_000_000, 30, 5)
    cfg = SnifflesConfig()
    with lib.Batch(cfg, [ti]) as b:
        b.call_candidates()
        b.finalize()
        before = b.fetch(1)
        cl = b.fetch_clusters(2)
        after = b.fetch(1)
    assert (before.calls["cluster_seed_index"] == -1).all() and (after.calls["cluster_seed_index"] >= 0).all()
    for f in before.calls.dtype.names:
        if f != "cluster_seed_index":
            assert np.array_equal(before.calls[f], after.calls[f], equal_nan=before.calls[f].dtype.kind == "f"), f
    assert int(cl["seed_index"].max()) > len(cl["seed_index"])      # counts singleton bins too


# ---- the window front end (snf_stage_window.h), the sort path it replaces, and the scan chains in both forms
FRONT_VARIANTS = [dict(SNF_CHAIN="0"), dict(SNF_CHAIN="1"), dict(SNF_NO_WINFRONT="1"), dict(SNF_WIN_BITS="6"), dict(SNF_WIN_BITS="8", SNF_CHAIN="0"),
                  dict(SNF_WIN_BITS_MAX="12"), dict(SNF_WIN_BITS="11", SNF_W4_SPLIT="1"), dict(SNF_WIN_BITS="11"),
                  dict(SNF_GRAPH="1", SNF_CHAIN="1"), dict(SNF_GRAPH="1", SNF_CHAIN="0")]


def check_front_variants(L, oracle_mod, monkeypatch, env, passes=3):
    """Window widths down to windows that need the 1024-lead instance and up to 12 bits, the sort path (SNF_NO_WINFRONT=1), the scan
    chains as single launches (decoupled look-back) and as launch pairs, the pass replayed from a HIP graph: the calls of the oracle,
    pass after pass over the same handle (the counters, cursors and chain tags a pass leaves behind are right for the next one)."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    tis = [synth.gen_task(0, "chr20", 2_500_000, 30, 11), synth.gen_fuzz(41, task_id=1), synth.gen_task(2, "chr21", 1_500_000, 60, 12, err=0.005),
           synth.gen_fuzz(42, task_id=3)]
    for kw in ({}, dict(mosaic=True)):
        cfg = SnifflesConfig(**kw)
        exp = records.records(oracle_mod.run(cfg, tis, True), tis, "final")
        with lib.Batch(cfg, tis) as b:
            for _ in range(passes):
                b.run_pass()
                assert records.records(b.fetch(1), tis, "final") == exp
            b.call_candidates(); b.finalize()                      # ... and the two calls on their own afterwards
            assert records.records(b.fetch(1), tis, "final") == exp


@pytest.mark.parametrize("env", FRONT_VARIANTS[:8])
def test_front_end_and_scan_chain_variants_are_exact_emu(oracle_mod, monkeypatch, env, capfd):
    import emu.emu as E
    monkeypatch.setenv("SNF_PROF", "1")
    check_front_variants(E.lib(), oracle_mod, monkeypatch, env, passes=2)
    err = capfd.readouterr().err
    if env.get("SNF_WIN_BITS") == "11":      # windows of more than 64 leads exist: w4s_segment's small instance + the large one over a list, or one launch
        assert ("w4s_segment in two launches" in err) == ("SNF_W4_SPLIT" in env), err[-600:]


@pytest.mark.gpu
@pytest.mark.parametrize("env", FRONT_VARIANTS)
def test_front_end_chain_and_graph_variants_are_exact_gpu(oracle_mod, monkeypatch, env):
    check_front_variants(None, oracle_mod, monkeypatch, env, passes=5)
