"""The product's HIP sources - all four translation units, every kernel, the gfx950 wave / workgroup kernels and the
lock-step x_big kernels included - compiled UNCHANGED with g++ against a stand-in for the HIP runtime (tests/emu/simt:
fibres per thread, cross-lane operations, DPP, barriers, lock step by page tracking) and run in the GPU-less container
against the reference's goldens and the oracle.

This is the only host tier (a second one, serial-loop halves behind `#ifdef SNF_EMU` inside the product sources, was retired in
round 3): it runs the code the GPU runs - the launch sequence of snf_lib.hip with `wave_path` on, the fused scan chains, the
workgroup consensus kernels, the wave form of the extraction kernels; tests/test_emu_parity.py runs the fallback forms
(SNF_NO_WAVE / SNF_NO_FUSE) through the same library.  What it cannot show is anything about timing, memory ordering between workgroups, or the compiler - the parity
tests proper remain the `-m gpu` tests."""
import collections
import ctypes as C

import numpy as np
import pytest

import cases
import golden_util as gu
from sniffles_amd import lib, records, synth
from sniffles_amd.config import SnifflesConfig


@pytest.fixture
def simt():
    from emu import simt as S
    S.lib()
    return S


def run(L, cfg, tis, fin):
    with lib.Batch(cfg, tis) as b:
        b.call_candidates()
        if fin:
            b.finalize()
        return b.fetch(1 if fin else 0)


# ---------------------------------------------------------------------------------------------- the shim itself
def test_shim_models_the_cross_lane_operations(simt):
    """Known answers for the model: DPP row shifts / broadcasts (as wave scans), shuffles, ballots with only part of the wave
    taking part, barriers with waves that have already returned, and lock step (64 lanes incrementing one word = +1)."""
    L = simt.lib()
    out = (C.c_longlong * 16)()
    L.simt_selftest.argtypes = [C.POINTER(C.c_longlong)]
    L.simt_selftest.restype = C.c_int
    assert L.simt_selftest(out) == 0, list(out)
    assert list(out[:8]) == [0] * 8, list(out)           # mismatches per check
    assert out[8] == 1 and out[9] == 64                  # one word incremented by 64 lanes: lock step 1, fibres 64


def test_exact_division_matches_the_host_compilers(simt):
    """snf_exact.h::udivmod128_64 (double-precision estimates, exact 128-bit remainders) against __int128 division."""
    L = simt.lib()
    L.snf_simt_divcheck.argtypes = [C.c_long, C.c_ulonglong]
    L.snf_simt_divcheck.restype = C.c_long
    assert L.snf_simt_divcheck(3_000_000, 11) == 0


# ---------------------------------------------------------------------------------------------- clustering + calling
@pytest.mark.parametrize("name", sorted(cases.ALL))
def test_wave_path_matches_reference_golden(name, simt):
    build, kw, _ = cases.ALL[name]
    doc = gu.load(name)
    ti = build()
    cfg = gu.make_config(kw, ti)
    exp = doc["expected"]
    before = simt.counters()
    for stage, key, fin in (("cand", "candidates", False), ("final", "final", True)):
        res = run(simt.lib(), cfg, [ti], fin)
        got = records.records(res, [ti], stage)[0]
        if "error" in exp:
            assert got == {"error": exp["error"]}
            continue
        assert gu.diff_records(got, exp[key]) == []
        assert float(res.coverage_average_total[0]) == exp["coverage_average_total"]
    after = simt.counters()
    assert after["unmodelled"] == before["unmodelled"] and after["lockstep_conflicts"] == before["lockstep_conflicts"]


def test_wave_and_lockstep_kernels_really_ran(simt):
    """Guards the tier against silently testing nothing: a deep fuzz task must go through cross-lane operations, workgroup
    barriers and the lock-step kernels, and without lock step the same task must come out wrong."""
    import os
    ti = synth.gen_fuzz(903, task_id=0)
    cfg = SnifflesConfig()
    before = simt.counters()
    good = records.records(run(simt.lib(), cfg, [ti], True), [ti], "final")
    after = simt.counters()
    assert after["wave_ops"] - before["wave_ops"] > 1000 and after["block_syncs"] > before["block_syncs"]
    assert after["lockstep_merges"] - before["lockstep_merges"] > 10 and after["lockstep_faults"] > before["lockstep_faults"]
    os.environ["SNF_SIMT_NO_LOCKSTEP"] = "1"
    try:
        bad = records.records(run(simt.lib(), cfg, [ti], True), [ti], "final")
    finally:
        del os.environ["SNF_SIMT_NO_LOCKSTEP"]
    assert simt.counters()["unmodelled"] > after["unmodelled"] and bad != good


KW = [{}, dict(mosaic=True), dict(repeat=True, mosaic=True), dict(no_qc=True, phase=True), dict(minsupport=3, long_ins_length=200),
      dict(dev_no_resplit=True), dict(cluster_merge_pos=300, cluster_binsize=50)]


@pytest.mark.parametrize("ci", range(len(KW)))
def test_wave_path_matches_oracle_on_fuzz_batches(ci, simt, oracle_mod):
    tis = [synth.gen_fuzz(1000 * ci + 17 * k, task_id=k) for k in range(4)]
    cfg = SnifflesConfig(**KW[ci])
    before = simt.counters()
    got = records.records(run(simt.lib(), cfg, tis, True), tis, "final")
    assert got == records.records(oracle_mod.run(cfg, tis, True), tis, "final")
    assert simt.counters()["unmodelled"] == before["unmodelled"]


@pytest.mark.parametrize("env", [{"SNF_NO_BIG_STAGE": "1"}, {"SNF_E1_BATCH": "32"}, {"SNF_E1_BATCH": "2"}, {"SNF_NO_POOL_SLICES": "1"},
                                 {"SNF_NO_FUSE": "1"}, {"SNF_SORT64": "1"}, {"SNF_RUN_GAP": "0"}, {"SNF_CONS_NW": "1"},
                                 {"SNF_CONS_LARGE_NW": "8"}, {"SNF_NO_WAVE": "1"}])
def test_library_switches_keep_the_results(env, simt, oracle_mod, monkeypatch):
    """The scheduling / layout switches of the library (README) change how the kernels run, never what they return."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    tis = [synth.gen_fuzz(4242 + k, task_id=k) for k in range(3)] + [synth.gen_task(3, "chrS", 150_000, 40.0, seed=5)]
    cfg = SnifflesConfig()
    assert records.records(run(simt.lib(), cfg, tis, True), tis, "final") == records.records(oracle_mod.run(cfg, tis, True), tis, "final")


@pytest.mark.parametrize("fill", ["0x00", "0xff", "0x7f"])
def test_results_do_not_depend_on_fresh_device_memory(fill, simt, oracle_mod, monkeypatch):
    """hipMalloc returns whatever was there: the stand-in fills new blocks with a byte of the test's choosing (0xA5 by default)."""
    monkeypatch.setenv("SNF_SIMT_FILL", fill)
    tis = [synth.gen_fuzz(77 + k, task_id=k) for k in range(3)] + [synth.gen_task(3, "chrS", 200_000, 60.0, seed=8, mosaic_frac=0.3)]
    for kw in ({}, dict(mosaic=True, qc_nm=True)):
        cfg = SnifflesConfig(**kw)
        assert records.records(run(simt.lib(), cfg, tis, True), tis, "final") == records.records(oracle_mod.run(cfg, tis, True), tis, "final")


@pytest.mark.parametrize("order", ["reverse", "random:3", "random:11"])
def test_results_do_not_depend_on_the_order_lanes_run_in(order, simt, oracle_mod, monkeypatch):
    """Between two synchronisation points the stand-in runs the lanes of a wave one after the other, by default in ascending
    order; a kernel whose result changes with that order has a race no barrier covers."""
    monkeypatch.setenv("SNF_SIMT_ORDER", order)
    tis = [synth.gen_fuzz(177 + k, task_id=k) for k in range(3)] + [synth.gen_task(3, "chrS", 200_000, 60.0, seed=9, mosaic_frac=0.3)]
    cfg = SnifflesConfig(mosaic=True)
    assert records.records(run(simt.lib(), cfg, tis, True), tis, "final") == records.records(oracle_mod.run(cfg, tis, True), tis, "final")
    probs = _cons_problems()[:60]
    got, cls, _ = simt.consensus_batch([(p["best"], p["others"], p["skip"]) for p in probs], 6)
    assert got == [p["expected"] for p in probs]


def test_no_undefined_behaviour_in_the_kernels(simt, oracle_mod, capfd):
    """The same sources with -fsanitize=undefined,bounds-strict: a shift by 64, a signed overflow or an index beyond a
    statically sized (LDS) array gives the x86 answer here and possibly another one on the GPU - the sanitizer reports it."""
    L = simt.lib_sanitized()
    tis = [synth.gen_fuzz(277 + k, task_id=k) for k in range(3)] + [synth.gen_task(3, "chrS", 200_000, 90.0, seed=10, mosaic_frac=0.3)]
    for kw in ({}, dict(mosaic=True, qc_nm=True, repeat=True)):
        cfg = SnifflesConfig(**kw)
        assert records.records(run(L, cfg, tis, True), tis, "final") == records.records(oracle_mod.run(cfg, tis, True), tis, "final")
    probs = _cons_problems()
    for kw in (dict(), dict(mode=4), dict(nw=1), dict(nw=8)):
        got, _, _ = simt.consensus_batch([(p["best"], p["others"], p["skip"]) for p in probs], 6, **kw)
        assert got == [p["expected"] for p in probs]
    import emu.emu as E
    E.lib()                                              # the host tier becomes the library of this test
    import test_combine as TC
    import test_edit_distance as TE
    import test_extract as TX
    orig = E.lib
    E.lib = lambda: L
    try:
        TC.test_emulated_combine_matches_reference(TC.NAMES[-1])
        TE.test_edit_distance_banded_emulated(oracle_mod)
        TX.test_kernel_bodies_match_reference(sorted(cases.EXTRACT)[0], L)
    finally:
        E.lib = orig
    err = capfd.readouterr().err
    assert "runtime error" not in err, err[-3000:]


def _with_reads(ti, n_reads, seed, max_len=1500):
    """`ti` with `n_reads` more alignment records (read table only: coverage and REF haplotype counts), starts kept sorted."""
    rng = np.random.default_rng([seed, 991])
    L = int(ti.contig_len)
    rs = rng.integers(0, max(1, L - 10), n_reads)
    re_ = np.minimum(L, rs + rng.integers(50, max_len, n_reads))
    hp = rng.integers(0, 3, n_reads).astype(np.uint8)
    start = np.concatenate([ti.read_start.astype(np.int64), rs]); end = np.concatenate([ti.read_end.astype(np.int64), re_])
    hps = np.concatenate([ti.read_hp, hp])
    o = np.argsort(start, kind="stable")
    ti.read_start, ti.read_end, ti.read_hp = (np.ascontiguousarray(start[o], np.int32), np.ascontiguousarray(end[o], np.int32),
                                              np.ascontiguousarray(hps[o], np.uint8))
    return ti


@pytest.mark.parametrize("d4", [None, "thread"])
def test_coverage_samples_over_large_read_tables(d4, simt, oracle_mod, monkeypatch):
    """The five coverage samples of every call (d4s_coverage: a thread per sample, rank queries as a 16-ary descent over sampled
    levels of the GLOBAL read arrays, snf_exact.h::rank_upper_16ary) on read tables that reach every level: task segments that start
    at unaligned offsets, cross 4096- and 65536-entry boundaries, hold fewer than 16 reads, end in a partial node; runs of equal
    starts; samples left and right of every read.  SNF_D4=thread: the former thread-per-call kernel stays selectable."""
    if d4:
        monkeypatch.setenv("SNF_D4", d4)
    sizes = (3, 4090, 9, 70_001, 12_345, 17)
    tis = [_with_reads(synth.gen_fuzz(900 + k, task_id=k, n_leads=300), n, k) for k, n in enumerate(sizes)]
    tis[3].read_start[1000:1400] = tis[3].read_start[1000]            # a run of equal starts across node borders
    tis[3].read_end[2000:2300] = int(tis[3].read_start[2299]) + 100   # ... and of equal ends
    cfg = SnifflesConfig()
    got = records.records(run(simt.lib(), cfg, tis, True), tis, "final")
    assert got == records.records(oracle_mod.run(cfg, tis, True), tis, "final")
    assert sum(len(g) for g in got) > 50


def test_random_option_sets(simt, oracle_mod):
    """tools/dev/cfgfuzz.py: random combinations of some sixty hot-path options (filters, cluster / merge widths, mosaic and
    developer switches), three adversarial tasks each.  oracle/ref_cfgfuzz.py holds the oracle against the unmodified
    reference over the same option space (build container only)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("cfgfuzz", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "dev", "cfgfuzz.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    bad, calls = m.sweep(simt.lib(), 16, seed0=1000)
    assert bad == 0 and calls > 1000


def test_very_deep_clusters_in_lock_step(simt, oracle_mod):
    """150x: most clusters have more than 64 leads (x_big, cooperative rank sorts, LDS rows)."""
    ti = synth.gen_task(0, "chrD", 60_000, 150.0, seed=3)
    cfg = SnifflesConfig()
    assert records.records(run(simt.lib(), cfg, [ti], True), [ti], "final") == records.records(oracle_mod.run(cfg, [ti], True), [ti], "final")


# ---------------------------------------------------------------------------------------------- consensus instances
def _cons_problems():
    return gu.load("consensus_novel_from_reads")["problems"]


@pytest.mark.parametrize("kw", [dict(), dict(mode=2), dict(mode=4), dict(nw=1), dict(nw=8), dict(grid_cap=3), dict(mode=4, grid_cap=5),
                                dict(min_reads=5)], ids=str)
def test_consensus_instances_match_reference_vectors(kw, simt):
    """Every template instance of the workgroup consensus kernel the library launches (SMALL 4 waves / 1 wave, LARGE 4 / 8
    waves, ROWS), SMALL calls through LARGE, everything through ROWS, strided grids, the verbatim-copy kernel - against
    the 160 vectors of the unmodified reference function."""
    probs = _cons_problems()
    got, cls, handed = simt.consensus_batch([(p["best"], p["others"], p["skip"]) for p in probs], 6, **kw)
    exp = [p["expected"] if c else p["best"] for p, c in zip(probs, cls)]
    assert [i for i, (g, e) in enumerate(zip(got, exp)) if g != e] == []
    n = collections.Counter(cls)
    if kw.get("mode") == 4:
        assert n[4] == len(probs)
    elif kw.get("mode") == 2:
        assert n[2] == len(probs)
    elif not kw.get("min_reads"):
        assert n[1] >= 50 and n[2] >= 50


@pytest.mark.parametrize("kw", [dict(), dict(mode=2), dict(mode=4), dict(nw=8), dict(grid_cap=2)], ids=str)
def test_consensus_long_segments_match_reference_vectors(kw, simt):
    """Copied segments far longer than a lane walks on its own (SNF_CONS_LONGSEG: the whole wave compares and votes them),
    segments that leave the shift window and return, odd characters inside them - 60 vectors of the unmodified reference
    function (oracle/make_golden.py::main_consensus_long)."""
    probs = gu.load("consensus_long_segments")["problems"]
    got, cls, handed = simt.consensus_batch([(p["best"], p["others"], p["skip"]) for p in probs], 6, **kw)
    assert all(cls)
    assert [i for i, (g, p) in enumerate(zip(got, probs)) if g != p["expected"]] == []
    assert sum(p["expected"] != p["best"] for p in probs) >= 20


def test_consensus_escape_list_overflow_is_handed_to_rows(simt):
    """More non-ACGT votes than the escape list of the LDS-vote instances holds: the call is redone by ROWS (work list 7)
    and still equals the plain rule - checked against the ROWS instance alone, which the vectors above pin."""
    rng = np.random.default_rng(5)
    probs = []
    for k in range(6):
        L = 300 if k < 3 else 2000
        truth = rng.choice(list(b"ACGT"), L).astype(np.uint8)
        best = truth.copy()
        err = rng.random(L) < 0.03                                       # the best read carries errors the others outvote
        best[err] = rng.choice(list(b"ACGT"), int(err.sum()))
        others = []
        for _ in range(12):
            o = truth.copy()
            hit = rng.random(L) < 0.15
            o[hit] = rng.choice(list(b"NRYKM"), int(hit.sum()))       # odd characters everywhere
            others.append(bytes(o))
        probs.append((bytes(best), others, 3 + L // 100))
    got, cls, handed = simt.consensus_batch(probs, 6)
    ref, _, _ = simt.consensus_batch(probs, 6, mode=4)
    assert handed == len(probs) and got == ref
    assert any(g != p[0].decode("latin-1") for g, p in zip(got, probs))           # the vote really changes bases


def test_consensus_entry_point_of_the_library(simt, monkeypatch):
    """snf_consensus_batch itself (seam B4) through sniffles_amd.consensus."""
    from sniffles_amd import consensus
    monkeypatch.setattr(consensus._lib, "load", simt.lib)
    probs = _cons_problems()
    got = consensus.novel_from_reads_batch([(p["best"], p["others"], p["skip"]) for p in probs], klen=6)
    assert [i for i, (g, p) in enumerate(zip(got, probs)) if g != p["expected"]] == []


# (The combine / edit-distance / CombineTask / genotype / extraction / BAM -> VCF / drop-in tests live in their own modules and
# run on this same library since the serial-emulation tier was retired: tests/emu/emu.py hands out this tier.)


def test_call_variances_on_both_sides_of_the_small_spread_path(simt, oracle_mod):
    """util.stdev(util.trim(...)) in the call kernels takes two 32-bit sums and one fp64 division when a cluster's (trimmed) values lie
    within 8191 of each other, the exact 128-bit form otherwise (snf_wave_call.h::wave_stdev_trim_sorted, the grouped form in
    snf_wave_call_g.h decides for the whole wave): clusters of 5, 8, 9, 30 and 64 leads whose lengths / positions spread by 0 ... 17 000
    (trimmed: on both sides of 8192) and 3e6, DEL (svlen and ref_start move together) and INV (position only), against the oracle - every fp64 bit."""
    from sniffles_amd import lib as plib, records
    leads, reads = [], []
    base = 1_000_000
    k = 0
    for n in (5, 8, 9, 30, 64):
        for spread in (0, 8191, 8192, 16380, 16384, 16390, 17000, 3_000_000):      # (a quarter is trimmed on either side: ~half the spread is what the sums see)
            for svt in ("DEL", "INV"):
                pos = base + k * 9_000_000
                k += 1
                for i in range(n):
                    step = 0 if n == 1 else (spread * i) // (n - 1)
                    if svt == "DEL":      # same end point: the deletions start `step` earlier and are that much longer
                        leads.append(dict(svtype="DEL", ref_start=pos + 50 + (i % 3), svlen=-(40_000 + step), read=f"c{k}_{i}", strand="+-"[i % 2], source="SPLIT_SUP"))
                    else:
                        leads.append(dict(svtype="INV", ref_start=pos + (i % 3), svlen=20_000 + step, read=f"c{k}_{i}", strand="+-"[i % 2], source="SPLIT_SUP"))
                reads += [(pos - 60_000, pos + 60_000, 0)] * 12
    ti = cases.mk_task(leads, reads, base + (k + 1) * 9_000_000)
    for kw in (dict(minsupport=2), dict(minsupport=2, repeat=True), dict(minsupport=2, phase=False, cluster_merge_len=50.0)):
        cfg = SnifflesConfig(**kw)
        got = records.records(run(simt, cfg, [ti], True), [ti], "final")
        exp = records.records(oracle_mod.run(cfg, [ti], True), [ti], "final")
        assert got == exp
        assert len(exp[0]) >= 30
        sd = sorted(float.fromhex(c["stdev_len"]) if isinstance(c["stdev_len"], str) else float(c["stdev_len"]) for c in exp[0] if c.get("stdev_len") is not None)
        assert sd[0] < 10.0 and sd[-1] > 100_000.0      # both sides were reached


def test_fused_sequences_of_many_parts(simt, oracle_mod):
    """merge_inner's fused sequences (cluster.py:85-122) in clusters that fuse many reads at once (wave_copy_parts takes the parts four at
    a time): clusters of 24-31 reads with two and three pieces each (48-62 leads), pieces of 1 .. 2100 bases (odd tails, more than 1 KB),
    one piece without a sequence; a small cluster beside them.  The ALT sequences (consensus over the fused reads, or the best fused read
    verbatim) come out of those bytes: every record against the oracle."""
    from sniffles_amd import records
    rng = np.random.default_rng(5)
    leads, reads = [], []
    for c, (n_reads, pieces, plen) in enumerate([(24, 2, 180), (31, 2, 1100), (16, 3, 37), (4, 2, 2100), (20, 3, 1)]):
        pos = 20_000 + 40_000 * c
        allele = cases._rng_seq(rng, pieces * plen)
        for r in range(n_reads):
            q = 3000 + 11 * r
            for k in range(pieces):
                seq = cases._mutate(rng, allele[k * plen:(k + 1) * plen], 0.02)[:plen] or "A"
                if c == 0 and r == 5 and k == 1:
                    seq = None
                leads.append(dict(svtype="INS", ref_start=pos + 20 * k + (r % 3), svlen=plen, read=f"m{c}_{r}", qry_start=q + k * (plen + 40),
                                  qry_end=q + k * (plen + 40) + plen, seq=seq, strand="+-"[r % 2]))
        reads += [(pos - 5000, pos + 5000, 0)] * (n_reads + 6)
    ti = cases.mk_task(leads, reads, 260_000)
    for kw in (dict(minsupport=2), dict(minsupport=2, no_consensus=True)):
        cfg = SnifflesConfig(**kw)
        got = records.records(run(simt, cfg, [ti], True), [ti], "final")
        exp = records.records(oracle_mod.run(cfg, [ti], True), [ti], "final")
        assert got == exp
        assert sum(1 for c in exp[0] if c["svtype"] == "INS" and c.get("alt")) >= 4
