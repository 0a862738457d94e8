"""The C oracle must reproduce the reference's own outputs (tests/golden, generated from the
unmodified reference by oracle/make_golden.py) bit-for-bit: integer fields, float statistics,
filters, genotypes, phase strings and INS consensus sequences."""
import numpy as np
import pytest

import cases
import golden_util as gu
from sniffles_amd import records


@pytest.mark.parametrize("name", sorted(cases.ALL))
def test_oracle_matches_reference_golden(name, oracle_mod):
    build, kw, _ = cases.ALL[name]
    doc = gu.load(name)
    ti = build()
    assert gu.input_sha(ti) == doc["input_sha"], "seeded input drifted from the fixture"
    cfg = gu.make_config(kw, ti)
    exp = doc["expected"]
    for stage, key, fin in (("cand", "candidates", False), ("final", "final", True)):
        res = oracle_mod.run(cfg, [ti], finalize=fin)
        got = records.records(res, [ti], stage)[0]
        if "error" in exp:
            assert got == {"error": exp["error"]}
            continue
        assert not isinstance(got, dict), got
        assert gu.diff_records(got, exp[key]) == []
        assert float(res.coverage_average_total[0]) == exp["coverage_average_total"]


@pytest.mark.skipif(not __import__("make_ref").ref_root(), reason="needs the reference (its checkout, or the staged build oracle/_ref that make_ref.py compiles)")
def test_oracle_matches_reference_under_random_options(oracle_mod, monkeypatch, capsys):
    """oracle/ref_cfgfuzz.py: option sets drawn from the reference's own argparse definitions, the unmodified reference and the
    oracle on the same adversarial task with the same config object (the goldens pin the option sets of tests/cases.py only)."""
    import ref_cfgfuzz
    monkeypatch.setattr("sys.argv", ["ref_cfgfuzz.py", "40", "5000"])
    ref_cfgfuzz.main()
    out = capsys.readouterr().out
    assert "mismatching 0 " in out, out


def test_oracle_batch_equals_single(oracle_mod):
    """Tasks are independent: a batch of tasks gives the concatenation of the single-task results."""
    names = ["chr20_30x_ont", "bnd_first_error", "merge_inner", "fuzz_3_0"]
    tis = []
    for i, n in enumerate(names):
        ti = cases.ALL[n][0]()
        tis.append(ti)
    cfg = gu.make_config({}, tis[0])
    both = records.records(oracle_mod.run(cfg, tis, True), tis, "final")
    for ti, b in zip(tis, both):
        cfg1 = gu.make_config({}, ti)
        single = records.records(oracle_mod.run(cfg1, [ti], True), [ti], "final")[0]
        assert single == b


def test_np_pairwise_sum_matches_numpy(oracle_mod):
    rng = np.random.default_rng(0)
    for n in [0, 1, 7, 8, 9, 63, 127, 128, 129, 255, 1000, 4097]:
        x = rng.uniform(0, 0.1, n)
        assert oracle_mod.np_sum(x) == float(np.sum(x))


def test_stdev_matches_cpython_statistics(oracle_mod):
    import statistics
    rng = np.random.default_rng(1)
    for _ in range(300):
        n = int(rng.integers(2, 200))
        base = int(rng.integers(0, 250_000_000))
        x = (base + rng.integers(0, int(rng.choice([3, 100, 6000, 2_000_000])), n)).tolist()
        assert oracle_mod.stdev(x) == statistics.stdev(x)


def test_edit_distance_oracle_known_answers(oracle_mod):
    ed = oracle_mod.edit_distance
    assert ed(b"", b"") == 0
    assert ed(b"kitten", b"sitting") == 3
    assert ed(b"<DEL>", b"<DEL>") == 0
    assert ed(b"ACGT", b"") == 4
    assert ed(b"flaw", b"lawn") == 2
    assert ed(b"intention", b"execution") == 5


def test_diff_results_agrees_with_record_comparison(oracle_mod):
    """records.diff_results (the vectorised comparer of bench.py --verify and the full-size GPU test) must see what the
    per-record comparison sees: identical results -> no differences; any edited field, ALT byte or read id -> reported."""
    from sniffles_amd import synth
    from sniffles_amd.config import SnifflesConfig
    tis = [synth.gen_task(0, "chr22", 3_000_000, 30, 1), synth.gen_fuzz(5, task_id=1)]
    cfg = SnifflesConfig()
    a = oracle_mod.run(cfg, tis, True)
    for t, ti in enumerate(tis):
        b = oracle_mod.run(cfg, [ti], True)
        assert records.diff_results(a, t, b, 0) == []
        lo = int(a.task_call_off[t])
        k = lo + int(np.argmax(a.calls["alt_len"][lo:int(a.task_call_off[t + 1])] > 0))
        for field, delta in (("pos", 1), ("stdev_pos", 1e-9), ("gt_gq", 1), ("filter", 1)):
            c = oracle_mod.run(cfg, [ti], True)
            c.calls[field][k - lo] += delta
            assert records.diff_results(a, t, c, 0) != []
        c = oracle_mod.run(cfg, [ti], True)
        c.alt_pool[int(c.calls["alt_off"][k - lo])] ^= 1
        assert any("ALT" in d for d in records.diff_results(a, t, c, 0))
        c = oracle_mod.run(cfg, [ti], True)
        c.rnames[int(c.calls["rn_off"][0])] += 1000000
        assert any("reads" in d for d in records.diff_results(a, t, c, 0))
        c = oracle_mod.run(cfg, [ti], True)   # the order of the supporting reads of a call is free (list(set))
        o, n = int(c.calls["rn_off"][0]), int(c.calls["rn_len"][0])
        c.rnames[o:o + n] = c.rnames[o:o + n][::-1].copy()
        assert records.diff_results(a, t, c, 0) == []
