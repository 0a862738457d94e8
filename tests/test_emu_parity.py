"""The fallback forms of the pass WITHOUT a GPU: SNF_NO_WAVE=1 (thread-per-item kernels instead of the wave / workgroup kernels -
the bodies the big-cluster kernels and the rare ALT fallbacks also run) and SNF_NO_FUSE=1 (device-wide scans instead of the
fused flag / scan / emit pairs - what batches beyond 2^25 positions take), through the host tier (tests/emu/simt), against the
reference goldens and the oracle.  tests/test_simt_tier.py runs the default (wave, fused) forms the same way; the parity tests
proper are tests/test_gpu_parity.py (-m gpu) on the real gfx950 library."""
import numpy as np
import pytest

import cases
import golden_util as gu
from sniffles_amd import lib, records, synth
from sniffles_amd.config import SnifflesConfig


@pytest.fixture
def emu_lib():
    import emu.emu as E
    return E.lib()


@pytest.fixture(autouse=True)
def fallback_forms(monkeypatch):
    monkeypatch.setenv("SNF_NO_WAVE", "1")
    monkeypatch.setenv("SNF_NO_FUSE", "1")


def run(L, cfg, tis, fin):
    with lib.Batch(cfg, tis) as b:
        b.call_candidates()
        if fin:
            b.finalize()
        return b.fetch(1 if fin else 0)


@pytest.mark.parametrize("name", sorted(cases.ALL))
def test_emulated_kernels_match_reference_golden(name, emu_lib):
    build, kw, _ = cases.ALL[name]
    doc = gu.load(name)
    ti = build()
    cfg = gu.make_config(kw, ti)
    exp = doc["expected"]
    for stage, key, fin in (("cand", "candidates", False), ("final", "final", True)):
        res = run(emu_lib, cfg, [ti], fin)
        got = records.records(res, [ti], stage)[0]
        if "error" in exp:
            assert got == {"error": exp["error"]}
            continue
        assert gu.diff_records(got, exp[key]) == []
        assert float(res.coverage_average_total[0]) == exp["coverage_average_total"]


@pytest.mark.parametrize("gap", [None, "0", "150", "-1"])
def test_merge_scan_run_cuts_are_exact(gap, emu_lib, oracle_mod, monkeypatch):
    """The merge scan is cut into independent runs at gaps > run_gap and validated; invalid cuts fall
    back to the serial scan.  Any cut width must reproduce the oracle exactly."""
    if gap is None:
        monkeypatch.delenv("SNF_RUN_GAP", raising=False)
    else:
        monkeypatch.setenv("SNF_RUN_GAP", gap)
    for seed in range(6):
        tis = [synth.gen_fuzz(100 * seed + k, task_id=k) for k in range(3)]
        for kw in ({}, dict(repeat=True, mosaic=True)):
            cfg = SnifflesConfig(**kw)
            exp = records.records(oracle_mod.run(cfg, tis, True), tis, "final")
            got = records.records(run(emu_lib, cfg, tis, True), tis, "final")
            assert got == exp


def test_wide_sort_keys_are_exact(emu_lib, oracle_mod, monkeypatch):
    """SNF_SORT64 forces the 64-bit sort keys used when (task, svtype, bin) or the read-end key space exceeds 32 bits."""
    monkeypatch.setenv("SNF_SORT64", "1")
    tis = [synth.gen_fuzz(300 + k, task_id=k) for k in range(4)]
    cfg = SnifflesConfig()
    assert records.records(run(emu_lib, cfg, tis, True), tis, "final") == records.records(oracle_mod.run(cfg, tis, True), tis, "final")


def test_vote_column_equals_most_common_rule(emu_lib):
    """The column vote of the LDS-vote consensus kernels (packed per-base counters + escape list, csrc/snf_stage_final.h
    `vote_column`) against a direct restatement of consensus.py:365-380 / util.most_common on random columns, odd
    characters included."""
    import ctypes as C
    import numpy as np
    f = emu_lib.snf_emu_vote_column
    f.argtypes = [C.c_uint32, C.POINTER(C.c_uint32), C.c_int, C.c_int, C.c_int, C.c_int]
    f.restype = C.c_int
    rng = np.random.default_rng(11)
    code = {65: 0, 67: 1, 84: 2, 71: 3}
    alpha_plain, alpha_odd = [65, 67, 71, 84], [78, 97, 99, 82, 45 + 1, 255, 0]
    for trial in range(4000):
        nkept = int(rng.integers(0, 40))
        nvotes = int(rng.integers(0, nkept + 1))
        odd_rate = float(rng.choice([0.0, 0.0, 0.1, 0.5]))
        dom = int(rng.choice(alpha_plain + alpha_odd))
        votes = [int(dom if rng.random() < 0.6 else (rng.choice(alpha_odd) if rng.random() < odd_rate else rng.choice(alpha_plain)))
                 for _ in range(nvotes)]
        bq = int(rng.choice(alpha_plain + alpha_odd)) if rng.random() < 0.2 else int(rng.choice(alpha_plain))
        q = int(rng.integers(0, 300))
        cnt4, esc = 0, []
        for c in votes:
            if c in code:
                cnt4 += 1 << (8 * code[c])
            else:
                esc.append((q << 8) | c)
        for _ in range(int(rng.integers(0, 4))):        # escapes of other columns must be ignored
            esc.append((int(rng.integers(300, 400)) << 8) | int(rng.choice(alpha_odd)))
        order = rng.permutation(len(esc))
        esc = [esc[i] for i in order]
        # the rule
        exp = bq
        if not (len(votes) < 2 or len(votes) / (1 + nkept) < 0.25):
            cnt = {}
            for c in [bq] + votes:
                cnt[c] = cnt.get(c, 0) + 1
            ranked = sorted(((n, c) for c, n in cnt.items()), reverse=True)
            if len(ranked) > 1 and ranked[0][0] - ranked[1][0] >= 3:
                exp = ranked[0][1]
        arr = (C.c_uint32 * max(1, len(esc)))(*esc)
        assert f(cnt4, arr, len(esc), q, bq, nkept) == exp, (votes, bq, nkept)

