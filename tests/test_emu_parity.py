"""Kernel-logic parity WITHOUT a GPU: the exact kernel bodies of sniffles_amd/csrc (compiled for the
host by tests/emu, serial loops instead of launches) against the reference goldens and the oracle.
This is a development aid for the GPU-less build container; the parity tests proper are
tests/test_gpu_parity.py (-m gpu), which run the real gfx950 library."""
import numpy as np
import pytest

import cases
import golden_util as gu
from sniffles_amd import lib, records, synth
from sniffles_amd.config import SnifflesConfig


@pytest.fixture(scope="module")
def emu_lib():
    import emu.emu as E
    return E.lib()


def run(L, cfg, tis, fin):
    with lib.Batch(cfg, tis, _lib=L) as b:
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
