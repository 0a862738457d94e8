"""snf_batch_set_output (include/sniffles_amd.h): SNF_OUT_EXECUTE returns what `CallTask.execute` keeps of the finalized
candidates (`/root/reference/src/sniffles/parallel.py:265-271`: `[s for s in svcalls if s.qc]` unless `config.no_qc`, then
`sorted(svcalls, key=pos)` - a stable sort) - filtered, sorted and compacted on the device.  Checked against the same two
statements applied on the host to the SNF_OUT_CANDIDATES result (which the other parity tests pin on the oracle and the
reference's goldens); `snf_batch_export_device` hands out the same block in HBM."""
import ctypes as C

import numpy as np
import pytest

from sniffles_amd import abi, lib, synth
from sniffles_amd.config import SnifflesConfig


def tasks():
    # a BND-first task raises in the reference (UnboundLocalError): no records in either mode
    import cases
    return [synth.gen_task(0, "chr20", 1_200_000, 30, 3, mosaic_frac=0.3), cases.ALL["bnd_first_error"][0](), synth.gen_fuzz(21, task_id=2),
            synth.gen_fuzz(22, task_id=3), synth.gen_task(4, "chr21", 400_000, 25, 9)]


def run(L, cfg, tis, mode):
    with lib.Batch(cfg, tis) as b:
        b.set_output(mode)
        b.call_candidates()
        b.finalize()
        return b.fetch(1)


def expected_execute(cand, cfg):
    """CallTask.execute's two statements over the candidate-mode result: per task (order of kept call indices)."""
    keep = []
    for t in range(len(cand.task_status)):
        lo, hi = int(cand.task_call_off[t]), int(cand.task_call_off[t + 1])
        idx = np.arange(lo, hi)
        if not cfg.no_qc:
            idx = idx[cand.calls["qc"][lo:hi] != 0]
        if cfg.sort:
            idx = idx[np.argsort(cand.calls["pos"][idx], kind="stable")]
        keep.append(idx)
    return keep


def check(L):
    tis = tasks()
    for kw in ({}, {"mosaic": True}, {"no_qc": True}):
        cfg = SnifflesConfig(**kw)
        cand = run(L, cfg, tis, abi.OUT_CANDIDATES)
        got = run(L, cfg, tis, abi.OUT_EXECUTE)
        assert np.array_equal(got.task_status, cand.task_status) and int(cand.task_status[1]) == 1
        assert np.array_equal(got.coverage_average_total, cand.coverage_average_total, equal_nan=True)
        keep = expected_execute(cand, cfg)
        assert got.task_call_off.tolist() == np.concatenate([[0], np.cumsum([len(k) for k in keep])]).tolist()
        idx = np.concatenate(keep)
        assert len(got.calls) == len(idx) and (len(idx) > 0 or cfg.mosaic)
        if not cfg.no_qc:
            assert len(idx) < len(cand.calls) and (got.calls["qc"] != 0).all()
        for f in got.calls.dtype.names:
            if f == "rn_off":
                continue
            a, e = got.calls[f], cand.calls[f][idx]
            assert np.array_equal(a, e, equal_nan=a.dtype.kind == "f"), f
        for k, i in enumerate(idx.tolist()):
            assert got.alt(k) == cand.alt(i)
            assert got.rn(k).tolist() == cand.rn(i).tolist()
        # nothing but the kept calls' records and read names travels; the ALT section is the candidates' (written by the ALT
        # kernels themselves, the dropped calls' few bytes ride along)
        assert len(got.alt_pool) == len(cand.alt_pool) and got.alt_pool.tobytes() == cand.alt_pool.tobytes()
        assert len(got.rnames) == int(cand.calls["rn_len"][idx].sum())


def test_execute_mode_is_the_filter_and_sort_of_the_candidates_emu():
    import emu.emu as E
    check(E.lib())


def test_execute_mode_is_the_filter_and_sort_of_the_candidates_simt():
    from emu import simt as S
    check(S.lib())


@pytest.mark.gpu
def test_execute_mode_is_the_filter_and_sort_of_the_candidates_gpu():
    check(None)


def export_equals_fetch(L, device_alloc):
    tis = tasks()
    cfg = SnifflesConfig()
    with lib.Batch(cfg, tis) as b:
        b.set_output(abi.OUT_EXECUTE | abi.OUT_DEVICE)
        b.call_candidates()
        b.finalize()
        res = b.fetch(1)
        with pytest.raises(lib.SnifflesAmdError, match="too small"):
            b.export_device(0, 16)
        ptr, read_back = device_alloc(1 << 24)
        lay = b.export_device(ptr, 1 << 24)
        blob = read_back(lay["bytes"])
    assert lay["n_calls"] == len(res.calls) and lay["rnames_len"] == len(res.rnames) and lay["alt_pool_len"] == len(res.alt_pool)
    rec = np.frombuffer(blob[:lay["n_calls"] * abi.CALL_DTYPE.itemsize], abi.CALL_DTYPE)
    assert rec.tobytes() == res.calls.tobytes()
    assert blob[lay["off_rnames"]:lay["off_rnames"] + 4 * lay["rnames_len"]] == res.rnames.tobytes()
    assert blob[lay["off_alt"]:lay["off_alt"] + lay["alt_pool_len"]] == res.alt_pool.tobytes()
    with lib.Batch(cfg, tis) as b:     # without SNF_OUT_DEVICE the block may never have been in HBM
        b.call_candidates()
        b.finalize()
        with pytest.raises(lib.SnifflesAmdError, match="SNF_OUT_DEVICE"):
            b.export_device(0, 1 << 24)


def test_export_device_hands_out_the_fetched_block_emu():
    import emu.emu as E
    buf = {}

    def alloc(n):
        buf["a"] = np.zeros(n, np.uint8)
        return buf["a"].ctypes.data, lambda k: buf["a"][:k].tobytes()
    export_equals_fetch(E.lib(), alloc)


@pytest.mark.gpu
def test_export_device_hands_out_the_fetched_block_gpu():
    import torch
    buf = {}

    def alloc(n):
        buf["a"] = torch.zeros(n, dtype=torch.uint8, device="cuda")
        return buf["a"].data_ptr(), lambda k: buf["a"][:k].cpu().numpy().tobytes()
    export_equals_fetch(None, alloc)


# ---------------------------------------------------------------------------------------------- the rare ALT kernels
def slow_alt_task():
    """INS calls the LDS classes of the consensus stage do not serve: alleles of more than 8192 columns (ROWS instance) and
    an allele whose reads carry more non-ACGT bytes than the escape list of the vote counters holds (handed to work list 7
    at run time) - next to an ordinary call.  finalize enqueues without knowing; the fetch finds the work lists and runs the
    slow kernels after the fact."""
    import cases
    rng = np.random.default_rng(5)
    leads = []

    def site(pos, allele, n, tag, err=0.03, odd=0):
        for i in range(n):
            s = cases._mutate(rng, allele, err)
            if odd:
                b = list(s)
                for k in rng.choice(len(b), odd, replace=False):
                    b[k] = "N"
                s = "".join(b)
            leads.append(dict(svtype="INS", ref_start=pos + i % 3, svlen=len(s), seq=s, read=f"{tag}{i}", strand="+-"[i % 2]))
    site(20_000, cases._rng_seq(rng, 9000), 6, "L")
    site(60_000, cases._rng_seq(rng, 300), 8, "N", odd=40)
    site(90_000, cases._rng_seq(rng, 250), 7, "S")
    return cases.mk_task(leads, cases._reads(30, 0, 119_000), 120_000)


def check_slow_alt(L, oracle_mod):
    from sniffles_amd import records
    ti = slow_alt_task()
    cfg = SnifflesConfig()
    exp = oracle_mod.run(cfg, [ti], True)
    for mode in (abi.OUT_CANDIDATES, abi.OUT_EXECUTE):
        with lib.Batch(cfg, [ti]) as b:
            b.set_output(mode)
            b.call_candidates()
            b.finalize()
            got = b.fetch(1)
            again = b.fetch(1)            # (the settled state is stable)
        assert got.alt_pool.tobytes() == again.alt_pool.tobytes()
        if mode == abi.OUT_CANDIDATES:
            assert records.diff_results(got, 0, exp, 0) == []
        alts = sorted(got.alt(i) for i in range(len(got.calls)) if got.calls["alt_len"][i] > 0)
        want = sorted(exp.alt(i) for i in range(len(exp.calls)) if exp.calls["alt_len"][i] > 0 and (mode == abi.OUT_CANDIDATES or exp.calls["qc"][i]))
        assert alts == want and len(alts) == 3 and max(map(len, alts)) > 8192


def test_slow_alt_kernels_are_settled_by_the_fetch_simt(oracle_mod):
    from emu import simt as S
    before = S.counters()["launches"]
    check_slow_alt(S.lib(), oracle_mod)
    assert S.counters()["launches"] > before


def test_slow_alt_kernels_emu(oracle_mod):
    import emu.emu as E
    check_slow_alt(E.lib(), oracle_mod)


@pytest.mark.gpu
def test_slow_alt_kernels_are_settled_by_the_fetch_gpu(oracle_mod):
    check_slow_alt(None, oracle_mod)


# ---------------------------------------------------------------------------------------------- deferred read names
def check_deferred_names(L):
    """SNF_OUT_EXECUTE set before call_candidates: the supporting read names are written late, for the kept calls only.  A
    stage-0 fetch in between still sees every candidate's names, and a later switch of the mode still gets them all."""
    tis = tasks()
    cfg = SnifflesConfig()
    ref0 = None
    with lib.Batch(cfg, tis) as b:
        b.call_candidates()
        ref0 = b.fetch(0)
        b.finalize()
        ref1 = b.fetch(1)
    with lib.Batch(cfg, tis) as b:
        b.set_output(abi.OUT_EXECUTE)
        b.call_candidates()
        got0 = b.fetch(0)                       # all names, written by the late pass
        b.finalize()
        exe = b.fetch(1)
        b.finalize()                            # (again: idempotent)
        exe2 = b.fetch(1)
        b.set_output(abi.OUT_CANDIDATES)        # mode switched after the candidate stage
        b.finalize()
        cand = b.fetch(1)
    assert got0.calls.tobytes() == ref0.calls.tobytes() and got0.rnames.tobytes() == ref0.rnames.tobytes()
    assert exe.calls.tobytes() == exe2.calls.tobytes() and exe.rnames.tobytes() == exe2.rnames.tobytes()
    assert cand.calls.tobytes() == ref1.calls.tobytes() and cand.rnames.tobytes() == ref1.rnames.tobytes()
    keep = np.concatenate(expected_execute(ref1, cfg))
    assert [exe.rn(k).tolist() for k in range(len(exe.calls))] == [ref1.rn(i).tolist() for i in keep.tolist()]
    with lib.Batch(cfg, tis) as b:      # finalize straight after the candidate stage: names of the kept calls only
        b.set_output(abi.OUT_EXECUTE)
        b.call_candidates()
        b.finalize()
        exe3 = b.fetch(1)
    assert exe3.calls.tobytes() == exe.calls.tobytes() and exe3.rnames.tobytes() == exe.rnames.tobytes()


def test_deferred_read_names_simt():
    from emu import simt as S
    check_deferred_names(S.lib())


@pytest.mark.gpu
def test_deferred_read_names_gpu():
    check_deferred_names(None)


def check_timing_is_sampled():
    """`snf_batch_timing_every`: the HIP-event brackets ride on every n-th pass of a handle (the first one included), the LARGE
    consensus kernel - the one bench.py states the roofline on - on every pass; a pass that was not sampled reports nothing else."""
    from sniffles_amd import lib, synth
    from sniffles_amd.config import SnifflesConfig
    ti = synth.gen_task(0, "chr21", 1_500_000, 30, 5)
    with lib.Batch(SnifflesConfig(), [ti]) as b:
        seen = []
        for _ in range(10):
            b.run_pass(); b.fetch(1)
            seen.append({n for n, _, _ in b.timings()})
        assert len(seen[0]) > 5 and len(seen[8]) > 5 and "e45w_consensus_small" in seen[8], (seen[0], seen[8])   # passes 0 and 8: every bracket
        for k in (1, 2, 7, 9):
            assert seen[k] <= {"e45w_consensus_large"}, (k, seen[k])
        b.timing_every(1)
        b.run_pass(); b.fetch(1)
        assert {n for n, _, _ in b.timings()} == seen[8]
        b.timing_every(0)
        b.run_pass(); b.fetch(1)
        assert {n for n, _, _ in b.timings()} <= {"e45w_consensus_large"}


def test_kernel_timing_is_sampled_emu():
    import emu.emu as E
    E.lib()
    check_timing_is_sampled()


@pytest.mark.gpu
def test_kernel_timing_is_sampled_gpu():
    check_timing_is_sampled()


def check_consensus_class_order(oracle_mod, monkeypatch):
    """The two consensus classes side by side, SMALL behind LARGE, LARGE behind SMALL (what the library picks when another pass
    is in flight): the same records."""
    from sniffles_amd import lib, records, synth
    from sniffles_amd.config import SnifflesConfig
    tis = [synth.gen_task(0, "chr21", 1_200_000, 30, 11), synth.gen_fuzz(5, task_id=1)]
    cfg = SnifflesConfig()
    exp = records.records(oracle_mod.run(cfg, tis, True), tis, "final")
    for order in ("0", "1", "2"):
        monkeypatch.setenv("SNF_CONS_ORDER", order)
        with lib.Batch(cfg, tis) as b:
            b.run_pass()
            assert records.records(b.fetch(1), tis, "final") == exp, order


def test_consensus_class_order_emu(oracle_mod, monkeypatch):
    import emu.emu as E
    E.lib()
    check_consensus_class_order(oracle_mod, monkeypatch)


@pytest.mark.gpu
def test_consensus_class_order_gpu(oracle_mod, monkeypatch):
    check_consensus_class_order(oracle_mod, monkeypatch)


# ---------------------------------------------------------------------------------------------- result staged through HBM
def check_staged_result(oracle_mod, monkeypatch):
    """With another pass in flight the kernels store the result into HBM and two copies behind them take it to the pinned buffers
    (at the fetch, or by two small copy kernels inside the pass): the same block as the direct stores, pass after pass, in
    both output modes, also when the slow ALT kernels run at the fetch, and with two handles driven from two threads."""
    import threading
    from sniffles_amd import records
    tis = tasks()
    cfg = SnifflesConfig()

    def blocks(b, n):
        out = []
        for _ in range(n):
            b.run_pass()
            r = b.fetch(1)
            out.append((r.calls.tobytes(), r.rnames.tobytes(), r.alt_pool.tobytes(), r.task_call_off.tobytes()))
        return out
    for mode, copy in ((abi.OUT_CANDIDATES, "fetch"), (abi.OUT_EXECUTE, "fetch"), (abi.OUT_CANDIDATES, "kernel"), (abi.OUT_EXECUTE, "kernel")):
        monkeypatch.setenv("SNF_STAGE_OUT", "0")
        with lib.Batch(cfg, tis) as b:
            b.set_output(mode)
            want = blocks(b, 1)[0]
        monkeypatch.setenv("SNF_STAGE_OUT", "1")
        monkeypatch.setenv("SNF_STAGE_COPY", copy)         # the two copies at the fetch, or two copy kernels inside the pass
        with lib.Batch(cfg, tis) as b:
            b.set_output(mode)
            assert blocks(b, 3) == [want] * 3
            b.set_output(abi.OUT_EXECUTE if mode == abi.OUT_CANDIDATES else abi.OUT_CANDIDATES)   # another block size on the same handle
            other = blocks(b, 2)
            assert other[0] == other[1] and other[0] != want
    # slow ALT kernels (settled by the fetch): the ALT section is taken again behind them
    ti = slow_alt_task()
    exp = oracle_mod.run(cfg, [ti], True)
    with lib.Batch(cfg, [ti]) as b:
        for _ in range(3):
            b.run_pass()
            assert records.diff_results(b.fetch(1), 0, exp, 0) == []
    # default rule (no knob): staged when another pass is in flight - two handles, two threads, the same blocks
    monkeypatch.delenv("SNF_STAGE_OUT"); monkeypatch.delenv("SNF_STAGE_COPY")
    with lib.Batch(cfg, tis) as b1, lib.Batch(cfg, tis) as b2:
        b1.set_output(abi.OUT_EXECUTE); b2.set_output(abi.OUT_EXECUTE)
        got = {}

        def body(key, b):
            got[key] = blocks(b, 4)
        ths = [threading.Thread(target=body, args=(k, b)) for k, b in (("a", b1), ("b", b2))]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
    assert got["a"] == [want] * 4 and got["b"] == [want] * 4


def test_pacing_state_is_per_device(monkeypatch):
    """Whether a pass stages its result through HBM (and whether its start is paced) is decided by what is in flight on ITS device:
    two handles that overlap on device 0 switch to the staged result (the fetch then copies: a `d2h_block` entry in the timings), a
    handle that works alone on device 1 of the same process keeps its direct stores all the while (host tier with two device indices)."""
    import threading
    import time
    monkeypatch.setenv("SNF_SIMT_DEVICES", "2")
    monkeypatch.setenv("SNF_TIME_ALL", "1")           # (the copies of a staged result are bracketed by timing events only then)
    import emu.emu as E
    E.lib()
    assert lib.device_count() == 2
    tis = tasks()
    cfg = SnifflesConfig()
    with lib.Batch(cfg, tis, device=0) as a1, lib.Batch(cfg, tis, device=0) as a2, lib.Batch(cfg, tis, device=1) as lone:
        for b in (a1, a2, lone):
            b.set_output(abi.OUT_EXECUTE)
            b.timing_every(1)
        stop = threading.Event()
        staged = {"a": 0}

        def busy(b):
            while not stop.is_set():
                b.run_pass(); b.fetch(1)
                staged["a"] += any(t[0] == "d2h_block" for t in b.timings())
        ths = [threading.Thread(target=busy, args=(b,)) for b in (a1, a2)]
        for t in ths:
            t.start()
        lone_staged, want = 0, None
        t_end = time.time() + 60
        n = 0
        while (n < 6 or staged["a"] < 3) and time.time() < t_end:
            lone.run_pass()
            r = lone.fetch(1)
            blk = (r.calls.tobytes(), r.alt_pool.tobytes())
            want = want or blk
            assert blk == want
            lone_staged += any(t[0] == "d2h_block" for t in lone.timings())
            n += 1
        stop.set()
        for t in ths:
            t.join()
    assert staged["a"] >= 3, "the two handles of device 0 never overlapped"
    assert lone_staged == 0 and n >= 6


def test_staged_result_emu(oracle_mod, monkeypatch):
    import emu.emu as E
    E.lib()
    check_staged_result(oracle_mod, monkeypatch)


@pytest.mark.gpu
def test_staged_result_gpu(oracle_mod, monkeypatch):
    check_staged_result(oracle_mod, monkeypatch)
