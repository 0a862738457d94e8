"""One GPU server process for many workers (sniffles_amd/server.py): the per-task seam - Task.call_candidates / finalize_candidates,
CallTask.call_svs, CallTask.execute_calls - through a spawned server that owns the device (here: the host tier of the kernels), task
inputs and result blocks in shared memory, whatever has arrived run as ONE device batch.  The calls are the ones the in-process path
and the unmodified reference produce (goldens)."""
import multiprocessing as mp
import os
import sys

import pytest

import cases
import golden_util as gu
from sniffles_amd import parallel, pipeline, server, sv

HERE = os.path.dirname(os.path.abspath(__file__))
NAMES = ["bnd_stale_end", "merge_inner", "long_ins", "phase_rescue", "consensus_quirks", "chr21_30x_mosaic", "fuzz_4_2", "single_leads_noqc"]


def as_final(c):
    import test_dropin_api as T
    return T.as_record(c, "final")


def make_task(name):
    build, kw, _ = cases.ALL[name]
    ti = build()
    cfg = gu.make_config(kw, ti)
    task = parallel.CallTask(id=ti.task_id, sv_id=ti.sv_id_start, contig=ti.contig, start=0, end=ti.contig_len, config=cfg)
    task.lead_provider = pipeline._Extracted(ti)
    task.tandem_repeats = None
    return task, cfg, ti, gu.load(name)["expected"]


@pytest.fixture(scope="module")
def srv():
    h = server.start(device=0, init="emu.emu:lib", extra_path=[HERE], arena_mb=0)
    os.environ["SNF_GPU_SERVER"] = h.address
    yield h
    os.environ.pop("SNF_GPU_SERVER", None)
    h.stop()


@pytest.mark.parametrize("name", NAMES)
def test_seam_through_the_server(name, srv):
    task, cfg, ti, exp = make_task(name)
    cands = task.call_candidates(True, cfg)
    assert type(cands) is list and len(cands) == len(exp["final"]) and all(sv.is_stand_in(c) and c.qc for c in cands)
    assert task.coverage_average_total == exp["coverage_average_total"] and task.sv_id == ti.sv_id_start + len(cands)
    final = task.finalize_candidates(cands, True, cfg)
    assert [bool(c.qc) for c in final] == [r["qc"] for r in exp["final"]] and all(sv.is_stand_in(c) for c in final)
    assert [as_final(c) for c in final] == exp["final"] and all(type(c) is sv.SVCall and c.postprocess is None for c in final)
    task.close()
    # CallTask.execute's tail both ways: the two calls + filter + sort, and the one-step form
    t2, cfg2, _, _ = make_task(name)
    t3, cfg3, _, _ = make_task(name)
    want = final if cfg.no_qc else [c for c in final if c.qc]
    want = sorted(want, key=lambda c: c.pos) if cfg.sort else want
    assert [as_final(c) for c in t2.call_svs(cfg2)] == [as_final(c) for c in want]
    assert [as_final(c) for c in t3.execute_calls(cfg3)] == [as_final(c) for c in want]
    assert t3.sv_id == task.sv_id
    t2.close(); t3.close()


def test_error_of_the_reference_comes_through(srv):
    task, cfg, ti, exp = make_task("bnd_first_error")
    assert "error" in exp
    with pytest.raises(UnboundLocalError):
        task.call_candidates(True, cfg)
    # ... and the server goes on serving
    t2, cfg2, _, exp2 = make_task("long_ins")
    assert [as_final(c) for c in t2.finalize_candidates(t2.call_candidates(True, cfg2), True, cfg2)] == exp2["final"]
    t2.close()


def test_a_client_that_leaves_gives_its_segments_back(srv):
    """A worker that disappears with a reply it never released (killed, crashed): the server takes the result segment back when the
    connection closes, drops the mapping of the worker's input segment, and goes on serving."""
    import glob
    task, cfg, ti, exp = make_task("long_ins")
    c = server.Client(srv.address)
    rep = c.run_task(cfg, task._task_input(cfg))
    seg_name = rep.msg["seg"]
    assert len(rep.result.calls) == len(exp["final"])
    c.conn.close()                                  # no release, no goodbye
    for s_ in c._in:
        s_.unlink()
    import time
    for _ in range(50):                             # the same segment serves the next batch of that size once it is free again
        t2, cfg2, _, exp2 = make_task("long_ins")
        cands = t2.call_candidates(True, cfg2)
        name2 = sv._raw_dict(cands[0])["_lz"].on_detach.__self__.msg["seg"] if cands else None
        assert [as_final(c_) for c_ in t2.finalize_candidates(cands, True, cfg2)] == exp2["final"]
        t2.close()
        if name2 == seg_name:
            break
        time.sleep(0.05)
    assert name2 == seg_name
    assert len(glob.glob("/dev/shm/snfsrv_*")) <= 8


def _worker(address, names, barrier, out_q):
    try:
        for p in (os.path.dirname(HERE), HERE):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ["SNF_GPU_SERVER"] = address
        import test_server as TS
        tasks = [TS.make_task(n) for n in names]
        barrier.wait(timeout=600)
        out = {}
        for (task, cfg, ti, exp), n in zip(tasks, names):
            kept = task.call_svs(cfg)
            out[n] = ([TS.as_final(c) for c in kept], task.sv_id - ti.sv_id_start)
            task.close()
        out_q.put(out)
    except BaseException as e:  # noqa: BLE001
        import traceback
        out_q.put(dict(error=f"{e!r}\n{traceback.format_exc()}"))


def test_workers_share_one_server(srv):
    """Three worker processes (spawned: none of them touches the device) submit at a barrier: the server batches what has arrived,
    every worker gets its own task's calls."""
    ctx = mp.get_context("spawn")
    shards = [NAMES[0::3], NAMES[1::3], NAMES[2::3]]
    barrier, q = ctx.Barrier(len(shards)), ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(srv.address, sh, barrier, q), daemon=True) for sh in shards]
    for p in ps:
        p.start()
    got = {}
    for _ in ps:
        m = q.get(timeout=900)
        assert "error" not in m, m.get("error")
        got.update(m)
    for p in ps:
        p.join(timeout=60)
    for n in NAMES:
        task, cfg, ti, exp = make_task(n)
        want = [r for r in exp["final"] if cfg.no_qc or r["qc"]]
        want = sorted(want, key=lambda r: r["pos"]) if cfg.sort else want
        assert got[n][0] == want and got[n][1] == len(exp["final"]), n


@pytest.mark.gpu
def test_seam_through_the_server_gpu():
    """The same on the MI355X: the server process is the only one that opens the device; this process and three spawned workers
    go through it."""
    h = server.start(device=0)
    os.environ["SNF_GPU_SERVER"] = h.address
    try:
        for name in NAMES:
            test_seam_through_the_server(name, h)
        test_error_of_the_reference_comes_through(h)
        test_workers_share_one_server(h)
    finally:
        os.environ.pop("SNF_GPU_SERVER", None)
        h.stop()
