"""The C-ABI library builds for gfx950, loads, and exports every symbol include/sniffles_amd.h declares.
No compute calls (no GPU in the CPU tier)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def so():
    from sniffles_amd import build
    return build.build()


def header_functions():
    src = open(os.path.join(ROOT, "include", "sniffles_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(snf_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(so):
    lib = C.CDLL(so)
    fns = header_functions()
    assert len(fns) >= 14
    for f in fns:
        assert hasattr(lib, f), f"{f} declared in include/sniffles_amd.h but not exported"


def test_abi_version_and_struct_sizes(so):
    from sniffles_amd import abi, lib
    L = lib.load()
    assert L.snf_abi_version() == abi.ABI_VERSION
    # the ctypes mirrors must have the C layout (checked against sizes compiled into the oracle build)
    assert C.sizeof(abi.snf_call_t) % 8 == 0
    assert abi.CALL_DTYPE.itemsize == C.sizeof(abi.snf_call_t)


def test_no_device_fails_loudly(so):
    """Without a HIP device the product path must raise - there is no CPU fallback."""
    from sniffles_amd import lib
    from sniffles_amd.config import SnifflesConfig
    import cases
    if lib.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(lib.SnifflesAmdError, match="no HIP device"):
        lib.Batch(SnifflesConfig(), [cases.case_resplit_wrap()])


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "sniffles_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".h", ".hip")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "snf_oracle" not in txt and "ref_harness" not in txt, f


def test_more_than_65535_tasks_in_one_batch_are_refused():
    """Documented limit of a batch (include/sniffles_amd.h): the task index travels in 16 bits of the packed
    keys.  The upload fails with the reason - nothing is truncated; several batches serve larger jobs."""
    import emu.emu as E
    E.lib()                                              # the host tier becomes the library of this test
    from sniffles_amd import lib, synth
    from sniffles_amd.config import SnifflesConfig
    ti = synth.gen_fuzz(3, task_id=0)
    with pytest.raises(lib.SnifflesAmdError, match="too many tasks in one batch"):
        lib.Batch(SnifflesConfig(), [ti] * 65536)


def test_batch_open_is_create_add_upload_and_the_first_call_in_one(oracle_mod):
    """`snf_batch_open` (ABI 5; what `Task.prepare` runs on its helper thread): the same results as the call sequence it stands for,
    for every run mode; a failure (unknown mode, a task the reference could not have produced, no device) leaves no handle behind and
    reports the FIRST error; without a device the real library refuses like `snf_batch_create`."""
    import numpy as np
    import emu.emu as E
    from sniffles_amd import abi, lib, records, synth
    from sniffles_amd.config import SnifflesConfig
    cfg = SnifflesConfig()
    tis = [synth.gen_task(0, "chr21", 600_000, 30, 5), synth.gen_fuzz(9, task_id=1)]
    if lib.device_count() == 0:
        with pytest.raises(lib.SnifflesAmdError, match="no HIP device"):      # (the product library, no GPU: fails loudly here too)
            lib.Batch.open_in_background(cfg, tis, run=lib.Batch.RUN_CANDIDATES).result()
    E.lib()                                              # the host tier becomes the library of this test
    exp = records.records(oracle_mod.run(cfg, tis, True), tis, "final")
    with lib.Batch(cfg, tis) as b:
        b.set_output(abi.OUT_EXECUTE); b.run_pass()
        want_exe = b.fetch(1)
    for run in (lib.Batch.RUN_NONE, lib.Batch.RUN_CANDIDATES, lib.Batch.RUN_PASS | abi.OUT_CANDIDATES, lib.Batch.RUN_PASS | abi.OUT_EXECUTE):
        b = lib.Batch.open_in_background(cfg, tis, run=run).result()
        try:
            if run == lib.Batch.RUN_NONE:
                b.call_candidates()
            if run in (lib.Batch.RUN_NONE, lib.Batch.RUN_CANDIDATES):
                b.finalize()
            got = b.fetch(1)
            if run == lib.Batch.RUN_PASS | abi.OUT_EXECUTE:
                assert got.calls.tobytes() == want_exe.calls.tobytes() and got.alt_pool.tobytes() == want_exe.alt_pool.tobytes()
            else:
                assert records.records(got, tis, "final") == exp
        finally:
            b.close()
    with pytest.raises(lib.SnifflesAmdError, match="unknown run mode"):
        lib.Batch.open_in_background(cfg, tis, run=7).result()
    bad = synth.gen_fuzz(9, task_id=2)
    bad.leads["svtype"] = bad.leads["svtype"].copy()
    bad.leads["svtype"][0] = 99                          # not a code the reference's SV types have: the upload refuses the task
    with pytest.raises(lib.SnifflesAmdError):
        lib.Batch.open_in_background(cfg, [tis[0], bad], run=lib.Batch.RUN_CANDIDATES).result()
    pend = lib.Batch.open_in_background(cfg, tis, run=lib.Batch.RUN_CANDIDATES)
    pend.discard()                                       # prepared and not wanted: closed, no handle leaks
    assert pend._batch is None


def test_worker_processes_plumbing_on_the_host_tier(monkeypatch):
    """tools/bench_workers.py (bench.py's `wall_clock.worker_processes` leg): spawned workers, a common barrier, every input form and
    call shape - on the host tier of the test suite, so that the leg cannot rot unseen (never a measurement)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_workers as W
    monkeypatch.setenv("SNF_BENCH_EMU", "1")
    specs = W.genome_specs(scale=0.003)[:4]
    out = {}
    for form, shape in (("columns", "api"), ("columns", "execute"), ("leads", "api")):
        r = W.run(specs, {}, 2, form, shape, hw_queues=2)
        assert r["procs"] == 2 and r["hot_all_ms"] > 0 and r["n_out"] > 0
        out[(form, shape)] = r["n_out"]
    # the same calls from objects and from columns, through the two calls + CallTask.execute's own filter and sort, and in one step
    assert out[("columns", "api")] == out[("leads", "api")] == out[("columns", "execute")] > 0
    # ... and with ONE process on the device (sniffles_amd.server; here: the host tier started inside the server process)
    from sniffles_amd import server
    srv = server.start(device=0, init="emu.emu:lib", extra_path=[os.path.join(ROOT, "tests")], arena_mb=0)
    monkeypatch.setenv("SNF_GPU_SERVER", srv.address)
    try:
        r = W.run(specs, {}, 3, "leads", "api")
        assert r["gpu_server"] is True and r["n_out"] == out[("columns", "api")]
    finally:
        srv.stop()
