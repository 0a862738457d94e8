"""bench.py's `cpu_baseline` leg of kind "reference" (oracle/ref_pool.py): the UNMODIFIED reference's `Task.call_candidates` +
`finalize_candidates` + `CallTask.execute`'s filter / sort (`/root/reference/src/sniffles/parallel.py:104-201, 265-271`) in worker
processes over seeded contig tasks, compared record by record with the library's execute-mode block on the same tables - the
same comparison the bench line reports as `verified_vs_reference` at full size.  Runs wherever the reference is importable:
its checkout (build container) or the byte-compiled staged build `oracle/_ref` (oracle/make_ref.py), which is what the GPU
box has."""
import numpy as np
import pytest

from sniffles_amd import abi, lib, records, synth
from sniffles_amd.config import SnifflesConfig

pytestmark = pytest.mark.skipif(not __import__("make_ref").ref_root(), reason="needs the reference (its checkout, or the staged build oracle/_ref that make_ref.py compiles)")

SPECS0 = [(0, dict(task_id=0, contig="chr20", contig_len=1_500_000, coverage=30.0, seed=5)),
          (1, dict(task_id=1, contig="chr21", contig_len=900_000, coverage=30.0, seed=6)),
          (2, dict(task_id=2, contig="chrX", contig_len=600_000, coverage=30.0, seed=7))]


def check(L, extra, cfg_kw, device=None, gen=None):
    import ref_pool
    SPECS = [(k, dict(kw, **(gen or {}))) for k, kw in SPECS0]
    r = ref_pool.run_tasks(SPECS, extra, weights=[kw["contig_len"] for _, kw in SPECS], want_results=True, max_procs=2)
    assert r["procs"] == 2 and set(r["items"]) == {0, 1, 2}
    assert r["hot_all_core_s"] > 0 and r["hot_single_core_s"] >= r["hot_all_core_s"]
    tis = [synth.gen_task(**kw) for _, kw in SPECS]
    kw = dict(device=device or 0)
    with lib.Batch(SnifflesConfig(**cfg_kw), tis, **kw) as b:
        b.set_output(abi.OUT_EXECUTE)
        b.call_candidates(); b.finalize()
        exe = b.fetch(1)
    got = records.records(exe, tis, "final")
    n = 0
    for t, (key, _) in enumerate(SPECS):
        exp = r["items"][key]["records"]
        assert len(exp) == r["items"][key]["kept"] and len(got[t]) == len(exp)
        for a, e in zip(got[t], exp):
            assert a == e, (e["id"], [k for k in e if a.get(k) != e.get(k)])
        assert float(exe.coverage_average_total[t]) == r["items"][key]["coverage_average_total"]
        n += len(exp)
    assert n > (3 if gen else 20)
    assert all(np.all(np.diff([c["pos"] for c in g]) >= 0) for g in got)      # CallTask.execute's sort


@pytest.mark.parametrize("extra,cfg_kw", [((), {}), (("--mosaic",), {"mosaic": True})])
def test_reference_pool_equals_execute_block_emu(extra, cfg_kw):
    import emu.emu as E
    check(E.lib(), extra, cfg_kw, gen=dict(mosaic_frac=0.5) if cfg_kw else None)


@pytest.mark.gpu
def test_reference_pool_equals_execute_block_gpu():
    check(None, (), {})
