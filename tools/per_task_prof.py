#!/usr/bin/env python3
"""Where a per-contig task's wall clock goes (tools/r05_*.sh on the GPU box): 24 x CallTask.execute_calls with the phases of
sniffles_amd.parallel timed one by one, and the library's own split of the upload (SNF_PROF) on stderr."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sniffles_amd import abi, lib, parallel, pipeline, sv, synth  # noqa: E402
from sniffles_amd.config import SnifflesConfig  # noqa: E402

cfg = SnifflesConfig()
tasks = [synth.gen_task(task_id=i, contig=c, contig_len=synth.GRCH38[c], coverage=30.0, seed=1) for i, c in enumerate(synth.CONTIGS)]
with lib.Batch(cfg, tasks[-2:]) as b:      # process warm-up: HIP context, pinned arena, first slab
    b.run_pass(); b.fetch(1, copy=False)
if len(sys.argv) > 1 and sys.argv[1] == "prof":
    os.environ["SNF_PROF"] = "1"
acc = {}
def lap(name, t0):
    t = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t - t0); return t
for rnd in range(2):
    acc.clear()
    T0 = time.perf_counter()
    for ti in tasks:
        t = time.perf_counter()
        task = parallel.CallTask(id=ti.task_id, sv_id=0, contig=ti.contig, start=0, end=ti.contig_len, config=cfg, tandem_repeats=None, device=0)
        task.lead_provider = pipeline._Extracted(ti)
        t = lap("task object", t)
        task._open(cfg); t = lap("open (to_task_input + create + add + upload)", t)
        task._batch.set_output(abi.OUT_EXECUTE); task._batch.run_pass(); t = lap("pass enqueue", t)
        res = task._batch.fetch(1, copy=False); t = lap("fetch (host wait)", t)
        calls = sv.materialize_candidates(res, task._ti, 0, len(res.calls)); sv.apply_final(calls, res, task._ti)
        for c in calls:
            c.finalize()
        t = lap("materialise", t)
        task.close(); t = lap("close", t)
    total = time.perf_counter() - T0
    print(f"round {rnd}: 24 tasks {total * 1e3:.1f} ms | " + " | ".join(f"{k} {v * 1e3:.1f}" for k, v in acc.items()))
