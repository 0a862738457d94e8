"""Where the per-task API (Task.call_candidates + finalize_candidates, one contig at a time) spends a genome's 0.9 s."""
import os
import sys
import time

sys.path.insert(0, "/root/repo")
from sniffles_amd import parallel, pipeline, synth
from sniffles_amd.config import SnifflesConfig

tasks = synth.gen_genome(30.0, seed=1)
cfg = SnifflesConfig()
for rep in range(2):
    os.environ["SNF_PROF"] = "1" if rep else "0"
    if not rep:
        del os.environ["SNF_PROF"]
    tot = dict(open=0.0, cand=0.0, fin=0.0, close=0.0)
    t_all = time.perf_counter()
    for ti in tasks:
        task = parallel.CallTask(id=ti.task_id, sv_id=0, contig=ti.contig, start=0, end=ti.contig_len, config=cfg, tandem_repeats=None, device=0)
        task.lead_provider = pipeline._Extracted(ti)
        t0 = time.perf_counter()
        cands = task.call_candidates(False, cfg)
        t1 = time.perf_counter()
        task.finalize_candidates(cands, True, cfg)
        t2 = time.perf_counter()
        task.close()
        t3 = time.perf_counter()
        tot["cand"] += t1 - t0; tot["fin"] += t2 - t1; tot["close"] += t3 - t2
    print("rep", rep, "total %.1f ms" % ((time.perf_counter() - t_all) * 1e3), {k: round(v * 1e3, 1) for k, v in tot.items()}, flush=True)
