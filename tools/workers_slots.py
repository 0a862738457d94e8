#!/usr/bin/env python3
"""tools/workers_slots.py P SLOTS|server [form ...]: P worker processes sharing the GPU through the two-call seam (tools/bench_workers.py)
with at most SLOTS passes driving the device at a time (SNF_GPU_SLOTS, sniffles_amd/csrc/snf_lib.hip; 0 = unbounded), or - `server` -
through ONE GPU server process (sniffles_amd/server.py).  One JSON line per form."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    use_server = sys.argv[2] == "server"
    procs, slots = int(sys.argv[1]), (0 if use_server else int(sys.argv[2]))
    forms = sys.argv[3:] or ["columns", "leads"]
    if slots > 0:
        os.environ["SNF_GPU_SLOTS"] = str(slots)      # (spawned workers inherit it)
    else:
        os.environ.pop("SNF_GPU_SLOTS", None)
    from tools import bench_workers as W
    specs = W.genome_specs()
    for form in forms:
        r = W.run(specs, {}, procs, form, "api", hw_queues=2 if procs >= 8 else 0, gpu_server=use_server)
        print(json.dumps(dict(procs=procs, gpu_slots=slots, gpu_server=use_server, form=form,
                              **{k: r[k] for k in ("hot_all_ms", "wall_ms", "ingest_all_ms", "n_out", "slowest_worker_served")})), flush=True)
