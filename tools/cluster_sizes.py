"""Histogram of the refined clusters' sizes of the bench workload (how many of a wave's 64 lanes work in d1w_refine / d2w_call)."""
import sys, json
import numpy as np
sys.path.insert(0, ".")
import torch
torch.cuda.init()
import bench
from sniffles_amd import lib, synth
from sniffles_amd.config import SnifflesConfig
wl = bench.WORKLOADS[1]
import argparse
a = argparse.Namespace(config=1, coverage=None, scale=1.0, genomes=1)
specs = bench.task_specs(a, wl, 0, 0, 1)
tasks = [synth.gen_task(**kw) for _, kw in specs]
cfg = SnifflesConfig(**wl["cfg"])
with lib.Batch(cfg, tasks) as b:
    b.call_candidates()
    cl = b.fetch_clusters(2)
n = np.diff(cl["lead_off"])
h = {k: int(((n > lo) & (n <= hi)).sum()) for k, (lo, hi) in {"1": (0, 1), "2-4": (1, 4), "5-8": (4, 8), "9-16": (8, 16), "17-32": (16, 32), "33-64": (32, 64), ">64": (64, 1 << 30)}.items()}
print(json.dumps(dict(clusters=int(len(n)), leads=int(n.sum()), mean=float(n.mean()), hist=h)))
