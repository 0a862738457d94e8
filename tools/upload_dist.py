"""Distribution of the upload time of one genome (snf_batch_create + add_task + upload), the library's own split per upload
(SNF_PROF lines on stderr).  usage: python tools/upload_dist.py [repeats]   (GPU box)"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import torch
torch.cuda.init()
from sniffles_amd import lib, synth
from sniffles_amd.config import SnifflesConfig
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
args = argparse.Namespace(coverage=None, scale=1.0)
wl = bench.WORKLOADS[1]
cfg = SnifflesConfig(**wl["cfg"])
tasks = [synth.gen_task(**kw) for _, kw in bench.task_specs(args, wl, 0, 0, 1)]
nbytes = bench._input_bytes(tasks)
print("input bytes", nbytes, flush=True)
os.environ["SNF_PROF"] = "1"
keep = None
for k in range(n):
    t0 = time.perf_counter()
    b = lib.Batch(cfg, tasks, device=0)
    t1 = time.perf_counter()
    print(f"upload {k}: {(t1 - t0) * 1e3:.1f} ms ({nbytes / (t1 - t0) / 1e9:.1f} GB/s)", flush=True)
    if k == n // 2: keep = b        # (one handle stays alive: its slabs are not returned to the cache)
    else: b.close()
    sys.stderr.flush()
