import sys, time, threading, argparse
sys.path.insert(0, "/root/repo")
import torch; torch.cuda.init()
import bench
from sniffles_amd import abi, lib, synth
from sniffles_amd.config import SnifflesConfig
wl = bench.WORKLOADS[0]
args = argparse.Namespace(coverage=None, scale=1.0)
cfg = SnifflesConfig(**wl["cfg"])
tasks = [synth.gen_task(**kw) for _, kw in bench.task_specs(args, wl, 0, 0, 1)]
def run(every, big):
    keep = []
    if big:   # two large idle batches alive, as in the bench's compact block
        wl1 = bench.WORKLOADS[1]
        t1 = [synth.gen_task(**kw) for _, kw in bench.task_specs(args, wl1, 0, 0, 1)][:8]
        keep = [lib.Batch(cfg, t1, device=0) for _ in range(2)]
        for k in keep: k.set_output(abi.OUT_EXECUTE); k.run_pass(); k.fetch_raw(1)
    hs = [lib.Batch(cfg, tasks, device=0) for _ in range(2)]
    for h in hs:
        h.set_output(abi.OUT_EXECUTE)
        if every is not None: h.timing_every(every)
    def passes(n):
        def body(h):
            torch.cuda.set_device(0)
            for _ in range(n): h.run_pass(); h.fetch_raw(1)
        ths = [threading.Thread(target=body, args=(h,)) for h in hs]
        [t.start() for t in ths]; [t.join() for t in ths]
    passes(8); torch.cuda.synchronize()
    t0 = time.perf_counter(); passes(24); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    t1_ = time.perf_counter(); hs[0].run_pass(); hs[0].fetch_raw(1); lat = time.perf_counter() - t1_
    for h in hs + keep: h.close()
    return round(dt / 48 * 1e3, 4), round(lat * 1e3, 4)
for every, big in ((None, False), (0, False), (1, False), (None, True), (0, True)):
    print("timing_every", every, "large batches alive", big, "->", run(every, big), flush=True)
