"""Development check on the GPU: a population merge (candidates from this package's calling path, SNF blocks in memory as in
bench.py --config 4) through the columnar candidate store and through the object-by-object replay - every attribute of every
combined call must be equal.   python tools/dev/combine_twin_gpu.py [scale] [samples]"""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import bench_population as bp
from sniffles_amd import parallel, synth
from sniffles_amd.config import SnifflesConfig

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
S = int(sys.argv[2]) if len(sys.argv) > 2 else 10
contigs = [(ci, c, max(200000, int(synth.GRCH38[c] * scale))) for ci, c in enumerate(synth.CONTIGS)]
call_cfg = SnifflesConfig()


def readers():
    out = {}
    for s in range(S):
        tasks = [synth.gen_task(ci, c, L, 15.0, seed=100 + s, site_seed=501) for ci, c, L in contigs]
        out[s], _ = bp.build_sample(call_cfg, tasks, 0, s)
    return out


def run(objects, extra):
    os.environ["SNF_COMBINE_OBJECTS"] = "1" if objects else "0"
    cfg = SnifflesConfig(**extra)
    cfg.mode = "combine"
    cfg.snf_input_info = [dict(internal_id=s, sample_id=f"S{s}") for s in range(S)]
    cfg.sample_ids_vcf = [(s, f"S{s}") for s in range(S)]
    tasks = [parallel.CombineTask(id=ci, sv_id=0, contig=c, start=0, end=L - 1, config=cfg, device=0) for ci, c, L in contigs]
    calls = parallel.CombineTask.execute_many(tasks, readers())        # (the merge writes on the candidates: fresh readers per run)
    return [[dict(vars(c), forward_difference_sampler=vars(c.forward_difference_sampler)) for c in part] for part in calls]


bad = 0
for extra in ({}, dict(combine_pair_relabel=True, combine_output_filtered=True), dict(dev_combine_medians=True, combine_null_min_coverage=12)):
    col, obj = run(False, extra), run(True, extra)
    n = sum(len(p) for p in col)
    same = [len(a) == len(b) and all(x == y for x, y in zip(a, b)) for a, b in zip(col, obj)]
    print(extra, "combined calls", n, "tasks equal", sum(same), "of", len(same), flush=True)
    bad += len(same) - sum(same)
print("combine twin on the GPU: mismatching tasks", bad)
