"""Development sweep on the GPU: deep clusters (80x - 700x, ONT- and HiFi-like error, dense sites, mosaic sites) under nine flag
sets against the oracle - the big-cluster kernels (x_big<0/1/2>: LDS rows, rows exceeded), the batched finalize and the private
fused-sequence slices all get work here that the 30x workloads hardly give them.   python tools/dev/deepfuzz.py [n_seeds]"""
import sys
import time

sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/oracle")
import numpy as np
import oracle
from sniffles_amd import lib, records, synth
from sniffles_amd.config import SnifflesConfig

KW = [{}, dict(mosaic=True), dict(minsupport="auto", qc_nm=True), dict(no_qc=True),
      dict(qc_strand=True, minsvlen="50", cluster_merge_pos=50), dict(repeat=True, mosaic=True, mosaic_include_germline=True),
      dict(no_consensus=True), dict(symbolic=True), dict(phase=False)]
oracle.build()
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
bad = 0; n = 0; big = 0; t0 = time.time()
for ci, kw in enumerate(KW):
    cfg = SnifflesConfig(**kw)
    for seed in range(n_seeds):
        rng = np.random.default_rng(7000 + 31 * ci + seed)
        cov = float(rng.choice([80, 150, 300, 700]))
        tis = [synth.gen_task(k, f"chr{21 + k}", int(rng.choice([150_000, 300_000])), cov, 900 + 10 * seed + k, err=float(rng.choice([0.005, 0.04])),
                              read_len_mean=float(rng.choice([8000, 15000])), site_density=2e-4, mosaic_frac=float(rng.choice([0.0, 0.3])))
               for k in range(2)]
        exp = oracle.run(cfg, tis, True)
        with lib.Batch(cfg, tis) as b:
            b.call_candidates(); b.finalize(); got = b.fetch(1)
        n += len(exp.calls); big += int(sum(int(c["n_leads"]) > 64 for c in exp.calls))
        diffs = [d for t in range(len(tis)) for d in records.diff_results(got, t, exp, t)]
        if diffs or not np.array_equal(got.coverage_average_total, exp.coverage_average_total, equal_nan=True):
            bad += 1; print("MISMATCH cfg", ci, kw, "seed", seed, "cov", cov, diffs[:3], flush=True)
print("deepfuzz: calls", n, "of which > 64 leads", big, "mismatching batches", bad, "seconds", round(time.time() - t0, 1))
