import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/oracle")
import numpy as np
import oracle
from sniffles_amd import lib, records, synth
from sniffles_amd.config import SnifflesConfig
KW = [{}, dict(mosaic=True), dict(minsupport="auto", qc_nm=True), dict(no_qc=True),
      dict(qc_strand=True, minsvlen="50", cluster_merge_pos=50), dict(repeat=True, mosaic=True, mosaic_include_germline=True),
      dict(no_consensus=True), dict(symbolic=True), dict(phase=False)]
oracle.build()
bad = 0; n = 0; t0 = time.time()
for ci, kw in enumerate(KW):
    cfg = SnifflesConfig(**kw)
    for seed in range(1000, 1000 + int(sys.argv[1]), 5):
        tis = [synth.gen_fuzz(seed + k, task_id=k) for k in range(5)]
        exp = oracle.run(cfg, tis, True)
        with lib.Batch(cfg, tis) as b:
            b.call_candidates(); b.finalize(); got = b.fetch(1)
        a, e = records.records(got, tis, "final"), records.records(exp, tis, "final")
        n += sum(len(x) if isinstance(x, list) else 0 for x in e)
        if a != e or not np.array_equal(got.coverage_average_total, exp.coverage_average_total, equal_nan=True):
            bad += 1; print("MISMATCH cfg", ci, "seed", seed)
print("batches checked, calls", n, "bad", bad, "seconds", round(time.time() - t0, 1))
