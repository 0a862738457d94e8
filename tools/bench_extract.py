"""Extraction kernels on the MI355X: records/s, signatures/s and HBM roofline fraction of the two passes, with the
CPU oracle (pure-Python restatement, like the reference's own per-alignment Python loop) timed on a sample beside it.

    python tools/bench_extract.py [--reads 3000] [--tile 8] [--steps 10] > profiles/rNN_extract_bench.json

Synthetic ONT-like records (20-kb reads, about one CIGAR operation per 7 aligned bases, 20 % with SA tags); the record
table is tiled `--tile` times so that one run covers tens of thousands of records.  Algorithmic bytes per record =
record minus its sequence and quality bytes, plus 1.5 B per inserted base that reaches the sequence pool.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=3000)
    ap.add_argument("--tile", type=int, default=8)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--cpu-reads", type=int, default=150)
    ap.add_argument("--sa-frac", type=float, default=0.2)
    ap.add_argument("--no-tags", action="store_true", help="records without auxiliary fields (what the tag walk and the SA parser cost)")
    ap.add_argument("--read-len", type=int, default=20000)
    a = ap.parse_args()
    from sniffles_amd import bam, extract, synth_bam
    t0 = time.time()
    names, lens, recs = synth_bam.gen_records(2026, a.reads, style="ont", read_len_mean=a.read_len, sa_frac=a.sa_frac, with_tags=not a.no_tags,
                                              ref_lens=(60_000_000, 300000, 300000, 100000))
    gen_s = time.time() - t0
    R = bam.records_from_list(names, lens, recs * a.tile)
    n_cig = int(sum(int.from_bytes(r[16:18], "little") for r in recs)) * a.tile
    x = extract.Extractor()
    t0 = time.time()
    x.upload(R, "chrA", 0, 60_000_000)
    up_s = time.time() - t0
    x.run()
    ti, info = x.result()
    cnt, emt, wall = [], [], []
    for _ in range(a.steps):
        t0 = time.time()
        x.run()
        wall.append(time.time() - t0)
        _, inf = x.result()
        cnt.append(inf.ms_count)
        emt.append(inf.ms_emit)
    ms_c, ms_e = float(np.median(cnt)), float(np.median(emt))
    peak = 8000.0
    out = dict(
        sa_frac=a.sa_frac,
        workload=f"{R.n} synthetic ONT-like alignment records ({a.reads} distinct x {a.tile}), {n_cig} CIGAR operations, "
                 f"{R.blob.nbytes / 1e6:.0f} MB of inflated BAM records in HBM",
        records=R.n, reads_accepted=info.read_count, signatures=ti.n_leads, seq_pool_bytes=int(ti.seq_pool.shape[0]),
        algo_bytes=info.algo_bytes, ms_count_pass=ms_c, ms_emit_pass=ms_e,
        records_per_s=R.n / ((ms_c + ms_e) * 1e-3), signatures_per_s=ti.n_leads / ((ms_c + ms_e) * 1e-3),
        roofline=dict(bound="hbm", unit="GB/s", peak=peak,
                      count_pass=dict(achieved=info.algo_bytes / (ms_c * 1e6), frac=info.algo_bytes / (ms_c * 1e6) / peak),
                      emit_pass=dict(achieved=info.algo_bytes / (ms_e * 1e6), frac=info.algo_bytes / (ms_e * 1e6) / peak)),
        wall_ms_run_incl_scans_and_result_copy=float(np.median(wall)) * 1e3, upload_s=up_s, generate_s=gen_s)
    # CPU baseline: the oracle restatement on a sample of the same records (1 core)
    import extract_oracle as eo
    sub = R.select(range(min(a.cpu_reads, a.reads)))
    t0 = time.time()
    want = eo.extract_region(sub.blob, sub.rec_off, sub.ref_names, "chrA", 0, 60_000_000)
    cpu_s = time.time() - t0
    out["cpu_baseline"] = dict(kind="port", cores=1, sample=f"first {sub.n} records, oracle/extract_oracle.py (pure Python)",
                               records_per_s=sub.n / cpu_s, signatures_per_s=len(want["rows"]) / cpu_s)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
