"""The UNMODIFIED reference (CPython, stub pysam) timed on contigs of the bench workload, in the build container
(needs /root/reference; the GPU box has none).  Times Task.call_candidates + finalize_candidates only, like
bench.py's cpu_baseline; lead-table construction is outside the timed part.

    python tools/time_reference.py chr21 chr22 > profiles/rNN_reference_cpu.json
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]


def main():
    import ref_harness as rh
    from sniffles_amd import synth
    contigs = sys.argv[1:] or ["chr21", "chr22"]
    rows, tot_sig, tot_s = [], 0, 0.0
    for c in contigs:
        ti = synth.gen_task(synth.CONTIGS.index(c), c, synth.GRCH38[c], 30.0, seed=1)
        cfg = rh.make_config((), ti.qc_nm_threshold)
        task = rh.build_task(ti, cfg)
        t0 = time.perf_counter()
        cands = task.call_candidates(True, cfg)
        t1 = time.perf_counter()
        task.finalize_candidates(cands, False, cfg)
        t2 = time.perf_counter()
        rows.append(dict(contig=c, signatures=int(ti.n_leads), candidates=len(cands), call_candidates_s=round(t1 - t0, 3),
                         finalize_candidates_s=round(t2 - t1, 3)))
        tot_sig += int(ti.n_leads)
        tot_s += t2 - t0
    print(json.dumps(dict(what="unmodified reference (CPython 3.10, 1 core of the build container), Task.call_candidates + finalize_candidates, "
                               "30x synthetic contigs of the bench workload (seed 1)", contigs=rows, signatures=tot_sig,
                          seconds=round(tot_s, 2), signatures_per_s=round(tot_sig / tot_s))))


if __name__ == "__main__":
    main()
