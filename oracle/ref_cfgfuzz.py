"""TEST INFRASTRUCTURE ONLY (build container: needs /root/reference) - the oracle against the UNMODIFIED reference under
random combinations of the reference's own command-line options.

tests/golden pins the oracle on the option sets of tests/cases.py; this sweep checks the rest of the option space the hot path
reads (filters, cluster / merge widths, mosaic and developer switches): every iteration draws options from the reference's
argparse definitions, builds the reference's SnifflesConfig from the command line, runs the reference's call_candidates /
finalize_candidates (oracle/ref_harness.py) and the oracle on the same adversarial task with the SAME config object, and
compares every record field.   python oracle/ref_cfgfuzz.py [n] [seed0]
"""
import argparse
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import numpy as np

# option -> values worth drawing (everything else of the reference's parser keeps its default)
POOL = dict(
    minsupport=["auto", "1", "2", "5"], minsupport_auto_mult=[0.025, 0.2], minsvlen=["50", "~30", "100", "~200"],
    no_qc=[True], qc_stdev=[False], qc_stdev_abs_max=[50, 5], qc_strand=[True], qc_coverage=[5, 20], long_ins_length=[200, 60],
    long_del_length=[500, 100], long_inv_length=[500], long_del_coverage=[0.9, 0.3], long_dup_length=[500, 100], long_dup_coverage=[1.1, 2.0],
    qc_bnd_filter_strand=[False], phase_conflict_threshold=[0.0, 0.5], detect_large_ins=[False], cluster_binsize=[50, 200, 25],
    cluster_r=[1.0, 4.0, 0.2], cluster_repeat_h=[0.5, 5.0], cluster_repeat_h_max=[100.0], cluster_merge_pos=[50, 300, 0],
    cluster_merge_len=[0.5, 0.05], cluster_merge_bnd=[100, 5000], genotype_error=[0.01, 0.2], no_consensus=[True], symbolic=[True],
    mosaic=[True], mosaic_af_max=[0.4, 0.1], mosaic_af_min=[0.01, 0.1], mosaic_qc_invdup_min_length=[50], mosaic_qc_nm=[False],
    mosaic_qc_nm_mult=[1.0], mosaic_qc_coverage_max_change_frac=[0.3], mosaic_qc_strand=[False], mosaic_include_germline=[True],
    max_svlen_mosaic=[500], mosaic_min_reads=[1, 6], mosaic_use_strand_thresholds=[3], consensus_max_reads_bin=[3, 25],
    dev_no_resplit=[True], dev_no_resplit_repeat=[True], repeat=[True], qc_nm=[True], qc_nm_mult=[1.0, 3.0],
    qc_coverage_max_change_frac=[0.3, 0.05], coverage_updown_bins=[2, 9], cluster_resplit_binsize=[5, 60], dev_no_qc=[True],
    dev_min_leads_cluster=[3], dev_min_dup_vaf=[0.4], dev_longer_del=[300], dev_longer_dup=[300], dev_minreads_extra=[1],
    dev_maxsvlen_extra=[100], dev_inline_sa_support_max=[0.3], dev_min_close_edge_dist=[50], dev_min_read_close_edge_prop=[0.2],
    phase=[True],
)


def main():
    import golden_util as gu
    import oracle
    import ref_harness as rh
    from sniffles_amd import records, synth
    n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    oracle.build()
    C = rh.load_reference().config.SnifflesConfig          # the reference's own option definitions (config.py:174-446)
    parser = argparse.ArgumentParser(add_help=False)
    for add in (C.add_main_args, C.add_filter_args, C.add_cluster_args, C.add_genotype_args, C.add_multi_args,
                C.add_postprocess_args, C.add_mosaic_args, C.add_developer_args):
        add(parser)
    acts = {a.dest: a for a in parser._actions}
    missing = sorted(k for k in POOL if k not in acts)
    if missing:
        print("not options of the reference (skipped):", missing)
    bad = 0; n_calls = 0; t0 = time.time()
    for it in range(seed0, seed0 + n_iter):
        rng = np.random.default_rng([it, 524287])
        args = []
        for k, vals in POOL.items():
            a = acts.get(k)
            if a is None or rng.random() >= 0.2:
                continue
            v = vals[int(rng.integers(len(vals)))]
            if isinstance(a, argparse._StoreTrueAction):
                if v:
                    args.append(a.option_strings[0])
            elif isinstance(a, argparse._StoreFalseAction):
                if not v:
                    args.append(a.option_strings[0])
            else:
                args += [a.option_strings[0], str(v)]
        ti = synth.gen_fuzz(200000 + it, task_id=it % 5) if it % 4 else \
            synth.gen_task(it % 5, "chrG", 150_000, float(rng.choice([15, 40, 90])), seed=it, site_density=2e-4, mosaic_frac=0.3)
        try:
            cfg = rh.make_config(tuple(args), ti.qc_nm_threshold)
        except SystemExit:                      # argparse rejected the combination
            continue
        ref = rh.run_reference(ti, cfg=cfg)
        diffs = []
        for stage, key, fin in (("cand", "candidates", False), ("final", "final", True)):
            res = oracle.run(cfg, [ti], finalize=fin)
            got = records.records(res, [ti], stage)[0]
            if "error" in ref:
                if got != {"error": ref["error"]}:
                    diffs.append(f"{stage}: reference raised {ref['error']}, oracle {str(got)[:80]}")
                continue
            if isinstance(got, dict):
                diffs.append(f"{stage}: oracle {got}")
                continue
            diffs += [f"{stage}: {d}" for d in gu.diff_records(got, ref[key])]
            if float(res.coverage_average_total[0]) != ref["coverage_average_total"]:
                diffs.append(f"{stage}: coverage_average_total")
            n_calls += len(got)
        if diffs:
            bad += 1
            print("MISMATCH it", it, " ".join(args), "|", str(diffs[:2])[:700], flush=True)
    print("ref_cfgfuzz: iterations", n_iter, "records compared", n_calls, "mismatching", bad, "seconds", round(time.time() - t0, 1))


if __name__ == "__main__":
    main()
