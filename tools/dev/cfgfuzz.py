"""Development sweep WITHOUT a GPU: random combinations of the hot-path options (filters, cluster / merge widths, mosaic and
developer switches) on adversarial tasks, the library built for the host (tests/emu/simt: the HIP sources unchanged, wave and
lock-step kernels included; `--emu`: the serial emulation) against the oracle.   python tools/dev/cfgfuzz.py [n] [seed0] [--emu]"""
import sys
import time

sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/oracle")
import numpy as np
import oracle
from sniffles_amd import lib, records, synth
from sniffles_amd.config import SnifflesConfig

POOL = dict(
    phase=[False], minsupport=["auto", "1", "2", "5"], minsupport_auto_mult=[0.025, 0.2], minsvlen=["50", "~30", "100", "~200"],
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
)


def sweep(L, n_iter, seed0=0, verbose=True):
    """n_iter random option sets on three tasks each, library handle L against the oracle -> (mismatching batches, calls)"""
    oracle.build()
    bad = 0; calls = 0
    for it in range(seed0, seed0 + n_iter):
        rng = np.random.default_rng([it, 8191])
        kw = {k: v[int(rng.integers(len(v)))] for k, v in POOL.items() if rng.random() < 0.2}
        cfg = SnifflesConfig(**kw)
        tis = [synth.gen_fuzz(100000 + 7 * it + k, task_id=k) for k in range(2)]
        tis.append(synth.gen_task(2, "chrG", 120_000, float(rng.choice([12, 30, 90])), seed=it, err=float(rng.choice([0.005, 0.04])),
                                  site_density=2e-4, mosaic_frac=float(rng.choice([0.0, 0.3]))))
        exp = oracle.run(cfg, tis, True)
        with lib.Batch(cfg, tis) as b:
            b.call_candidates(); b.finalize(); got = b.fetch(1)
        calls += len(exp.calls)
        diffs = [d for t in range(len(tis)) for d in records.diff_results(got, t, exp, t)]
        if diffs or not np.array_equal(got.coverage_average_total, exp.coverage_average_total, equal_nan=True):
            bad += 1
            if verbose:
                print("MISMATCH it", it, kw, str(diffs[:2])[:600], flush=True)
    return bad, calls


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    n_iter = int(args[0]) if args else 100
    seed0 = int(args[1]) if len(args) > 1 else 0
    if "--emu" in sys.argv:
        from emu import emu as T
    else:
        from emu import simt as T
    t0 = time.time()
    bad, calls = sweep(T.lib(), n_iter, seed0)
    print("cfgfuzz: iterations", n_iter, "calls", calls, "mismatching batches", bad, "seconds", round(time.time() - t0, 1))
