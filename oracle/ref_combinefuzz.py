"""TEST INFRASTRUCTURE ONLY (build container: needs /root/reference) - the multi-sample merge driver
(sniffles_amd.parallel.CombineTask.execute: columnar candidate store, grouping and SVGroup.call kernels, run here through
the host builds of the library) against the UNMODIFIED reference's CombineTask.execute under random option sets.

tests/golden/combine_task_* pin five option sets; this sweep draws the --combine-* options (and the filters the merge
reads) from the reference's own argparse definitions, builds a small synthetic population, lets the reference produce the
per-sample SNF blocks and the combined calls (oracle/ref_harness.py::run_reference_combine_task) and replays the same blocks
through the product's driver.   python oracle/ref_combinefuzz.py [n] [seed0] [--simt]
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

POOL = dict(
    combine_match=[50, 500, 1000], combine_match_max=[300, 5000], combine_separate_intra=[True], combine_pctseq=[0.0, 0.5, 0.9],
    combine_high_confidence=[0.3, 0.6], combine_low_confidence=[0.05, 0.4], combine_low_confidence_abs=[1, 4],
    combine_null_min_coverage=[1, 12], combine_output_filtered=[True], combine_support_threshold=[1, 5],
    combine_pair_relabel=[True], combine_pair_relabel_threshold=[5, 40], dev_combine_medians=[True],
    minsvlen=["50", "100"], cluster_binsize=[50, 200], cluster_binsize_combine_mult=[1, 10], no_qc=[True], mosaic=[True],
)


def main():
    import golden_util as gu
    import ref_harness as rh
    import test_combine_task as T
    from test_combine import group_record
    from sniffles_amd import parallel, synth
    args_in = [a for a in sys.argv[1:] if not a.startswith("--")]
    n_iter = int(args_in[0]) if args_in else 20
    seed0 = int(args_in[1]) if len(args_in) > 1 else 0
    if "--simt" in sys.argv:
        from emu import simt as E
    else:
        from emu import emu as E
    L = E.lib()
    C = rh.load_reference().config.SnifflesConfig
    parser = argparse.ArgumentParser(add_help=False)
    for add in (C.add_main_args, C.add_filter_args, C.add_cluster_args, C.add_genotype_args, C.add_multi_args,
                C.add_postprocess_args, C.add_mosaic_args, C.add_developer_args):
        add(parser)
    acts = {a.dest: a for a in parser._actions}
    missing = sorted(k for k in POOL if k not in acts)
    if missing:
        print("not options of the reference (skipped):", missing)
    bad = 0; n_calls = 0; n_cands = 0; t0 = time.time(); ref_failed = {}
    for it in range(seed0, seed0 + n_iter):
        rng = np.random.default_rng([it, 131071])
        args = []
        for k, vals in POOL.items():
            a = acts.get(k)
            if a is None or rng.random() >= 0.3:
                continue
            v = vals[int(rng.integers(len(vals)))]
            if isinstance(a, argparse._StoreTrueAction):
                if v:
                    args.append(a.option_strings[0])
            else:
                args += [a.option_strings[0], str(v)]
        ns = int(rng.integers(2, 7))
        dens = float(rng.choice([10, 40])) * 27000 / 3.1e9
        tis = [synth.gen_task(it % 7, "chr19", int(rng.choice([300_000, 700_000])), float(rng.choice([10, 20])), seed=9000 + 13 * it + s,
                              site_seed=777 + it, site_density=dens) for s in range(ns)]
        try:
            exp = rh.run_reference_combine_task(tis, tuple(args))
        except SystemExit:
            continue
        except Exception as e:                    # the reference's own failure on this option set: nothing to compare with
            ref_failed[type(e).__name__ + ": " + str(e)[:60]] = ref_failed.get(type(e).__name__ + ": " + str(e)[:60], []) + [" ".join(args)]
            continue
        doc = dict(expected=exp, reference_args=[a for a in args])
        try:
            cfg = T._twin_cfg(doc, ())
        except Exception as e:                    # "~50"-style values etc. the twin helper does not parse
            print("skipped", args, type(e).__name__, e)
            continue
        readers = {s: T.BlocksReader(exp["contig"], exp["samples"][s]) for s in range(exp["n_samples"])}
        task = parallel.CombineTask(id=7, sv_id=0, contig=exp["contig"], start=0, end=exp["contig_len"], config=cfg)
        got = [group_record(c) for c in task.execute(readers)]
        want = exp["calls"]
        n_calls += len(want)
        n_cands += sum(len(v) for smp in exp["samples"] for blk in smp for v in blk["cands"].values())
        diffs = [] if len(got) == len(want) else [f"{len(got)} calls, reference {len(want)}"]
        for g, w in zip(got, want):
            d = gu.diff_records([g], [w])
            if d:
                diffs.append(d[0]); break
        if diffs:
            bad += 1
            print("MISMATCH it", it, ns, "samples", " ".join(args), "|", str(diffs)[:600], flush=True)
    for k, v in ref_failed.items():
        print("the reference raised", k, "on", len(v), "option sets, e.g.", v[0])
    print("ref_combinefuzz: iterations", n_iter, "candidates", n_cands, "combined calls", n_calls, "mismatching", bad, "seconds", round(time.time() - t0, 1))


if __name__ == "__main__":
    main()
