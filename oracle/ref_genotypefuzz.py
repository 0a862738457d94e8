"""TEST INFRASTRUCTURE ONLY (build container: needs /root/reference) - force calling (seam B6, GenotypeTask.execute:
target SVs matched against the sample's candidates and annotated with its coverage in the library, run here through its host
builds) against the UNMODIFIED reference's GenotypeTask.execute on random adversarial tasks, random target sets derived from
the reference's own candidates (tests/genotype_util.py: moved, resized, retyped, foreign targets) and random options.
tests/golden/genotype_targets.json.gz pins six cases.   python oracle/ref_genotypefuzz.py [n] [seed0] [--simt]
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import numpy as np

# (probability, command line, the same as SnifflesConfig keywords)
OPTIONS = [
    (0.25, ("--mosaic",), dict(mosaic=True)), (0.2, ("--minsupport", "auto"), dict(minsupport="auto")),
    (0.2, ("--minsvlen", "30"), dict(minsvlen="30")), (0.2, ("--no-qc",), dict(no_qc=True)), (0.2, ("--qc-nm",), dict(qc_nm=True)),
    (0.2, ("--cluster-binsize", "50"), dict(cluster_binsize=50)), (0.2, ("--cluster-merge-pos", "50"), dict(cluster_merge_pos=50)),
    (0.2, ("--long-ins-length", "300"), dict(long_ins_length=300)), (0.2, ("--repeat",), dict(repeat=True)),
    (0.15, ("--dev-no-resplit",), dict(dev_no_resplit=True)), (0.2, ("--genotype-error", "0.2"), dict(genotype_error=0.2)),
    (0.2, ("--coverage-updown-bins", "2"), dict(coverage_updown_bins=2)), (0.2, ("--no-consensus",), dict(no_consensus=True)),
]


def main():
    import genotype_util as gutil
    import golden_util as gu
    import ref_harness as rh
    from test_dropin_api import leads_of
    from sniffles_amd import leadprov, parallel, sv, synth
    args_in = [a for a in sys.argv[1:] if not a.startswith("--")]
    n_iter = int(args_in[0]) if args_in else 10
    seed0 = int(args_in[1]) if len(args_in) > 1 else 0
    if "--simt" in sys.argv:
        from emu import simt as E
    else:
        from emu import emu as E
    L = E.lib()
    bad = 0; n_t = 0; n_m = 0; t0 = time.time()
    for it in range(seed0, seed0 + n_iter):
        rng = np.random.default_rng([it, 67108863])
        args, kw = [], {}
        for p, frag, k in OPTIONS:
            if rng.random() < p:
                args += list(frag); kw.update(k)
        ti = synth.gen_fuzz(300000 + it, task_id=it % 4) if it % 3 else \
            synth.gen_task(it % 4, "chrG", 200_000, float(rng.choice([15, 40])), seed=it, site_density=2e-4, mosaic_frac=0.3)
        ref_run = rh.run_reference(ti, tuple(args))
        if "error" in ref_run:
            continue
        specs = gutil.target_specs(ref_run["candidates"], ti.contig_len, 800 + it)
        if not specs:
            continue
        exp = rh.run_reference_genotype(ti, specs, tuple(args))
        cfg = gu.make_config(kw, ti)
        lp = leadprov.LeadProvider(cfg, 0, ti.contig, contig_len=ti.contig_len)
        for ld in leads_of(ti):
            lp.record_lead(ld, int(ld.ref_start / cfg.cluster_binsize) * cfg.cluster_binsize)
        for s, e, hp in zip(ti.read_start.tolist(), ti.read_end.tolist(), ti.read_hp.tolist()):
            lp.record_read(s, e, hp)
        targets = gutil.make_targets(specs, sv.SVCall, sv.SVCallBNDInfo, sv.new_call)
        task = parallel.GenotypeTask(id=ti.task_id, sv_id=ti.sv_id_start, contig=ti.contig, start=0, end=ti.contig_len, config=cfg,
                                     lead_provider=lp, genotype_svs=targets)
        task.tandem_repeats = None if ti.tr_start is None else list(zip(ti.tr_start.tolist(), ti.tr_end.tolist()))
        diffs = []
        try:
            got = task.execute()
            task.close()
            if "error" in exp:
                diffs.append(f"the reference raised {exp['error']}")
            else:
                rec = gutil.result_records(got)
                want = exp["targets"]
                if len(rec) != len(want):
                    diffs.append(f"{len(rec)} targets, reference {len(want)}")
                for g, w in zip(rec, want):
                    if g != w:
                        diffs.append(str({k: (g.get(k), w.get(k)) for k in w if g.get(k) != w.get(k)})[:400]); break
                n_t += len(want); n_m += sum(1 for t in want if t["match"])
        except UnboundLocalError:
            if "error" not in exp:
                diffs.append("raised UnboundLocalError, the reference did not")
        except Exception as e:
            diffs.append(f"raised {type(e).__name__}: {str(e)[:300]}")
        if diffs:
            bad += 1
            print("MISMATCH it", it, " ".join(args), "|", diffs[:2], flush=True)
    print("ref_genotypefuzz: iterations", n_iter, "targets", n_t, "matched", n_m, "mismatching", bad, "seconds", round(time.time() - t0, 1))


if __name__ == "__main__":
    main()
