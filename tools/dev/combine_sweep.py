"""Development sweep (build container only): populations of synthetic samples merged with different `--combine-*` settings
by the UNMODIFIED reference (own .snf files) and by sniffles_amd (own .snf files, host emulation); merged VCF text must
be equal.    python tools/dev/combine_sweep.py [first_seed] [n]"""
import io
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]

FLAGS = [(), ("--combine-output-filtered",), ("--combine-pair-relabel",), ("--combine-separate-intra",),
         ("--combine-high-confidence", "0.5", "--combine-low-confidence", "0.6"), ("--combine-null-min-coverage", "20"),
         ("--combine-support-threshold", "5"), ("--combine-match", "100", "--combine-match-max", "300"), ("--combine-pctseq", "0.9"),
         ("--combine-low-confidence-abs", "3", "--combine-pair-relabel", "--combine-pair-relabel-threshold", "10"),
         ("--combine-close-handles",), ("--combine-pctseq", "0")]


def main():
    import numpy as np
    import emu.emu as E
    E.lib()  # host tier: becomes the library sniffles_amd works on
    import ref_harness as rh
    import vcf_util as vu
    from sniffles_amd import bam, pipeline, synth_bam
    from test_pipeline import config_for
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    n = int(sys.argv[2]) if len(sys.argv) > 2 else len(FLAGS)
    bad = 0
    for seed in range(first, first + n):
        rng = np.random.default_rng(seed)
        args = FLAGS[seed % len(FLAGS)]
        ns = int(rng.integers(2, 7))
        recs = []
        for s in range(ns):
            out = synth_bam.gen_sample(seed * 10 + s, ref_names=("chrP",), ref_lens=(1_000_000,), cov=float(rng.choice([8, 14, 24])),
                                       site_seed=seed, site_spacing=int(rng.choice([5000, 9000, 20000])), split_spacing=int(rng.choice([0, 110000])))
            recs.append(bam.records_from_list(out[0], out[1], out[2]))
        d = tempfile.mkdtemp(prefix="cmb_")
        want = rh.run_reference_population(recs, d, args, vu.FIXED)
        paths = []
        for s, r in enumerate(recs):
            p = os.path.join(d, "ours", f"sample{s}.snf")
            os.makedirs(os.path.dirname(p), exist_ok=True)
            pipeline.call_sample(r, config_for(()), snf_path=p)
            paths.append(p)
        buf = io.StringIO()
        pipeline.combine(paths, config_for(args), vcf_handle=buf)
        ok = buf.getvalue() == want["vcf"]
        print(f"seed {seed} {ns} samples {args}: {len(vu.split_text(want['vcf'])[1])} merged records  {'ok' if ok else 'DIFF'}", flush=True)
        if not ok:
            bad += 1
            for a, b in zip(buf.getvalue().split("\n"), want["vcf"].split("\n")):
                if a != b:
                    print("  ours:", a[:400]); print("  ref: ", b[:400]); break
    print("mismatches:", bad)


if __name__ == "__main__":
    main()
