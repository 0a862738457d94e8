"""TEST INFRASTRUCTURE ONLY (build container: needs /root/reference) - a population end to end (BASELINE.json configs[4]
shape): per-sample BAM records -> .snf files -> merged multi-sample VCF.  sniffles_amd.pipeline.call_sample + pipeline.combine
(library through its host builds; SNF container, columnar candidate store, grouping / SVGroup.call kernels, VCF writer of this
package) against the UNMODIFIED reference (oracle/ref_harness.py::run_reference_population: its call_sample flow per sample, its
SNF files, its CombineTask.execute, its VCF writer) on random populations and random --combine-* command lines; the merged VCF
text is compared character by character.  tests/golden/population_* pin two populations with default options.
python oracle/ref_populationfuzz.py [n] [seed0] [--simt]
"""
import io
import os
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import numpy as np

OPTIONS = [
    (0.2, ("--combine-match", "50")), (0.2, ("--combine-match", "1000")), (0.2, ("--combine-match-max", "300")),
    (0.15, ("--combine-separate-intra",)), (0.2, ("--combine-pctseq", "0.0")), (0.15, ("--combine-pctseq", "0.9")),
    (0.2, ("--combine-high-confidence", "0.4")), (0.2, ("--combine-low-confidence", "0.4")), (0.2, ("--combine-low-confidence-abs", "4")),
    (0.2, ("--combine-null-min-coverage", "12")), (0.25, ("--combine-output-filtered",)), (0.2, ("--combine-support-threshold", "1")),
    (0.15, ("--combine-support-threshold", "5")), (0.2, ("--combine-pair-relabel",)), (0.15, ("--combine-pair-relabel-threshold", "5")),
    (0.2, ("--dev-combine-medians",)), (0.15, ("--minsvlen", "100")), (0.15, ("--cluster-binsize", "50")),
    (0.15, ("--cluster-binsize-combine-mult", "1")), (0.15, ("--no-qc",)),
]


def main():
    import ref_harness as rh
    import vcf_util as vu
    from test_pipeline import config_for
    from test_vcf import assert_same_text
    from sniffles_amd import bam, pipeline, synth_bam
    args_in = [a for a in sys.argv[1:] if not a.startswith("--")]
    n_iter = int(args_in[0]) if args_in else 5
    seed0 = int(args_in[1]) if len(args_in) > 1 else 0
    if "--simt" in sys.argv:
        from emu import simt as E
    else:
        from emu import emu as E
    L = E.lib()
    bad = 0; n_rec = 0; n_samples = 0; t0 = time.time()
    for it in range(seed0, seed0 + n_iter):
        rng = np.random.default_rng([it, 33554431])
        args, seen = [], set()
        for p, frag in OPTIONS:
            if rng.random() < p and frag[0] not in seen:
                seen.add(frag[0]); args += list(frag)
        ns = int(rng.integers(2, 6))
        spacing = int(rng.choice([7000, 14000])); split = int(rng.choice([0, 90000])); two = rng.random() < 0.3
        recs_list = []
        for s in range(ns):
            out = synth_bam.gen_sample(90000 + 11 * it + s, ref_names=("chr3", "chr4") if two else ("chr3",),
                                       ref_lens=(1_000_000, 1_000_700) if two else (1_000_000,), cov=float(rng.choice([6, 10, 18])),
                                       read_len_mean=9000, site_spacing=spacing, split_spacing=split, site_seed=4000 + it)
            r = bam.records_from_list(out[0], out[1], out[2])
            r.tandem_repeats = out[3] if len(out) > 3 else None
            recs_list.append(r)
        diffs = []
        with tempfile.TemporaryDirectory() as wd:
            try:
                ref = rh.run_reference_population(recs_list, wd, tuple(args), vu.FIXED)
            except SystemExit:
                continue
            except Exception as e:
                print("the reference raised", type(e).__name__, str(e)[:100], "on", " ".join(args), flush=True)
                continue
            try:
                paths = []
                for s, r in enumerate(recs_list):
                    path = os.path.join(wd, f"own{s}.snf")
                    pipeline.call_sample(r, config_for(()), snf_path=path, tandem_repeats=r.tandem_repeats)
                    paths.append(path)
                # the sample ids in the header come from the file names: same basenames as the reference's files
                renamed = []
                od = os.path.join(wd, "own"); os.mkdir(od)
                for s, pth in enumerate(paths):
                    q = os.path.join(od, f"sample{s}.snf"); os.rename(pth, q); renamed.append(q)
                buf = io.StringIO()
                pipeline.combine(renamed, config_for(args), vcf_handle=buf)
                assert_same_text(buf.getvalue(), ref["vcf"])
                n_rec += len(vu.split_text(ref["vcf"])[1]); n_samples += ns
                # and over the reference's own files
                buf2 = io.StringIO()
                pipeline.combine(ref["snf"], config_for(args), vcf_handle=buf2)
                assert_same_text(buf2.getvalue(), ref["vcf"])
                # the same merge straight from the group table (no SVCall objects; records formatted by threads when every column is
                # there), into a text handle and into a text file over a binary buffer: the text of the object path
                buf3 = io.StringIO()
                pipeline.combine(renamed, config_for(args), vcf_handle=buf3, objects=False)
                assert buf3.getvalue() == buf.getvalue(), "text path (objects=False) differs from the object path"
                raw = io.BytesIO(); h4 = io.TextIOWrapper(raw, encoding="utf-8", newline="")
                os.environ["SNF_TEXT_THREADS"] = "3"
                try:
                    pipeline.combine(renamed, config_for(args), vcf_handle=h4, objects=False)
                finally:
                    os.environ.pop("SNF_TEXT_THREADS", None)
                h4.flush()
                assert raw.getvalue().decode("utf-8") == buf.getvalue(), "text path into a binary-backed handle differs from the object path"
            except AssertionError as e:
                diffs.append(str(e)[:500])
            except Exception as e:
                diffs.append(f"raised {type(e).__name__}: {str(e)[:300]}")
        if diffs:
            bad += 1
            print("MISMATCH it", it, ns, "samples", " ".join(args), "|", diffs[:2], flush=True)
    print("ref_populationfuzz: iterations", n_iter, "samples", n_samples, "merged VCF records", n_rec, "mismatching", bad, "seconds", round(time.time() - t0, 1))


if __name__ == "__main__":
    main()
