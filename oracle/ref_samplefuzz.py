"""TEST INFRASTRUCTURE ONLY (build container: needs /root/reference) - one sample end to end, BAM records in, VCF text out
(BASELINE.json configs[0] shape): sniffles_amd.pipeline.call_sample (extraction, clustering, calling, QC, genotyping,
consensus in the library - run here through its host builds -, VCF writer of this package) against the UNMODIFIED reference's
call_sample flow (oracle/ref_harness.py::run_reference_call_sample) on random synthetic samples under random command lines;
the VCF text is compared character by character.  tests/golden/sample_* pin six samples.
python oracle/ref_samplefuzz.py [n] [seed0] [--simt]
"""
import io
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import numpy as np

# (probability, command-line fragment)
OPTIONS = [
    (0.25, ("--mosaic",)), (0.2, ("--minsupport", "auto")), (0.15, ("--minsupport", "2")), (0.1, ("--minsupport", "6")),
    (0.15, ("--minsvlen", "30")), (0.1, ("--minsvlen", "120")), (0.15, ("--no-qc",)), (0.2, ("--qc-nm",)),
    (0.15, ("--cluster-binsize", "50")), (0.1, ("--cluster-binsize", "200")), (0.15, ("--cluster-merge-pos", "50")),
    (0.15, ("--long-ins-length", "600")), (0.15, ("--mapq", "0")), (0.1, ("--mapq", "61")), (0.15, ("--min-alignment-length", "300")),
    (0.15, ("--no-consensus",)), (0.1, ("--symbolic",)), (0.2, ("--output-rnames",)), (0.1, ("--dev-no-resplit",)),
    (0.1, ("--dev-no-resplit-repeat",)), (0.1, ("--cluster-merge-len", "0.6")), (0.1, ("--cluster-r", "1.0")),
    (0.1, ("--qc-coverage", "8")), (0.1, ("--max-splits-kb", "0.5")), (0.1, ("--max-splits-base", "1")), (0.1, ("--qc-stdev-abs-max", "20")),
    (0.1, ("--long-del-length", "1500")), (0.1, ("--long-dup-length", "1500")), (0.1, ("--mosaic-af-max", "0.4")),
]


def canon(text):
    """RNAMES is list(set_of_read_names) in the reference (sv.py:555): its order is the hash order of that process"""
    import re
    return re.sub(r"RNAMES=([^;\t\n]*)", lambda m: "RNAMES=" + ",".join(sorted(m.group(1).split(","))), text)


def main():
    import ref_harness as rh
    import vcf_util as vu
    from test_pipeline import config_for
    from test_vcf import assert_same_text
    from sniffles_amd import bam, pipeline, synth_bam
    args_in = [a for a in sys.argv[1:] if not a.startswith("--")]
    n_iter = int(args_in[0]) if args_in else 10
    seed0 = int(args_in[1]) if len(args_in) > 1 else 0
    if "--simt" in sys.argv:
        from emu import simt as E
    else:
        from emu import emu as E
    L = E.lib()
    bad = 0; n_rec = 0; n_reads = 0; n_regions = 0; t0 = time.time()
    for it in range(seed0, seed0 + n_iter):
        rng = np.random.default_rng([it, 8388607])
        args, seen = [], set()
        for p, frag in OPTIONS:
            if rng.random() < p and frag[0] not in seen:
                seen.add(frag[0]); args += list(frag)
        two = rng.random() < 0.3
        out = synth_bam.gen_sample(70000 + it, ref_names=("chr5", "chr6") if two else ("chr5",),
                                   ref_lens=(1_000_500, 1_020_000) if two else (1_000_500,), cov=float(rng.choice([6, 10, 16])),
                                   read_len_mean=int(rng.choice([6000, 12000])), site_spacing=int(rng.choice([9000, 18000])),
                                   err=float(rng.choice([0.005, 0.03])), split_spacing=int(rng.choice([0, 60000])),
                                   tr_frac=float(rng.choice([0.0, 0.3])))
        recs = bam.records_from_list(out[0], out[1], out[2])
        recs.tandem_repeats = out[3] if len(out) > 3 else None
        # --regions: a BED file of one to three intervals (overlapping ones included) on some of the contigs
        regions, bed = None, None
        if rng.random() < 0.35:
            import tempfile
            regions = {}
            lines = []
            for name, ln in zip(out[0], out[1]):
                if rng.random() < 0.7:
                    for _ in range(int(rng.integers(1, 4))):
                        a = int(rng.integers(0, ln - 50_000)); b = a + int(rng.integers(20_000, 600_000))
                        regions.setdefault(name, []).append((name, a, min(b, ln)))
                        lines.append(f"{name}\t{a}\t{min(b, ln)}")
            if regions:
                f = tempfile.NamedTemporaryFile("w", suffix=".bed", delete=False)
                f.write("\n".join(lines) + "\n"); f.close()
                bed = f.name
            else:
                regions = None
        ref_args = tuple(args) + (("--regions", bed) if bed else ())
        try:
            ref = rh.run_reference_call_sample(recs, ref_args, None, vu.FIXED)
        except SystemExit:
            continue
        except Exception as e:
            print("the reference raised", type(e).__name__, str(e)[:100], "on", " ".join(args), flush=True)
            continue
        buf = io.StringIO()
        diffs = []
        try:
            def own_cfg():
                c = config_for(args)
                c.regions_by_contig = regions or {}
                return c
            res = pipeline.call_sample(recs, own_cfg(), vcf_handle=buf, tandem_repeats=recs.tandem_repeats)
            if res.read_count != ref["read_count"]:
                diffs.append(f"read_count {res.read_count} != {ref['read_count']}")
            assert_same_text(canon(buf.getvalue()), canon(ref["vcf"]))
            n_rec += res.vcf_records; n_reads += res.read_count; n_regions += 1 if regions else 0
            # the same text straight from the record table, without SVCall objects (vcf.VCF.write_records)
            buf2 = io.StringIO()
            res2 = pipeline.call_sample(recs, own_cfg(), vcf_handle=buf2, tandem_repeats=recs.tandem_repeats, objects=False)
            assert buf2.getvalue() == buf.getvalue(), "record-table writer differs from the object path"
            assert res2.vcf_records == res.vcf_records and not res2.calls
        except AssertionError as e:
            diffs.append(str(e)[:500])
        except Exception as e:
            diffs.append(f"raised {type(e).__name__}: {str(e)[:300]}")
        if diffs:
            bad += 1
            print("MISMATCH it", it, " ".join(args), "regions", regions, "|", diffs[:2], flush=True)
    print("ref_samplefuzz: iterations", n_iter, "with --regions", n_regions, "reads", n_reads, "VCF records", n_rec, "mismatching", bad, "seconds", round(time.time() - t0, 1))


if __name__ == "__main__":
    main()
