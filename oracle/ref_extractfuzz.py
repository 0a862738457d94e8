"""TEST INFRASTRUCTURE ONLY (build container: needs /root/reference) - signature extraction (sniffles_amd.extract.extract_region:
the extraction kernels, run here through the host builds of the library - thread form and, with --simt, the wave form) and
its oracle (oracle/extract_oracle.py) against the UNMODIFIED reference's build_leadtab (over oracle/pysam_stub.py) on random
record tables, regions and read filters.  tests/golden/extract_* pin eleven cases.   python oracle/ref_extractfuzz.py [n] [seed0] [--simt]
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import numpy as np


class DevCfg:
    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)


def main():
    import extract_oracle as eo
    import extract_util as xu
    import ref_harness as rh
    from sniffles_amd import bam, extract, synth_bam
    args_in = [a for a in sys.argv[1:] if not a.startswith("--")]
    n_iter = int(args_in[0]) if args_in else 20
    seed0 = int(args_in[1]) if len(args_in) > 1 else 0
    if "--simt" in sys.argv:
        from emu import simt as E
    else:
        from emu import emu as E
    L = E.lib()
    bad = 0; n_leads = 0; n_reads = 0; t0 = time.time()
    for it in range(seed0, seed0 + n_iter):
        rng = np.random.default_rng([it, 2147483647])
        ci = int(rng.choice([0, 0, 0, 2]))
        gen = dict(seed=50000 + it, n_reads=int(rng.integers(50, 500)), sa_frac=float(rng.choice([0.0, 0.25, 0.6])), contig_index=ci,
                   read_len_mean=int(rng.choice([1500, 3000, 9000])), with_tags=bool(rng.random() < 0.85),
                   phased=float(rng.choice([0.0, 0.5, 1.0])), style=str(rng.choice(["fuzz", "fuzz", "ont"])))
        names, lens, rl = synth_bam.gen_records(**gen)
        recs = bam.records_from_list(names, lens, rl)
        contig, clen = names[ci], lens[ci]
        args, cfg, overrides = [], {}, {}
        def opt(p): return rng.random() < p
        if opt(0.4): v = int(rng.choice([0, 10, 40])); args += ["--mapq", str(v)]; cfg["mapq"] = v
        if opt(0.4): v = int(rng.choice([100, 500, 3000])); args += ["--min-alignment-length", str(v)]; cfg["min_alignment_length"] = v
        if opt(0.3): v = int(rng.choice([30, 100])); args += ["--minsvlen", str(v)]; cfg["minsvlen_screen"] = int(0.9 * v)
        if opt(0.3): v = int(rng.choice([300, 1000])); args += ["--long-ins-length", str(v)]; cfg["long_ins_length"] = v
        if opt(0.3): v = int(rng.choice([1, 7])); args += ["--max-splits-base", str(v)]; cfg["max_splits_base"] = v
        if opt(0.3): v = float(rng.choice([0.7, 0.01])); args += ["--max-splits-kb", str(v)]; cfg["max_splits_kb"] = v
        if opt(0.3): args += ["--dev-keep-lowqual-splits"]; cfg["dev_keep_lowqual_splits"] = True
        if opt(0.3): v = int(rng.choice([1536, 16, 2048])); args += ["--exclude-flags", str(v)]; cfg["exclude_flags"] = v
        if opt(0.3): v = int(rng.choice([400, 60])); args += ["--dev-seq-cache-maxlen", str(v)]; cfg["dev_seq_cache_maxlen"] = v
        if opt(0.25): args += ["--detect-large-ins", "False"]; cfg["detect_large_ins"] = False
        if opt(0.2): overrides["phase"] = False; cfg["advanced_tags"] = False
        if opt(0.5):
            st = int(rng.integers(0, clen // 2)); en = int(rng.integers(st + 1000, clen + 1))
        else:
            st, en = 0, clen
        rid0 = int(rng.choice([0, 7, 123456]))
        ref = rh.run_reference_extract(recs, contig, st, en, tuple(args), rid0, overrides)
        diffs = []
        try:
            out = eo.extract_region(recs.blob, recs.rec_off, recs.ref_names, contig, st, en, eo.Cfg(**cfg), rid0)
            if "error" in ref:
                diffs.append(f"oracle: the reference raised {ref['error']}")
            else:
                xu.check_against_golden(ref, out["rows"], out["reads"], out["qc_nm_threshold"], out["read_id"], clen)
        except AssertionError as e:
            diffs.append("oracle: " + str(e)[:300])
        except Exception as e:
            if "error" not in ref:
                diffs.append(f"oracle raised {type(e).__name__}: {e}")
        try:
            ti, info = extract.extract_region(recs, contig, st, en, DevCfg(**cfg), rid0)
            if "error" in ref:
                diffs.append(f"kernels: the reference raised {ref['error']}")
            else:
                reads = list(zip(ti.read_start.tolist(), ti.read_end.tolist(), ti.read_hp.tolist()))
                xu.check_against_golden(ref, xu.canon_leads(ti), reads, float(ti.qc_nm_threshold).hex(), info.read_id, ti.contig_len)
                n_leads += ti.n_leads; n_reads += len(reads)
        except AssertionError as e:
            diffs.append("kernels: " + str(e)[:300])
        except Exception as e:
            if "error" not in ref:
                diffs.append(f"kernels raised {type(e).__name__}: {str(e)[:200]}")
        if diffs:
            bad += 1
            print("MISMATCH it", it, gen, (st, en), rid0, " ".join(args), overrides, "|", diffs[:2], flush=True)
    print("ref_extractfuzz: iterations", n_iter, "reads accepted", n_reads, "leads", n_leads, "mismatching", bad, "seconds", round(time.time() - t0, 1))


if __name__ == "__main__":
    main()
