"""Development sweep (build container only): seeded fuzz alignment records through the UNMODIFIED reference's
build_leadtab (over oracle/pysam_stub) and through the extraction kernels on the host emulation.
    python tools/dev/extract_sweep.py [first_seed] [n]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]


def main():
    import numpy as np
    import emu.emu as E
    E.lib()  # host tier: becomes the library sniffles_amd works on
    import extract_util as xu
    import ref_harness as rh
    from sniffles_amd import bam, extract, synth_bam
    from test_extract import DevCfg
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    variants = [((), {}), (("--mapq", "0", "--min-alignment-length", "200", "--dev-keep-lowqual-splits", "--max-splits-kb", "1.0"),
                           dict(mapq=0, min_alignment_length=200, dev_keep_lowqual_splits=True, max_splits_kb=1.0)),
                (("--minsvlen", "30", "--long-ins-length", "1000"), dict(minsvlen_screen=27, long_ins_length=1000)),
                (("--qc-nm",), dict(qc_nm=True))]
    bad = leads = errors = 0
    for seed in range(first, first + n):
        rng = np.random.default_rng(seed)
        names, lens, recs = synth_bam.gen_records(seed, int(rng.integers(150, 400)), sa_frac=float(rng.choice([0.1, 0.45, 0.8])),
                                                  read_len_mean=int(rng.choice([800, 2500, 6000])), phased=float(rng.choice([0.0, 0.5, 1.0])),
                                                  with_tags=bool(rng.random() < 0.9))
        R = bam.records_from_list(names, lens, recs)
        st, en = (0, 400000) if seed % 3 else (35000, 340000)
        args, kw = variants[seed % len(variants)]
        want = rh.run_reference_extract(R, "chrA", st, en, args, read_id_offset=seed * 1000)
        try:
            ti, info = extract.extract_region(R, "chrA", st, en, DevCfg(**kw), seed * 1000)
            got_err = None
        except Exception as e:
            got_err = str(e)
        if "error" in want:
            ok = got_err is not None
            errors += 1
        elif got_err is not None:
            ok = False
        else:
            try:
                reads = list(zip(ti.read_start.tolist(), ti.read_end.tolist(), ti.read_hp.tolist()))
                xu.check_against_golden(want, xu.canon_leads(ti), reads, float(ti.qc_nm_threshold).hex(), info.read_id, ti.contig_len)
                ok = True
                leads += ti.n_leads
            except AssertionError:
                ok = False
        bad += not ok
        print(seed, args[:2], R.n, "records", "reference raised " + want["error"] if "error" in want else f"{len(want['leads'])} leads", "ok" if ok else "DIFF", flush=True)
    print("leads compared:", leads, "runs the reference raised on:", errors, "mismatching runs:", bad)


if __name__ == "__main__":
    main()
