"""Development sweep (build container only): force calling with random target lists against the live reference.
    python tools/dev/genotype_sweep.py [n_seeds]"""
import gzip
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]


def main():
    import cases
    import emu.emu as E
    E.lib()  # host tier: becomes the library sniffles_amd works on
    import genotype_util as gutil
    import ref_harness as rh
    from test_genotype import make_task
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    bad = tot = 0
    for name in ("chr21_30x_mosaic", "fuzz_4_2", "bnd_stale_end", "chr22_60x_hifi", "fuzz_0_0", "fuzz_9_0", "merge_inner"):
        build, kw, args = cases.ALL[name]
        ti = build()
        with gzip.open(os.path.join(ROOT, "tests", "golden", name + ".json.gz"), "rb") as f:
            exp = json.loads(f.read().decode())["expected"]
        if "candidates" not in exp:
            continue
        for seed in range(n):
            specs = gutil.target_specs(exp["candidates"], ti.contig_len, 9000 + seed)
            want = rh.run_reference_genotype(ti, specs, args)
            _, task = make_task(name, specs, E.lib())
            try:
                got = dict(targets=gutil.result_records(task.execute()))
            except UnboundLocalError:
                got = dict(error="UnboundLocalError")
            task.close()
            ok = got == json.loads(json.dumps(want))
            tot += len(specs)
            bad += not ok
            print(name, seed, len(specs), "ok" if ok else "DIFF", flush=True)
    print("targets:", tot, "mismatching runs:", bad)


if __name__ == "__main__":
    main()
