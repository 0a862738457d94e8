"""Per-record time of the extraction passes (instrumented build: bash tools/build_variant.sh xtrace -DSNF_XTRACE).

    SNF_LIB_SO=variants/xtrace.so python tools/xtrace.py [--pass count|emit] [--sa-frac 0.2]

Every accepted record leaves {start, total, clips + tags, SA section} in wall-clock ticks (10 ns); printed: the span of the
pass, how busy the device's wave slots were over it, the slowest records with what they hold.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=3000)
    ap.add_argument("--tile", type=int, default=8)
    ap.add_argument("--sa-frac", type=float, default=0.2)
    ap.add_argument("--pass", dest="which", default="count")
    a = ap.parse_args()
    out = "/tmp/xtrace.bin"
    os.environ["SNF_XTRACE_OUT"] = out
    os.environ["SNF_XTRACE_PASS"] = a.which
    from sniffles_amd import bam, extract, synth_bam
    names, lens, recs = synth_bam.gen_records(2026, a.reads, style="ont", read_len_mean=20000, sa_frac=a.sa_frac,
                                              ref_lens=(60_000_000, 300000, 300000, 100000))
    R = bam.records_from_list(names, lens, recs * a.tile)
    x = extract.Extractor()
    x.upload(R, "chrA", 0, 60_000_000)
    for _ in range(3):
        x.run()
    _, info = x.result()
    t = np.fromfile(out, np.uint32).reshape(-1, 4).astype(np.int64)
    ok = t[:, 1] > 0
    start = (t[:, 0] - t[ok, 0].min()) & 0xffffffff
    end = start + t[:, 1]
    span = int(end[ok].max())
    print(f"pass {a.which}: kernel {info.ms_count if a.which == 'count' else info.ms_emit:.3f} ms; stamped records {int(ok.sum())}, span {span / 100:.1f} us, "
          f"sum of record times {t[ok, 1].sum() / 100:.0f} us = {t[ok, 1].sum() / max(span, 1):.0f} waves busy on average")
    for lo in range(0, span, max(span // 12, 1)):
        hi = lo + max(span // 12, 1)
        live = int((ok & (start < hi) & (end > lo)).sum())
        print(f"  {lo / 100:7.1f} .. {hi / 100:7.1f} us: records in flight at some point {live}")
    q = np.percentile(t[ok, 1], [50, 90, 99, 100]) / 100
    print("record time us: median %.1f  p90 %.1f  p99 %.1f  max %.1f" % tuple(q))
    has_sa = np.array([b"SAZ" in r for r in recs] * a.tile)
    for name, m in (("with SA", ok & has_sa), ("without SA", ok & ~has_sa)):
        if m.any():
            print(f"  {name}: n {int(m.sum())}, mean total {t[m, 1].mean() / 100:.1f} us, mean clips+tags {t[m, 2].mean() / 100:.1f}, mean SA section {t[m, 3].mean() / 100:.1f}, max {t[m, 1].max() / 100:.1f}")
    print("slowest records: index  total us  clips+tags us  SA us  start us  n_cigar  aux bytes  SA elements  flag")
    for i in np.argsort(-t[:, 1])[:12]:
        r = recs[i % len(recs)]
        n_cig = int.from_bytes(r[16:18], "little"); l_seq = int.from_bytes(r[20:24], "little"); flag = int.from_bytes(r[18:20], "little")
        aux = len(r) - (36 + r[12] + 4 * n_cig + (l_seq + 1) // 2 + l_seq)
        k = r.find(b"SAZ")
        n_sa = r[k:r.index(b"\0", k)].count(b";") if k >= 0 else 0
        print(f"  {i:6d} {t[i, 1] / 100:8.1f} {t[i, 2] / 100:8.1f} {t[i, 3] / 100:8.1f} {start[i] / 100:8.1f} {n_cig:7d} {aux:6d} {n_sa:3d} {flag:#06x}")


if __name__ == "__main__":
    main()
