"""TEST INFRASTRUCTURE ONLY - regenerates tests/golden/*.json.gz from the UNMODIFIED reference.

Run in the build container (needs /root/reference):  python oracle/make_golden.py
Each fixture holds: the case name, a SHA-256 over the input arrays (so the test notices if the
seeded generators drift), the reference's candidate-stage and final-stage records, and
coverage_average_total.  Inputs are rebuilt at test time from tests/cases.py.
"""
from __future__ import annotations

import gzip
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def input_sha(ti) -> str:
    h = hashlib.sha256()
    for name in sorted(ti.leads):
        h.update(name.encode())
        h.update(ti.leads[name].tobytes())
    for a in (ti.seq_pool, ti.read_start, ti.read_end, ti.read_hp):
        h.update(a.tobytes())
    if ti.tr_start is not None:
        h.update(ti.tr_start.tobytes())
        h.update(ti.tr_end.tobytes())
    h.update(repr((ti.task_id, ti.contig, ti.contig_len, ti.sv_id_start, ti.qc_nm_threshold)).encode())
    return h.hexdigest()


def main(names=None):
    import cases
    import ref_harness as rh
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    for name, (build, kw, args) in cases.ALL.items():
        if names and name not in names:
            continue
        ti = build()
        ref = rh.run_reference(ti, args)
        doc = dict(case=name, config=kw, reference_args=list(args), input_sha=input_sha(ti), expected=ref)
        path = os.path.join(out_dir, name + ".json.gz")
        with gzip.GzipFile(path, "wb", mtime=0) as f:
            f.write(json.dumps(doc, sort_keys=True, separators=(",", ":")).encode())
        n = "error:" + ref["error"] if "error" in ref else f"{len(ref['candidates'])} cand / {len(ref['final'])} final"
        print(f"{name:32s} {ti.n_leads:7d} leads  {n}")




def main_combine(names=None):
    import cases
    import ref_harness as rh
    out_dir = os.path.join(ROOT, "tests", "golden")
    for name, (build, args) in cases.COMBINE.items():
        if names and name not in names:
            continue
        tis = build()
        ref = rh.run_reference_combine(tis, args)
        doc = dict(case=name, reference_args=list(args), input_sha=[input_sha(t) for t in tis], expected=ref)
        with gzip.GzipFile(os.path.join(out_dir, name + ".json.gz"), "wb", mtime=0) as f:
            f.write(json.dumps(doc, sort_keys=True, separators=(",", ":")).encode())
        print(f"{name:32s} {len(tis)} samples  {sum(len(p['cands']) for p in ref['problems'])} cands  "
              f"{sum(len(p['groups']) for p in ref['problems'])} groups  {sum(len(c['calls']) for c in ref['calls'])} calls")


if __name__ == "__main__":
    sel = set(sys.argv[1:])
    main(sel)
    main_combine(sel)
