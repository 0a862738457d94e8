"""Helpers shared by the parity tests: load fixtures, rebuild inputs, diff records."""
import gzip
import hashlib
import json
import os

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with gzip.open(os.path.join(GOLDEN_DIR, name + ".json.gz"), "rb") as f:
        return json.loads(f.read().decode())


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


def make_config(kw, ti):
    from sniffles_amd.config import SnifflesConfig
    cfg = SnifflesConfig(**kw)
    cfg.qc_nm_threshold = ti.qc_nm_threshold
    cfg.average_regional_nm = ti.qc_nm_threshold
    return cfg


def diff_records(got, exp, limit=3):
    """Return a list of human-readable differences between two record lists (bit-exact comparison;
    every field incl. float STDEV/VAF/nm and the INS consensus ALT must be identical)."""
    out = []
    if len(got) != len(exp):
        out.append(f"record count {len(got)} != {len(exp)}")
    for g, e in zip(got, exp):
        if g != e:
            d = {k: (g.get(k), e.get(k)) for k in e if g.get(k) != e.get(k)}
            for k in ("alt", "rnames"):
                if k in d:
                    d[k] = "differs"
            out.append(f"{e['id']}: {d}")
            if len(out) >= limit:
                break
    return out
