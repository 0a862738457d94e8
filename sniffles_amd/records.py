"""Canonical per-call records from a fetched result (host-side string formatting).

The same dict layout is produced by the test harness (under oracle/) from the
reference's own `SVCall` objects, so parity tests compare like with like.
"""
from __future__ import annotations

from .abi import FILTERS, Result, none_if_nan
from .soa import SVTYPES


def call_id(svtype: str, sv_id: int, task_id: int) -> str:
    """sv.py:563: f"{svtype}.{task.sv_id:X}S{task.id:X}" """
    return f"{svtype}.{sv_id:X}S{task_id:X}"


def bnd_alt(mate_contig: str, mate_ref_start: int, is_first: bool, is_reverse: bool) -> str:
    """sv.py:630-634"""
    br = "]" if is_reverse else "["
    return ("N" if is_first else "") + br + f"{mate_contig}:{mate_ref_start}" + br + ("N" if not is_first else "")


def _ps_str(code: int, ti):
    if code == -1:
        return None
    if code == -2:
        return "NULL"
    return ti.ps_name(int(code))


def record(res: Result, i: int, tasks, stage: str) -> dict:
    c = res.calls[i]
    ti = tasks[int(c["task_index"])]
    svtype = SVTYPES[int(c["svtype"])]
    alt = res.alt(i)
    bnd = None
    if svtype == "BND":
        bnd = [ti.contig_name(int(c["mate_contig"])), int(c["mate_ref_start"]), bool(c["bnd_is_first"]),
               bool(c["bnd_is_reverse"])]
        alt = bnd_alt(*bnd)
    elif alt is None:
        alt = f"<{svtype}>"
    rec = dict(
        id=call_id(svtype, int(c["sv_id"]), ti.task_id), contig=ti.contig, pos=int(c["pos"]), end=int(c["end"]),
        svtype=svtype, svlen=int(c["svlen"]), support=int(c["support"]), qual=int(c["qual"]),
        precise=bool(c["precise"]), fwd=int(c["fwd"]), rev=int(c["rev"]), filter=FILTERS[int(c["filter"])],
        qc=bool(c["qc"]), nm=float(c["nm"]), alt=alt,
        stdev_pos=float(c["stdev_pos"]), stdev_len=none_if_nan(c["stdev_len"]),
        support_long=None if c["support_long"] < 0 else int(c["support_long"]),
        support_sa=None if c["support_sa"] < 0 else int(c["support_sa"]),
        cov=[int(x) for x in c["cov"]],
        rnames=sorted(ti.qname(int(q)) for q in res.rn(i)),
        bnd=bnd,
    )
    if stage == "final":
        if c["gt_set"]:
            hp = None if c["gt_hp"] < 0 else str(int(c["gt_hp"]))
            rec["gt"] = [int(c["gt_a"]), int(c["gt_b"]), int(c["gt_gq"]), int(c["gt_dr"]), int(c["gt_dv"]),
                         [hp, _ps_str(int(c["gt_ps"]), ti)]]
        else:
            rec["gt"] = None
        rec["vaf"] = none_if_nan(c["vaf"])
        if c["ph_set"]:
            rec["phase"] = (f"{int(c['ph_hp'])},{_ps_str(int(c['ph_ps']), ti)},{int(c['ph_hp_support'])},"
                            f"{int(c['ph_ps_support'])},{'PASS' if c['ph_hp_pass'] else 'FAIL'},"
                            f"{'PASS' if c['ph_ps_pass'] else 'FAIL'}")
        else:
            rec["phase"] = None
    return rec


def records(res: Result, tasks, stage: str) -> list:
    """Per task: list of records or {'error': name} (reference failure modes)."""
    out = []
    for t in range(len(tasks)):
        if int(res.task_status[t]) == 1:
            out.append(dict(error="UnboundLocalError"))
            continue
        lo, hi = int(res.task_call_off[t]), int(res.task_call_off[t + 1])
        out.append([record(res, i, tasks, stage) for i in range(lo, hi)])
    return out


def diff_results(got: Result, gt: int, exp: Result, et: int, limit: int = 5) -> list:
    """Vectorised bit-exact comparison of task `gt` of `got` with task `et` of `exp` (whole-genome sized results: no
    per-call Python objects).  Every field of the call records is compared (NaN == NaN for the float fields that use
    NaN as "absent"), the ALT bytes and the supporting read ids (as a set per call: `rnames` is `list(set)` in the
    reference), the task status and coverage_average_total.  Returns human-readable differences (empty = identical)."""
    import numpy as np
    out = []
    sg, se = int(got.task_status[gt]), int(exp.task_status[et])
    if sg != se:
        return [f"task status {sg} != {se}"]
    if sg != 0:
        return out
    cg, ce = float(got.coverage_average_total[gt]), float(exp.coverage_average_total[et])
    if not (cg == ce or (cg != cg and ce != ce)):
        out.append(f"coverage_average_total {cg!r} != {ce!r}")
    a = got.calls[int(got.task_call_off[gt]):int(got.task_call_off[gt + 1])]
    b = exp.calls[int(exp.task_call_off[et]):int(exp.task_call_off[et + 1])]
    if a.shape[0] != b.shape[0]:
        return out + [f"call count {a.shape[0]} != {b.shape[0]}"]
    n = a.shape[0]
    if n == 0:
        return out
    bad = np.zeros(n, bool)
    names_bad = []
    for name in a.dtype.names:
        if name in ("task_index", "alt_off", "rn_off"):
            continue
        x, y = a[name], b[name]
        eq = x == y
        if name == "cluster_seed_index":       # -1: not provided (occupancy prefilter on, see include/sniffles_amd.h)
            eq |= (x == -1) | (y == -1)
        if x.dtype.kind == "f":
            eq |= np.isnan(x) & np.isnan(y)
        if eq.ndim > 1:
            eq = eq.all(axis=1)
        if not eq.all():
            names_bad.append(f"{name} ({int((~eq).sum())} calls, first at {int(np.argmin(eq))})")
            bad |= ~eq
    if names_bad:
        out.append("fields differ: " + ", ".join(names_bad[:limit]))
        return out   # lengths may differ: the byte comparisons below assume equal alt_len / rn_len

    def gather(pool, off, ln):
        ln = np.maximum(ln.astype(np.int64), 0)
        tot = int(ln.sum())
        seg = np.repeat(np.arange(n, dtype=np.int64), ln)
        first = np.cumsum(ln) - ln
        idx = np.repeat(off.astype(np.int64), ln) + (np.arange(tot, dtype=np.int64) - np.repeat(first, ln))
        return pool[idx], seg

    ba, seg = gather(got.alt_pool, a["alt_off"], a["alt_len"])
    bb, _ = gather(exp.alt_pool, b["alt_off"], b["alt_len"])
    ne = ba != bb
    if ne.any():
        calls_bad = np.unique(seg[ne])
        out.append(f"ALT bytes differ in {calls_bad.shape[0]} calls, first call {int(calls_bad[0])} (svtype {int(a['svtype'][calls_bad[0]])}, pos {int(a['pos'][calls_bad[0]])})")
    ra, seg = gather(got.rnames, a["rn_off"], a["rn_len"])
    rb, _ = gather(exp.rnames, b["rn_off"], b["rn_len"])
    ra = ra[np.lexsort((ra, seg))]
    rb = rb[np.lexsort((rb, seg))]
    ne = ra != rb
    if ne.any():
        out.append(f"supporting reads differ in {np.unique(seg[ne]).shape[0]} calls")
    return out
