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
