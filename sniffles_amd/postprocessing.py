"""`postprocessing.coverage(calls, lead_provider)` with the reference's signature (`src/sniffles/postprocessing.py:69-130`).

Inside `Task.call_candidates` the five coverage samples of every candidate are taken on the device as part of the
candidate stage; this entry point serves the other caller of the reference function - `GenotypeTask.execute`
(`parallel.py:353`), whose calls come from a VCF and not from this batch.  The samples are rank queries on the task's
read table in HBM (`snf_batch_coverage_calls`); there is no dense coverage vector and no CPU fallback.
The other functions of the reference module (`qc_sv`, `annotate_sv`, `genotype_sv`, ...) run inside
`Task.finalize_candidates` on the GPU and have no Python counterpart.
"""
from __future__ import annotations

from .soa import SVT


def coverage(calls, lead_provider) -> float:
    """Annotates `coverage_upstream / start / center / end / downstream` of the calls and returns `coverage.mean()`.
    `lead_provider`: the task's lead provider after `Task.call_candidates` (its read table is on the device)."""
    batch = getattr(lead_provider, "device_batch", None)
    if batch is None:
        raise RuntimeError("postprocessing.coverage needs the task's device batch: call Task.call_candidates first "
                           "(the coverage lives on the GPU; there is no CPU fallback)")
    other = SVT["DEL"]     # every type but INS / BND takes end = pos + abs(svlen)
    codes = [SVT[c.svtype] if c.svtype in ("INS", "BND") else other for c in calls]
    first = [1 if (c.svtype == "BND" and c.bnd_info.is_first) else 0 for c in calls]
    cov = [[c.coverage_upstream, c.coverage_start, c.coverage_center, c.coverage_end, c.coverage_downstream] for c in calls]
    if any(x is None for row in cov for x in row):
        raise TypeError("coverage fields must be integers (the reference initialises them to 0, sv.py:117-121)")
    out, status, mean = batch.coverage_calls(0, codes, [c.pos for c in calls], [c.svlen for c in calls], first, cov)
    for c, row in zip(calls, out.tolist()):
        (c.coverage_upstream, c.coverage_start, c.coverage_center, c.coverage_end, c.coverage_downstream) = row
    if status == 1:        # a BND before any other call: `end` is unbound in the reference's loop
        raise UnboundLocalError("local variable 'end' referenced before assignment")
    return mean
