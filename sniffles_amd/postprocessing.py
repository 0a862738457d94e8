"""`postprocessing.coverage(calls, lead_provider)` with the reference's signature (`src/sniffles/postprocessing.py:69-130`).

Inside `Task.call_candidates` the five coverage samples of every candidate are taken on the device as part of the
candidate stage; this entry point serves the other caller of the reference function - `GenotypeTask.execute`
(`parallel.py:353`), whose calls come from a VCF and not from this batch.  The samples are rank queries on the task's
read table in HBM (`snf_batch_coverage_calls`); there is no dense coverage vector and no CPU fallback.
`genotype_sv(svcall, config)` / `genotype_svs(svcalls, config)` serve the third caller: `CombineTask.execute` re-genotypes the
candidates of SNF files older than 2.5.3 (`--reqc`, `parallel.py:507-508`) - the likelihood arithmetic is the device function
the finalize kernels use (`snf_genotype_batch`), the strings (phase tuple, INFO PHASE) stay on the host.
The other functions of the reference module (`qc_sv`, `annotate_sv`, ...) run inside `Task.finalize_candidates` on the GPU
and have no Python counterpart.
"""
from __future__ import annotations

import numpy as np

from . import abi, lib
from .soa import SVT


def genotype_svs(svcalls, config, device: int = 0) -> None:
    """`postprocessing.genotype_sv(svcall, config)` (postprocessing.py:607-623) for a list of calls, one device launch."""
    svcalls = list(svcalls)
    if not svcalls:
        return
    rec = np.zeros(len(svcalls), abi.CALL_DTYPE)
    for i, c in enumerate(svcalls):
        cov = (c.coverage_upstream, c.coverage_start, c.coverage_center, c.coverage_end, c.coverage_downstream)
        if any(x is None for x in cov):
            raise NotImplementedError("genotype_sv: a coverage field is None (a call that never went through postprocessing.coverage)")
        r = rec[i]
        r["svtype"], r["svlen"], r["support"] = SVT[c.svtype], c.svlen, c.support
        r["support_sa"] = c.info.get("SUPPORT_SA", -1) if c.info.get("SUPPORT_SA") is not None else -1
        r["cov"] = cov
        r["filter"], r["qc"] = abi.FILTERS.index(c.filter), int(bool(c.qc))
        r["gt_set"], r["vaf"], r["ph_set"] = 0, np.nan, 0     # the phase strings are handled below, on the host
    lib.genotype_batch(config, rec, device=device)
    K = {n: k for k, n in enumerate(abi.CALL_DTYPE.names)}
    for c, r in zip(svcalls, rec.tolist()):
        c.filter, c.qc = abi.FILTERS[r[K["filter"]]], bool(r[K["qc"]])
        if r[K["gt_set"]]:
            try:
                phase = c.genotypes[0][5]        # Genotyper._get_phase (genotyping.py:74-81)
            except (KeyError, IndexError):
                phase = None
            a, b = r[K["gt_a"]], r[K["gt_b"]]
            c.genotypes[0] = (a, b, r[K["gt_gq"]], r[K["gt_dr"]], r[K["gt_dv"]], phase)
            c.info["VAF"] = r[K["vaf"]]
        # hom-alt calls keep their haplotype even if the phase filter failed (postprocessing.py:612-623)
        try:
            a, b, gq, dr, dv, phase = c.genotypes[0]
            phase_info = c.info.get("PHASE")
            if a == b and a == 1 and phase_info:
                hp, ps, hp_supp, ps_supp, hp_filt, ps_filt = phase_info.split(",")
                if "0" != hp:
                    hp_filt = "PASS"
                    c.genotypes[0] = (a, b, gq, dr, dv, (hp, ps))
                    c.info["PHASE"] = f"{hp},{ps},{hp_supp},{ps_supp},{hp_filt},{ps_filt}"
        except KeyError:
            pass


def genotype_sv(svcall, config, phase=None, device: int = 0) -> None:
    if phase is not None:
        raise NotImplementedError("genotype_sv with an explicit phase is the finalize stage's call (it runs on the GPU)")
    genotype_svs([svcall], config, device=device)


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
