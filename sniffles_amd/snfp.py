"""The edit-distance call site of the population SNF (`src/sniffles/snfp.py:91-107`): does a call match a
population variant?  `PopulationVariant.match` gates on position / length like the merge (`combine_match`,
`combine_match_max`) and, for insertions, on `edlib.align(self.alt, svcall.alt)['editDistance']`; here every alignment of a
query is one entry of ONE `snf_edit_distance_batch` launch (banded by the cut-off the gate implies).

Only the matching is served - the population file itself (`PopulationSNF`: header, blocks, `store`) is container I/O outside the
hot path.  The record below has the reference's fields, so objects unpickled from a reference-written population SNF work as they
are (anything with `pos`, `svlen`, `svtype`, `alt`).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

from . import lib


@dataclass
class PopulationVariant:
    contig: str
    pos: int
    id: str
    alt: str
    svtype: str
    svlen: int
    end: int
    af: float
    genotyped_sample_count: int
    variant_sample_count: int

    @staticmethod
    def _calculate_frequency(genotypes: dict, ploidy: int = 2):
        """(population AF, genotyped samples, samples carrying the SV) from {sample id: genotype tuple} (snfp.py:40-64)."""
        total = variant = genotyped = carrying = 0
        for gt in genotypes.values():
            if gt[0] == '.':
                continue
            genotyped += 1
            n = gt[0] + gt[1]
            total += ploidy
            variant += n
            if n > 0:
                carrying += 1
        return variant / total, genotyped, carrying

    def match(self, svcall, config, device: int = 0) -> Optional[int]:
        """The distance (smaller is better) or None when `svcall` is not this variant (snfp.py:91-107)."""
        return match_batch([(self, svcall)], config, device)[0]


def _gate(pv, svcall, config) -> Optional[int]:
    dist = abs(pv.pos - svcall.pos) + abs(abs(pv.svlen) - abs(svcall.svlen))
    minlen = float(min(abs(pv.svlen), abs(svcall.svlen)))
    if dist > config.combine_match * math.sqrt(minlen) or dist > config.combine_match_max:
        return None
    return dist


def match_batch(pairs, config, device: int = 0) -> list:
    """`pv.match(svcall)` for every (population variant, call) pair; the insertions' sequence comparisons of all pairs go to
    the GPU in one launch.  The reference rejects when `(svlen - d) / svlen <= combine_pctseq`, i.e. accepts iff
    d < (1 - pctseq) * svlen: that bound is the band of the alignment (distances beyond it need not be exact)."""
    out = [_gate(pv, sv, config) for pv, sv in pairs]
    limit = config.combine_pctseq
    todo = [k for k, (pv, sv) in enumerate(pairs) if out[k] is not None and pv.svtype == 'INS' and limit]
    if todo:
        seqs = [(pairs[k][0].alt.encode("latin-1"), pairs[k][1].alt.encode("latin-1")) for k in todo]
        # reject iff (svlen - d) / svlen <= limit  <=>  d >= svlen * (1 - limit): any distance >= that bound may come back as -1
        bounds = [max(0, int(math.ceil(abs(pairs[k][0].svlen) * (1.0 - limit))) + 1) for k in todo]
        d = lib.edit_distance_batch(seqs, device=device, max_dist=bounds)
        for k, dk in zip(todo, d.tolist()):
            pv = pairs[k][0]
            if dk < 0 or (pv.svlen - dk) / pv.svlen <= limit:
                out[k] = None
    return out
