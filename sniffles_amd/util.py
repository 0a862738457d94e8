"""Small numeric helpers with the reference's semantics (reference `src/sniffles/util.py:25-80`); used by the
host-side multi-sample bookkeeping (`SVGroup.call`)."""
from __future__ import annotations

import statistics


def stdev(nums):
    nums = list(nums)
    return statistics.stdev(nums) if len(nums) > 1 else 0


def median(nums):
    return int(statistics.median(nums))


def mean(nums):
    nums = list(nums)
    return sum(nums) / len(nums)


def mean_or_none_round(nums):
    nums = list(nums)
    return round(sum(nums) / len(nums)) if nums else None


def load_tandem_repeats(filename: str, padding: int) -> dict:
    """Tandem-repeat annotation (BED: contig, start, end, ...) -> {contig: [(start - padding, end + padding), ...]},
    every contig's list ascending (reference `util.load_tandem_repeats`, util.py:121-144: lines with fewer than three
    columns are skipped, an unsorted file is sorted after loading).  The task input wants the list per contig
    (`Task.tandem_repeats`, `pipeline.call_sample(tandem_repeats=...)`)."""
    per_contig, unsorted = {}, False
    with open(filename, "r") as handle:
        for line in handle:
            cols = line.split("\t")
            if len(cols) < 3:
                continue
            start, end = int(cols[1]), int(cols[2])
            rows = per_contig.setdefault(cols[0], [])
            if rows and start < rows[-1][0]:        # against the PADDED start of the previous entry, like the reference
                unsorted = True
            rows.append((max(0, start - padding), end + padding))
    if unsorted:
        for rows in per_contig.values():
            rows.sort()
    return per_contig
