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
