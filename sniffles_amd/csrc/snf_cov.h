// snf_cov.h - the reference's dense uint16 coverage vector (leadprov.py:451, 510) as queries on the sparse read table, in
// its two rarely needed refinements:
//   * LeadProvider._mask_N_coverage (leadprov.py:420-443): coverage reads as 0 wherever the reference base is 'N'
//     (intervals nm_start / nm_end per task, sorted and disjoint)
//   * the vector is uint16: a position covered by 65536 + k reads reads as k (numpy wraps `coverage[s:e] += 1`), which
//     also enters coverage.mean() and the per-bin means of SNFile.annotate_block_coverages (snf.py:249-267)
// Point queries already wrap (mod 2^16); cov_range_sum gives the exact sum of the masked, wrapped vector over a range by
// walking the read starts, read ends and mask boundaries inside it - used for the means of tasks that have a mask or are
// deep enough to wrap (Reads::exact), everything else keeps the closed forms.
#pragma once
#include "snf_exact.h"

namespace snf {

struct Reads {   // one task's read table + mask
  const int32_t *r_start, *re_sorted, *rs_top, *re_top;
  int64_t lo, hi;      // the task's reads
  int64_t L;           // contig length
  const int32_t *nm_start, *nm_end;
  int64_t nm_lo, nm_hi;   // the task's mask intervals (nm_lo == nm_hi: none)
};

// is x inside a mask interval?
SNF_HD bool cov_masked(const Reads& q, int64_t x) {
  if (q.nm_lo >= q.nm_hi) return false;
  int64_t lo = q.nm_lo, hi = q.nm_hi;      // first interval whose end is > x
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if ((int64_t)q.nm_end[mid] <= x) lo = mid + 1; else hi = mid; }
  return lo < q.nm_hi && (int64_t)q.nm_start[lo] <= x;
}

// sum over x in [a, b) (0 <= a <= b <= L) of the masked uint16 coverage
SNF_HD uint64_t cov_range_sum(const Reads& q, int64_t a, int64_t b) {
  if (a >= b) return 0;
  int64_t ns = bound_top_i32<true>(q.r_start, q.rs_top, q.lo, q.hi, a);      // reads with start <= a
  int64_t ne = bound_top_i32<true>(q.re_sorted, q.re_top, q.lo, q.hi, a);    // reads with end <= a
  int64_t k = q.nm_lo;                                                        // first mask interval whose end is > a
  { int64_t lo = q.nm_lo, hi = q.nm_hi; while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if ((int64_t)q.nm_end[mid] <= a) lo = mid + 1; else hi = mid; } k = lo; }
  uint64_t sum = 0;
  int64_t x = a;
  while (x < b) {
    // the depth is constant up to the next read start / read end / mask boundary behind x
    int64_t nx = b;
    if (ns < q.hi && (int64_t)q.r_start[ns] < nx) nx = q.r_start[ns];
    if (ne < q.hi && (int64_t)q.re_sorted[ne] < nx) nx = q.re_sorted[ne];
    bool masked = false;
    if (k < q.nm_hi) {
      if ((int64_t)q.nm_start[k] <= x) { masked = true; if ((int64_t)q.nm_end[k] < nx) nx = q.nm_end[k]; }
      else if ((int64_t)q.nm_start[k] < nx) nx = q.nm_start[k];
    }
    if (!masked) sum += (uint64_t)(nx - x) * (uint64_t)((uint64_t)((ns - q.lo) - (ne - q.lo)) & 0xffffu);
    x = nx;
    while (ns < q.hi && (int64_t)q.r_start[ns] <= x) ns++;
    while (ne < q.hi && (int64_t)q.re_sorted[ne] <= x) ne++;
    while (k < q.nm_hi && (int64_t)q.nm_end[k] <= x) k++;
  }
  return sum;
}

}  // namespace snf
