// snf_group_call.h - SVGroup.call (sv.py:320-481) and the keep / flush walk of CombineTask.execute (parallel.py:536-572)
// over a columnar candidate table: one thread per group.  A group is a handful of candidates (one or two per sample), so the
// parallel dimension is the 10^4..10^6 groups of a merge; the body is integer work plus a few correctly rounded divisions.
#pragma once
#include "snf_exact.h"
#include "../../include/sniffles_amd.h"

namespace snf {

struct GroupCallView {
  snf_group_call_config_t cfg;
  int64_t n_groups;
  const int64_t* group_off; const int32_t* member;
  const snf_group_cand_t* cand; const int32_t* cand_win; const int32_t* group_win_hi;
  const int32_t* win_bin; const double* win_thr;
  snf_group_out_t* out; uint8_t* chosen; double* pos_mean;
  int32_t* scratch;   // one int per member: sorting space of the medians
};

// util.mean_or_none_round / round(util.mean(..)): CPython's int / int is correctly rounded and so is this quotient of two
// exactly representable doubles; round() of a float is half-to-even = rint
SNF_HD int32_t mean_round(int64_t sum, int64_t n) { return (int32_t)rint((double)sum / (double)n); }

// int(statistics.median(sorted s)): the middle element, or int((a + b) / 2) - truncation towards zero like C's division
SNF_HD int32_t median_int(const int32_t* s, int64_t n) {
  if (n & 1) return s[n / 2];
  const int64_t t = (int64_t)s[n / 2 - 1] + (int64_t)s[n / 2];
  return (int32_t)(t / 2);
}

SNF_HD void group_call_body(int64_t g, const GroupCallView& v) {
  const int64_t lo = v.group_off[g], hi = v.group_off[g + 1], n = hi - lo;
  snf_group_out_t o;
  o.flush_win = -1; o.emit = 0; o.n_pass = 0; o.n_present = 0; o.pos = o.svlen = o.end = 0; o.alt_member = (int32_t)lo;
  o.qual = SNF_NONE_I32; o.support = 0; o.fwd = o.rev = 0; o.precise = 0; o.n = (int32_t)n; o.stdev_pos = o.stdev_len = 0.0;
  for (int z = 0; z < 5; z++) o.cov[z] = SNF_NONE_I32;
  if (n <= 0) { v.out[g] = o; return; }
  const snf_group_cand_t* C = v.cand;
  const int32_t* M = v.member;
  // ---- running mean and the windows the group is alive in (sv.py:297-318, parallel.py:541-556)
  {
    double pm = 0.0;
    int64_t k = 0;
    int32_t w = v.cand_win[M[lo]];
    const int32_t w_hi = v.group_win_hi[g];
    for (;;) {
      while (k < n && v.cand_win[M[lo + k]] == w) {
        const double p = (double)C[M[lo + k]].pos;
        if (k == 0) pm = p;
        else { pm *= (double)k; pm += p; pm /= (double)(k + 1); }
        v.pos_mean[lo + k] = pm;
        k++;
      }
      const bool keep = fabs(pm - (double)v.win_bin[w]) < v.win_thr[w];
      if (!keep) { o.flush_win = w; break; }
      if (w + 1 >= w_hi) { o.flush_win = -1; break; }
      w++;
    }
    if (k < n) { o.emit = -1; v.out[g] = o; for (int64_t q = k; q < n; q++) v.pos_mean[lo + q] = pm; for (int64_t q = 0; q < n; q++) v.chosen[lo + q] = 0; return; }
  }
  // ---- which candidate speaks for its sample (sv.py:388-404), distinct samples, sums
  int64_t s_qual = 0, n_qual = 0, s_support = 0, s_fwd = 0, s_rev = 0, n_precise = 0, s_cov[5] = {0, 0, 0, 0, 0}, n_cov[5] = {0, 0, 0, 0, 0};
  bool any_pass = false;
  for (int64_t k = 0; k < n; k++) {
    const snf_group_cand_t& c = C[M[lo + k]];
    int64_t prev = -1;      // the member of the same sample that holds its genotype so far
    bool seen = false;
    for (int64_t j = 0; j < k; j++)
      if (C[M[lo + j]].sample == c.sample) { seen = true; if (v.chosen[lo + j]) prev = j; }
    if (!seen) { o.n_present++; v.chosen[lo + k] = 1; }
    else {
      const snf_group_cand_t& pc = C[M[lo + prev]];
      const bool take = pc.gt_a < 0 || (c.gt_a >= 0 && (c.gt_a > pc.gt_a || (c.gt_a == pc.gt_a && c.gt_b >= pc.gt_b)));
      v.chosen[lo + k] = take ? 1 : 0;
      if (take) v.chosen[lo + prev] = 0;
    }
    if (c.qc) o.n_pass++;
    if (c.qc && c.pass) any_pass = true;
    if (c.qual != SNF_NONE_I32) { s_qual += c.qual; n_qual++; }
    s_support += c.support; s_fwd += c.fwd; s_rev += c.rev; n_precise += c.precise ? 1 : 0;
    for (int z = 0; z < 5; z++) if (c.cov[z] != SNF_NONE_I32) { s_cov[z] += c.cov[z]; n_cov[z]++; }
  }
  // ---- the call (sv.py:325-340, 419-481)
  const snf_group_call_config_t& f = v.cfg;
  const double ns = (double)f.n_samples;
  const bool single_noqc = f.no_qc && f.n_samples == 1;
  const bool confident = (o.n_pass > 0 && (double)o.n_pass / ns >= f.combine_high_confidence) ||
                         ((double)o.n_present / ns >= f.combine_low_confidence && o.n_present >= f.combine_low_confidence_abs);
  bool emit = (confident || single_noqc) && (f.combine_output_filtered || any_pass || single_noqc);
  int32_t* s = v.scratch + lo;
  for (int64_t k = 0; k < n; k++) s[k] = C[M[lo + k]].pos;
  o.stdev_pos = stdev_i32(s, n);
  sort_inplace(s, n, LessI32());
  const int32_t med_pos = median_int(s, n);
  for (int64_t k = 0; k < n; k++) s[k] = C[M[lo + k]].svlen;
  o.stdev_len = stdev_i32(s, n);
  sort_inplace(s, n, LessI32());
  const int32_t med_len = median_int(s, n);
  const snf_group_cand_t& first = C[M[lo]];
  int32_t end;
  if (first.is_ins) {
    end = med_pos;
    int64_t best = iabs64((int64_t)first.alt_len - med_len);
    for (int64_t k = 0; k < n; k++) {
      const int64_t d = iabs64((int64_t)C[M[lo + k]].alt_len - med_len);
      if (d < best) { best = d; o.alt_member = (int32_t)(lo + k); }
    }
  } else {
    end = med_pos + (int32_t)iabs64(med_len);
  }
  o.pos = f.dev_combine_medians ? med_pos : first.pos;
  o.svlen = f.dev_combine_medians ? med_len : first.svlen;
  o.end = f.dev_combine_medians ? end : first.end;
  if (n_qual) o.qual = mean_round(s_qual, n_qual);
  o.support = mean_round(s_support, n);
  o.fwd = (int32_t)s_fwd; o.rev = (int32_t)s_rev;
  o.precise = (double)n_precise / (double)n > 0.5 ? 1 : 0;
  for (int z = 0; z < 5; z++) if (n_cov[z]) o.cov[z] = mean_round(s_cov[z], n_cov[z]);
  if (iabs64(o.svlen) < f.minsvlen_screen) emit = false;
  o.emit = emit ? 1 : 0;
  v.out[g] = o;
}

}  // namespace snf
