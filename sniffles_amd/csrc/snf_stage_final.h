// snf_stage_final.h - kernel bodies for Task.finalize_candidates (parallel.py:129-201).
//
// Reference semantics: qc_sv / qc_sv_support / qc_sv_post_annotate (postprocessing.py:133-600),
// phase_sv / genotype_sv (postprocessing.py:607-654), Genotyper.calculate (genotyping.py:124-241),
// rescue_phasing (parallel.py:203-249), annotate_sv INS branch (postprocessing.py:33-66),
// consensus.novel_from_reads (consensus.py:280-394).
#pragma once
#include "snf_stage_call.h"

namespace snf {

#define SNF_PS_NULL_CODE 0x7fffffff

struct LeadIter {  // iterates cluster.leads of a call (for BND: the leads resolve_bnd selected)
  const View& v; const CallX& x; int32_t k;
  SNF_HD LeadIter(const View& v_, const CallX& x_) : v(v_), x(x_), k(-1) {}
  SNF_HD bool next(int32_t* slot, uint32_t* orig) {
    for (k++; k < x.fn; k++) {
      int32_t s = v.FI[x.flo + k];
      if (v.F_sel[s]) { *slot = s; *orig = (uint32_t)v.F_orig[s]; return true; }
    }
    return false;
  }
};

// per-call aggregates over cluster.leads that the scalar QC / phasing logic needs; the thread path collects them
// serially (collect_agg), the wave path (snf_wave_call.h) with ballots and reductions
struct LeadAgg {
  int nstrands;          // len(set(lead.strand))
  int64_t close_edge;    // leads with qry_start <= d or |read_len - qry_start| <= d (postprocessing.py:574-577)
  int hp_val; int64_t hp_support, hp_other;          // phase_sv majorities (postprocessing.py:626-654)
  int32_t ps_val; int64_t ps_support, ps_other;      // ps_val == SNF_PS_NULL_CODE: "NULL"
  const double* nm_row;  // NM ratio of lead k of the call, gathered by the wave (LDS), or nullptr: read from the input column
  double nm_mean; int has_nm_mean;   // np.nanmean of the leads' NM ratios, when the wave has already formed it (e1w_finalize)
};

SNF_HD double py_round(double x) { return rint(x); }  // round-half-even (default rounding mode)

SNF_HD int64_t rescale_support(const snf_call_t& c, const snf_config_t& cfg) {
  if (c.svtype != SNF_INS || c.svlen < cfg.long_ins_length) return c.support;
  double scale = cfg.long_ins_rescale_mult * ((double)c.svlen / (double)cfg.long_ins_length);
  return (int64_t)py_round((double)c.support * (cfg.long_ins_rescale_base + scale));
}

SNF_HD bool qc_support_auto(const snf_call_t& c, double cov_global, const snf_config_t& cfg) {
  int64_t support = rescale_support(c, cfg);
  int64_t lst[3]; int k = 0;
  if (c.cov[0] != 0) lst[k++] = c.cov[0];
  if (c.cov[4] != 0) lst[k++] = c.cov[4];
  if (k == 0) for (int j = 1; j <= 3; j++) if (c.cov[j] != 0) lst[k++] = c.cov[j];
  double regional;
  if (k == 0) regional = cov_global;
  else {
    int64_t s = 0; for (int j = 0; j < k; j++) s += lst[j];
    regional = py_round((double)s / (double)k);
    if (regional == 0) regional = cov_global;
  }
  double gw = 1.0 - cfg.minsupport_auto_regional_coverage_weight;
  double cov = regional * cfg.minsupport_auto_regional_coverage_weight + cov_global * gw;
  double min_support = py_round(cfg.minsupport_auto_base + cfg.minsupport_auto_mult * cov);
  return (double)support >= min_support;
}

SNF_HD bool qc_sv_support(snf_call_t& c, double cov_global, const snf_config_t& cfg) {
  bool ok = cfg.minsupport < 0 ? qc_support_auto(c, cov_global, cfg) : (c.support >= cfg.minsupport);
  if (!ok) { c.filter = SNF_F_SUPPORT_MIN; return false; }
  return true;
}

#define SNF_FAIL(f) do { c.filter = (f); return false; } while (0)

SNF_HD bool qc_sv(const View& v, snf_call_t& c, const LeadAgg& g) {
  const snf_config_t& cfg = v.cfg;
  int t = c.svtype;
  bool single = t == SNF_SINGLE_LEFT || t == SNF_SINGLE_RIGHT;
  double alen = (double)iabs64(c.svlen);
  if (cfg.qc_stdev) {
    if (c.stdev_pos > (double)cfg.qc_stdev_abs_max) SNF_FAIL(SNF_F_STDEV_POS);
    if (t != SNF_BND && !single && c.stdev_pos / alen > 2.0) SNF_FAIL(SNF_F_STDEV_POS);
    if (!(c.stdev_len != c.stdev_len) && c.stdev_len != 0) {
      if (t != SNF_BND && c.stdev_len / alen > 1.0) SNF_FAIL(SNF_F_STDEV_LEN);
      if (c.stdev_len > (double)cfg.qc_stdev_abs_max) SNF_FAIL(SNF_F_STDEV_LEN);
    }
  }
  if (single && !cfg.dev_output_candidates) SNF_FAIL(SNF_F_SINGLE_BREAK);
  if (iabs64(c.svlen) < cfg.minsvlen && t != SNF_BND) {
    if (c.support < 10 || cfg.minsvlen_hard_cap) SNF_FAIL(SNF_F_SVLEN_MIN);
  }
  if (t == SNF_BND) {
    if (cfg.qc_bnd_filter_strand && g.nstrands < 2) SNF_FAIL(SNF_F_STRAND_BND);
  }
  double up = c.cov[0], ce = c.cov[2], dn = c.cov[4];
  if (t == SNF_DEL && cfg.long_del_length != -1 && iabs64(c.svlen) >= cfg.long_del_length && !cfg.mosaic &&
      iabs64(c.svlen) <= cfg.dev_longer_del) {
    double scaled = cfg.long_del_coverage / 2.0;
    if (ce > (up + dn) * scaled) {
      if (up > ce && ce > dn) { if (dn / up < 0.7) SNF_FAIL(SNF_F_COV_CHANGE_DEL); }
      else if (up < ce && ce < dn) { if (up / dn < 0.7) SNF_FAIL(SNF_F_COV_CHANGE_DEL); }
    }
    if (up > dn) { if (0.5 > dn / up || ce > dn) SNF_FAIL(SNF_F_COV_CHANGE_DEL); }
    else if (up < dn) { if (0.5 > up / dn || up < ce) SNF_FAIL(SNF_F_COV_CHANGE_DEL); }
  } else if (t == SNF_DUP && cfg.long_dup_length != -1 && iabs64(c.svlen) >= cfg.long_dup_length && !cfg.mosaic &&
             iabs64(c.svlen) <= cfg.dev_longer_dup) {
    double scaled = cfg.long_dup_coverage / 2.0;
    if (ce < (up + dn) * scaled) {
      if (up > ce && ce > dn) { if (dn / up < 0.7) SNF_FAIL(SNF_F_COV_CHANGE_DUP); }
      else if (up < ce && ce < dn) { if (up / dn < 0.7) SNF_FAIL(SNF_F_COV_CHANGE_DUP); }
      if (up > dn) { if (0.5 > dn / up || ce < dn) SNF_FAIL(SNF_F_COV_CHANGE_DUP); }
      else if (up < dn) { if (0.5 > up / dn || up > ce) SNF_FAIL(SNF_F_COV_CHANGE_DUP); }
    }
  } else if (t == SNF_INS && (c.cov[0] < cfg.qc_coverage || c.cov[4] < cfg.qc_coverage)) {
    SNF_FAIL(SNF_F_COV_CHANGE_INS);
  }
  if (t == SNF_INS || t == SNF_DEL) {
    bool no_split_sa = c.support_sa <= 0;  // None or 0
    if (c.sa_frac > cfg.dev_inline_sa_support_max && c.sa_count > 5 && no_split_sa) SNF_FAIL(SNF_F_INLINE_SA);
  }
  // qc_coverage_samples(): the sampler is never fed -> (True, None) (sv.py:219-223)
  double f = cfg.qc_coverage_max_change_frac;
  if (f != -1.0) {
    double u = c.cov[0] ? c.cov[0] : 1.0, s = c.cov[1] ? c.cov[1] : 1.0, m = c.cov[2] ? c.cov[2] : 1.0,
           e = c.cov[3] ? c.cov[3] : 1.0, d = c.cov[4] ? c.cov[4] : 1.0;
    if (fabs(u - s) / fmax(u, s) > f) SNF_FAIL(SNF_F_COV_CHANGE_FRAC_US);
    if (fabs(s - m) / fmax(s, m) > f) SNF_FAIL(SNF_F_COV_CHANGE_FRAC_SC);
    if (fabs(m - e) / fmax(m, e) > f) SNF_FAIL(SNF_F_COV_CHANGE_FRAC_CE);
    if (fabs(e - d) / fmax(e, d) > f) SNF_FAIL(SNF_F_COV_CHANGE_FRAC_ED);
  }
  return true;
}

// phase_sv: majority HP / PS over distinct read_id (last lead of a read wins); fills the phase part of LeadAgg
#define SNF_BIG_FINAL_CAP 512   /* leads of a call whose rows x_big<2> keeps in LDS */
#if defined(__HIP_DEVICE_COMPILE__)
// collect_agg for a call with more than 64 leads, executed by the whole wave (x_big<2>, "uniform" mode: every lane runs the
// finalize body in lock step).  The serial form walks the leads one by one through six dependent gathers and looks for a later
// lead of the same read with a second loop: with a few hundred leads that is 10^4-10^5 dependent loads per call.  Here a lane
// takes every 64th lead (one 64-byte record each), read ids / flags / the phase-set list sit in LDS rows (x_big<2>; leads beyond
// the rows' capacity use the global scratch rows), "is there a later lead of this read" is a broadcast scan of the LDS row,
// and the survivors are compacted with ballots.  All 64 lanes leave with the same aggregates.
SNF_D void collect_agg_wave(const View& v, const CallX& x, int task, LeadAgg* g) {
  const int lane = (int)(threadIdx.x & 63);
  const int32_t n = x.fn;
  const bool in_lds = v.stage_w != nullptr && n <= v.stage_cap;
  int32_t* rid = in_lds ? v.stage_w : v.w4 + x.flo;                                  // read id of lead k (w1 still holds the read names for d3_rnames)
  int32_t* flg = in_lds ? v.stage_w + v.stage_cap : v.w5 + x.flo;                    // selected | hap << 1
  int32_t* psv_ = in_lds ? v.stage_w + 2 * v.stage_cap : v.w6 + x.flo;               // phase-set code of lead k
  int32_t* a0 = in_lds ? v.stage_w + 3 * v.stage_cap : v.w0 + x.flo;                 // phase sets of the contributing reads
  int32_t* tmp = in_lds ? v.stage_w + 4 * v.stage_cap : v.w7 + x.flo;
  const int32_t ps_null = v.t_ps_null[task];
  int f = 0, r = 0; int64_t close = 0;
  for (int32_t k = lane; k < n; k += 64) {
    const int32_t s = v.FI[x.flo + k];
    const int sel = v.F_sel[s] ? 1 : 0;
    const LeadRec rec = v.Lrec[v.F_lpos[s]];
    rid[k] = (int32_t)rec.read_id;
    flg[k] = sel | ((int)rec.hap << 1);
    const int32_t p = rec.ps;
    psv_[k] = (p == SNF_PS_NONE || p == ps_null) ? SNF_PS_NULL_CODE : p;
    if (sel) {
      if (rec.strand == 0) f = 1; else r = 1;
      const int64_t qs = rec.qry_start;
      if (qs <= v.cfg.dev_min_close_edge_dist || iabs64((int64_t)rec.read_len - qs) <= v.cfg.dev_min_close_edge_dist) close++;
    }
  }
  g->nstrands = (__ballot(f != 0) ? 1 : 0) + (__ballot(r != 0) ? 1 : 0);
  for (int d = 32; d >= 1; d >>= 1) close += __shfl_xor(close, d, 64);
  g->close_edge = close;
  g->hp_val = 0; g->hp_support = -1; g->hp_other = 0; g->ps_val = 0; g->ps_support = -1; g->ps_other = 0;
  __syncthreads();
  int64_t hc[3] = {0, 0, 0};
  int32_t np_ = 0;
  if (v.cfg.phase) {
    // reads_phases = {read_id: (hap, ps)}: the last selected lead of a read wins
    for (int32_t base = 0; base < n; base += 64) {
      const int32_t k = base + lane;
      const int32_t my = k < n ? rid[k] : 0, mf = k < n ? flg[k] : 0;
      bool later = false;
      for (int32_t k2 = base + 1; k2 < n; k2++) later |= (k2 > k) && (flg[k2] & 1) && rid[k2] == my;   // same address in all lanes
      const bool contrib = k < n && (mf & 1) && !later;
      for (int hh = 0; hh < 3; hh++) hc[hh] += __builtin_popcountll(__ballot(contrib && (mf >> 1) == hh));
      const unsigned long long m = __ballot(contrib);
      if (contrib) a0[np_ + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = psv_[k];
      np_ += __builtin_popcountll(m);
    }
    __syncthreads();
  } else {
    // without --phase the reference still looks at nothing but strands and edges (hc, a0 stay empty)
  }
  int hpv = 0; int64_t hp_support = -1;
  for (int h = 0; h < 3; h++) if (hc[h] > 0 && hc[h] >= hp_support) { hp_support = hc[h]; hpv = h; }
  int64_t other_hp = 0;
  for (int h = 0; h < 3; h++) if (h != hpv) other_hp += hc[h];
  wave_sort_inplace(a0, (int64_t)np_, LessI32{}, tmp);
  // most_common over the sorted list: one lane per element, a run start counts its run
  int32_t best_len = -1, best_val = 0; int64_t other_ps = 0;
  for (int pass = 0; pass < 2; pass++) {
    int32_t my_len = 0, my_val = 0; int64_t oth = 0;
    for (int32_t i = lane; i < np_; i += 64) {
      const int32_t val = a0[i];
      if (i > 0 && a0[i - 1] == val) continue;
      int32_t j = i + 1; while (j < np_ && a0[j] == val) j++;
      const int32_t len = j - i;
      if (pass == 0) { if (len > my_len || (len == my_len && val >= my_val) || my_len == 0) { my_len = len; my_val = val; } }
      else if (val != best_val && val != SNF_PS_NULL_CODE) oth += len;
    }
    if (pass == 0) {
      // (count, value) descending: the longest run, ties -> the larger value
      for (int d = 32; d >= 1; d >>= 1) {
        const int32_t ol = __shfl_xor(my_len, d, 64), ov = __shfl_xor(my_val, d, 64);
        if (ol > my_len || (ol == my_len && ol > 0 && ov > my_val)) { my_len = ol; my_val = ov; }
      }
      if (my_len > 0) { best_len = my_len; best_val = my_val; }
    } else {
      for (int d = 32; d >= 1; d >>= 1) oth += __shfl_xor(oth, d, 64);
      other_ps = oth;
    }
  }
  g->hp_val = hpv; g->hp_support = hp_support; g->hp_other = other_hp;
  g->ps_val = best_len > 0 ? best_val : 0; g->ps_support = best_len > 0 ? best_len : -1; g->ps_other = other_ps;
  __syncthreads();
  // rescue_phasing sums the leads' NM ratios in list order (np.nanmean): gathered by the wave into the rows that are free now
  g->nm_row = nullptr; g->has_nm_mean = 0; g->nm_mean = 0.0;
  if (in_lds && 2 * n <= 2 * v.stage_cap && v.cfg.phase && v.cfg.mode_call_sample) {
    double* row = (double*)v.stage_w;     // rid + flg rows = stage_cap doubles
    for (int32_t k = lane; k < n; k += 64) row[k] = v.in_nm[(uint32_t)v.F_orig[v.FI[x.flo + k]]];
    __syncthreads();
    g->nm_row = row;
  }
}
#endif

SNF_HD void collect_agg(const View& v, const CallX& x, int task, LeadAgg* g) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (v.wave_uniform) { collect_agg_wave(v, x, task, g); return; }
#endif
  g->nm_row = nullptr; g->has_nm_mean = 0; g->nm_mean = 0.0;
  int64_t hc[3] = {0, 0, 0};
  int32_t* a0 = v.w0 + x.flo;
  int32_t np_ = 0;
  int32_t ps_null = v.t_ps_null[task];
  int f = 0, r = 0; int64_t close = 0;
  for (int32_t k = 0; k < x.fn; k++) {
    int32_t s = v.FI[x.flo + k];
    if (!v.F_sel[s]) continue;
    uint32_t o = (uint32_t)v.F_orig[s];
    if (v.in_strand[o] == 0) f = 1; else r = 1;
    int64_t qs = v.in_qry_start[o];
    if (qs <= v.cfg.dev_min_close_edge_dist || iabs64((int64_t)v.in_read_len[o] - qs) <= v.cfg.dev_min_close_edge_dist) close++;
    if (!v.cfg.phase) continue;
    uint32_t rid = v.in_read_id[o];
    bool later = false;
    for (int32_t k2 = k + 1; k2 < x.fn; k2++) {
      int32_t s2 = v.FI[x.flo + k2];
      if (v.F_sel[s2] && v.in_read_id[(uint32_t)v.F_orig[s2]] == rid) { later = true; break; }
    }
    if (later) continue;
    hc[v.in_hap[o]]++;
    int32_t p = v.in_ps[o];
    a0[np_++] = (p == SNF_PS_NONE || p == ps_null) ? SNF_PS_NULL_CODE : p;
  }
  g->nstrands = f + r; g->close_edge = close;
  // most_common: sorted((count, value), reverse=True)
  int hpv = 0; int64_t hp_support = -1;
  for (int h = 0; h < 3; h++) if (hc[h] > 0 && hc[h] >= hp_support) { hp_support = hc[h]; hpv = h; }
  int64_t other_hp = 0;
  for (int h = 0; h < 3; h++) if (h != hpv) other_hp += hc[h];
  SNF_SORT(v.wave_uniform != 0, a0, (int64_t)np_, LessI32{}, v.w7 + x.flo);
  int32_t psv = 0; int64_t ps_support = -1;
  for (int32_t i = 0; i < np_;) {
    int32_t j = i; while (j < np_ && a0[j] == a0[i]) j++;
    if (j - i >= ps_support) { ps_support = j - i; psv = a0[i]; }
    i = j;
  }
  int64_t other_ps = 0;
  for (int32_t i = 0; i < np_;) {
    int32_t j = i; while (j < np_ && a0[j] == a0[i]) j++;
    if (a0[i] != psv && a0[i] != SNF_PS_NULL_CODE) other_ps += j - i;
    i = j;
  }
  g->hp_val = hpv; g->hp_support = hp_support; g->hp_other = other_hp;
  g->ps_val = psv; g->ps_support = ps_support; g->ps_other = other_ps;
}

SNF_HD void phase_sv(const View& v, snf_call_t& c, const LeadAgg& g, int* hp_ret, int* ps_ret) {
  bool hp_pass = ((double)g.hp_other / (double)(g.hp_support + g.hp_other) < v.cfg.phase_conflict_threshold) && g.hp_support > 0;
  bool ps_pass = ((double)g.ps_other / (double)(g.ps_support + g.ps_other) < v.cfg.phase_conflict_threshold) &&
                 g.ps_val != SNF_PS_NULL_CODE && g.ps_support > 0;
  c.ph_set = 1; c.ph_hp = g.hp_val; c.ph_ps = g.ps_val == SNF_PS_NULL_CODE ? -2 : g.ps_val;
  c.ph_hp_support = (int32_t)g.hp_support; c.ph_ps_support = (int32_t)g.ps_support;
  c.ph_hp_pass = hp_pass; c.ph_ps_pass = ps_pass;
  *hp_ret = ((g.hp_val == 1 || g.hp_val == 2) && hp_pass) ? g.hp_val : -1;
  *ps_ret = ps_pass ? g.ps_val : -1;
}

SNF_HD bool coverage_from_list(const int64_t* lst, int k, int64_t* out) {
  int64_t s = 0; int m = 0;
  for (int j = 0; j < k; j++) if (lst[j] != 0) { s += lst[j]; m++; }
  if (m == 0) return false;  // UnknownGenotypeError
  *out = (int64_t)py_round((double)s / (double)m);
  return true;
}

SNF_HD void genotype_sv(const View& v, snf_call_t& c, int hp_ret, int ps_ret) {
  const snf_config_t& cfg = v.cfg;
  int t = c.svtype;
  int64_t support = t == SNF_INS ? rescale_support(c, cfg) : c.support;
  int64_t coverage = 0, l3[3]; bool ok;
  if (t == SNF_INS) { l3[0] = c.cov[2]; ok = coverage_from_list(l3, 1, &coverage); }
  else if (t == SNF_DEL) {
    int64_t sa = c.support_sa > 0 ? c.support_sa : 0;
    l3[0] = c.cov[1] + sa; l3[1] = c.cov[2] + sa; l3[2] = c.cov[3] + sa; ok = coverage_from_list(l3, 3, &coverage);
  } else if (t == SNF_DUP) {
    l3[0] = c.cov[1]; l3[1] = c.cov[3]; ok = coverage_from_list(l3, 2, &coverage);
    if (ok) coverage += (int64_t)py_round((double)support * 0.75);
  } else if (t == SNF_INV) {
    l3[0] = c.cov[0]; l3[1] = c.cov[4]; ok = coverage_from_list(l3, 2, &coverage);
    if (ok) coverage += (int64_t)py_round((double)support * 0.5);
  } else { l3[0] = c.cov[1]; l3[1] = c.cov[2]; l3[2] = c.cov[3]; ok = coverage_from_list(l3, 3, &coverage); }
  if (!ok) { c.filter = SNF_F_GT_FAILED; c.qc = 0; return; }
  if (support > coverage) coverage = support;
  double af = (double)support / (double)coverage;
  int64_t max_lead = support > coverage ? support : coverage;
  int64_t ns = support, ncv = coverage;
  if (max_lead > 250) {
    double norm = 250.0 / (double)max_lead;
    ns = (int64_t)py_round((double)support * norm);
    ncv = (int64_t)py_round((double)coverage * norm);
  }
  // likelihoods p**k * (1-p)**(n-k), their ordering, GQ and z only depend on (ns, ncv) <= 250:
  // table built on the host with CPython's libm (snf_lib: build_gt_lut)
  GtEntry e = v.gt_lut[ns * SNF_GT_N + ncv];
  bool update_this_dup = t == SNF_DUP && af >= cfg.dev_min_dup_vaf;
  bool flt = e.z < cfg.genotype_min_z_score && !cfg.mosaic;
  if (t == SNF_INS && flt && c.svlen >= cfg.long_ins_length && cfg.detect_large_ins) flt = false;
  if (c.filter == SNF_F_PASS && flt) { c.filter = update_this_dup ? SNF_F_PASS : SNF_F_GT; c.qc = !cfg.pass_only; }
  int a = e.order0 == 2 ? 1 : 0, b = e.order0 >= 1 ? 1 : 0;
  if (update_this_dup && e.order0 == 0) { a = 0; b = 1; }
  c.gt_set = 1; c.gt_a = a; c.gt_b = b; c.gt_gq = e.gq;
  c.gt_dr = (int32_t)(coverage - support); c.gt_dv = (int32_t)support;
  c.gt_hp = hp_ret; c.gt_ps = ps_ret;
  c.vaf = af;
  if (a == 1 && b == 1 && c.ph_set && c.ph_hp != 0) { c.ph_hp_pass = 1; c.gt_hp = c.ph_hp; c.gt_ps = c.ph_ps; }
}

SNF_HD bool qc_sv_post_annotate(const View& v, snf_call_t& c, const LeadAgg& g, int task) {
  const snf_config_t& cfg = v.cfg;
  int t = c.svtype;
  double af = (c.vaf != c.vaf) ? 0.0 : c.vaf;
  bool sv_is_mosaic = af <= cfg.mosaic_af_max;
  double cov_avg = v.t_cov_avg[task];
  if ((c.cov[2] < cfg.qc_coverage && (!c.gt_set || (c.gt_a + c.gt_b < 2))) && (t != SNF_DEL && iabs64(c.svlen) > cfg.long_del_length))
    SNF_FAIL(SNF_F_COV_MIN_GT);
  if (cfg.mosaic && !sv_is_mosaic) { if (!qc_sv_support(c, cov_avg, cfg)) return false; }
  int qc_nm = cfg.qc_nm;
  double thr = v.t_qc_nm_thr[task] * cfg.qc_nm_mult;
  if (cfg.mosaic && sv_is_mosaic) qc_nm = cfg.mosaic_qc_nm;
  if (qc_nm && c.nm > thr && (!c.gt_set || c.gt_b == 0)) SNF_FAIL(SNF_F_ALN_NM);
  if (!cfg.mosaic && sv_is_mosaic) {
    bool skip_this_dup = t == SNF_DUP && af >= cfg.dev_min_dup_vaf;
    if (!skip_this_dup) SNF_FAIL(SNF_F_MOSAIC_VAF);
  }
  if (cfg.mosaic && sv_is_mosaic) {
    int min_mosaic_support = cfg.mosaic_min_reads;
    bool accepted = t == SNF_INS || t == SNF_DEL || t == SNF_DUP || t == SNF_INV || t == SNF_BND;
    if (!(c.stdev_len != c.stdev_len) && accepted) {
      bool filter_low_supp = (!c.precise || c.stdev_len / (double)iabs64(c.svlen) > 0.1 || c.stdev_pos > 5) &&
                             1 <= cfg.max_svlen_mosaic;
      min_mosaic_support = (t == SNF_BND || t == SNF_INV || filter_low_supp) ? cfg.mosaic_min_reads : cfg.mosaic_min_reads - 1;
    }
    if (c.support < min_mosaic_support) SNF_FAIL(SNF_F_SUPPORT_MIN);
    if (t != SNF_BND && iabs64(c.svlen) > cfg.max_svlen_mosaic) SNF_FAIL(SNF_F_SVLEN_MAX_MOSAIC);
  }
  if (t != SNF_BND) {
    bool is_long_ins = t == SNF_INS && c.svlen >= cfg.long_ins_length;
    if (!(cfg.mosaic && sv_is_mosaic) && cfg.qc_strand) {
      if (!is_long_ins && g.nstrands < 2) SNF_FAIL(SNF_F_STRAND);
    } else if ((cfg.mosaic && sv_is_mosaic) && cfg.mosaic_qc_strand) {
      if (!is_long_ins && g.nstrands < 2 && c.support >= cfg.mosaic_use_strand_thresholds) SNF_FAIL(SNF_F_STRAND_MOSAIC);
    }
  }
  if (cfg.mosaic && sv_is_mosaic) {
    if ((t == SNF_INV || t == SNF_DUP) && c.svlen < cfg.mosaic_qc_invdup_min_length) SNF_FAIL(SNF_F_SVLEN_MIN_MOSAIC);
  }
  if (c.cov[2] < cfg.qc_coverage && t != SNF_DEL && t != SNF_INS) {
    int64_t lhs = t == SNF_INV ? c.svlen : 0;  // (svtype == "INV" and svlen) > long_inv_length
    if (lhs > cfg.long_inv_length && !(cfg.mosaic && sv_is_mosaic)) { /* pass */ }
    else SNF_FAIL(SNF_F_COV_MIN);
  }
  if (cfg.mosaic) {
    if (sv_is_mosaic && (af < cfg.mosaic_af_min || af > cfg.mosaic_af_max)) SNF_FAIL(SNF_F_MOSAIC_VAF);
    else if (!sv_is_mosaic && !cfg.mosaic_include_germline) SNF_FAIL(SNF_F_NOT_MOSAIC_VAF);
    if (sv_is_mosaic && t != SNF_BND && t != SNF_SINGLE_LEFT && t != SNF_SINGLE_RIGHT) {
      int64_t close = g.close_edge;
      if ((double)close / (double)c.support >= cfg.dev_min_read_close_edge_prop) SNF_FAIL(SNF_F_MOSAIC_SV_CLOSE_EDGE);
    }
  }
  return true;
}

// REF haplotype count of the 100-bp bin `b` (leadprov.py:387-398) as rank queries over the sorted reads
SNF_HD int32_t hapref_count(const View& v, int task, int h, int64_t b) {
  int64_t lo = v.t_read_off[task], hi = v.t_read_off[task + 1];
  int64_t lim = (b + 1) * (int64_t)v.cfg.cluster_binsize;  // start_bin <= b  <=>  start < (b+1)*binsize
  int64_t is = bound_top_i32<false>(v.r_start, v.rs_top, lo, hi, lim), ie = bound_top_i32<false>(v.re_sorted, v.re_top, lo, hi, lim);
  const uint64_t ds = v.pc_s2[is] - v.pc_s2[lo], de = v.pc_e2[ie] - v.pc_e2[lo];  // packed (hp1, hp2) counts, no carries
  int64_t ns, ne;
  if (h == 1) { ns = (int64_t)(ds >> 32); ne = (int64_t)(de >> 32); }
  else if (h == 2) { ns = (int64_t)(ds & 0xffffffffull); ne = (int64_t)(de & 0xffffffffull); }
  else { ns = (is - lo) - (int64_t)(ds >> 32) - (int64_t)(ds & 0xffffffffull); ne = (ie - lo) - (int64_t)(de >> 32) - (int64_t)(de & 0xffffffffull); }
  int64_t cnt = ns - ne;
  return (int32_t)(cnt > 65535 ? 65535 : cnt);
}

struct NmGet {
  const View& v; const CallX& x; const double* row;
  SNF_HD double raw(int64_t i) const { return row ? row[i] : v.in_nm[(uint32_t)v.F_orig[v.FI[x.flo + i]]]; }
  SNF_HD double operator()(int64_t i) const {
    double nm = raw(i);
    return nm != nm ? 0.0 : nm;
  }
};

template <int DEPTH>
SNF_HD void rescue_phasing(const View& v, snf_call_t& c, const CallX& x, int task, const LeadAgg& g) {
  const snf_config_t& cfg = v.cfg;
  if (!cfg.mode_call_sample) return;
  int64_t n = x.fn;
  double sv_nm;
  if (g.has_nm_mean) sv_nm = g.nm_mean;
  else {
    int64_t cnt = 0;
    const NmGet get{v, x, g.nm_row};
    for (int64_t i = 0; i < n; i++) { double nm = get.raw(i); if (nm == nm) cnt++; }
    sv_nm = np_pairwise_sum<DEPTH>(get, n) / (double)cnt;  // np.nanmean
  }
  if (sv_nm > cfg.genotype_error || n <= 3) return;
  if (!c.ph_set || !c.ph_hp_pass) return;
  int hp = c.ph_hp;
  if (hp != 1 && hp != 2) return;
  int32_t h = v.cl_head[x.cluster];
  int32_t b = v.seed_bin[h];
  int32_t sv_reads = v.bin_hap[3 * b + hp];
  int32_t all_reads = hapref_count(v, task, hp, (int64_t)v.seed_start[h] / v.cfg.cluster_binsize);
  if (all_reads == 0) return;
  if ((double)sv_reads / (double)all_reads >= 0.75) {
    if (c.filter == SNF_F_MOSAIC_VAF) { c.filter = SNF_F_PASS; c.gt_b = 1; c.qc = 1; }
  }
}

// write back only the fields finalize_call owns (the consensus chain updates alt_len/alt_off concurrently)
SNF_HD void store_final_fields(snf_call_t& dst, const snf_call_t& c) {
  dst.qc = c.qc; dst.filter = c.filter;
  dst.gt_set = c.gt_set; dst.gt_a = c.gt_a; dst.gt_b = c.gt_b; dst.gt_gq = c.gt_gq; dst.gt_dr = c.gt_dr; dst.gt_dv = c.gt_dv;
  dst.gt_hp = c.gt_hp; dst.gt_ps = c.gt_ps; dst.vaf = c.vaf;
  dst.ph_set = c.ph_set; dst.ph_hp = c.ph_hp; dst.ph_ps = c.ph_ps; dst.ph_hp_support = c.ph_hp_support;
  dst.ph_ps_support = c.ph_ps_support; dst.ph_hp_pass = c.ph_hp_pass; dst.ph_ps_pass = c.ph_ps_pass;
}

// scalar tail of finalize_candidates for one call, given the lead aggregates
// DEPTH: recursion frames of the pairwise sum (1 when the call has <= 128 leads, 26 covers 2^31)
template <int DEPTH>
SNF_HD void finalize_call(const View& v, snf_call_t& c, const CallX& x, const LeadAgg& g, int task) {
  const snf_config_t& cfg = v.cfg;
  c.qc = c.qc && qc_sv(v, c, g);
  if (!cfg.mosaic && c.qc) c.qc = c.qc && qc_sv_support(c, v.t_cov_avg[task], cfg);
  int hp_ret = -1, ps_ret = -1;
  if (cfg.phase) phase_sv(v, c, g, &hp_ret, &ps_ret);
  genotype_sv(v, c, hp_ret, ps_ret);
  c.qc = c.qc && qc_sv_post_annotate(v, c, g, task);
  bool phasing_rescue = c.svtype != SNF_BND && iabs64(c.svlen) <= cfg.dev_maxsvlen_extra &&
                        c.support >= (int)((double)cfg.dev_minreads_extra * 0.60);
  if (cfg.phase && !c.qc && phasing_rescue) rescue_phasing<DEPTH>(v, c, x, task, g);
}

SNF_HD void e1_finalize_body(int64_t i, const View& v) {
  if (i >= v.cnt->n_calls) return;
  snf_call_t& cref = v.calls[i];
  int task = cref.task_index;
  if (v.t_status[task] != SNF_TASK_OK) return;
  const CallX x = v.callx[i];
  if (v.wave_path && (x.fn <= 64 || v.big_wave)) return;  // e1w_finalize / x_big<2> (snf_wave_call.h)
  snf_call_t c = cref;
  LeadAgg g;
  collect_agg(v, x, task, &g);
  finalize_call<26>(v, c, x, g, task);
  store_final_fields(cref, c);
}

// ------------------------------------------------------------------------------------------ consensus
// E2: best lead + sizes per INS call (postprocessing.py:33-66)
SNF_HD int64_t cons_npos(int64_t L, int klen, int skip) { int64_t m = L - klen; return m <= 0 ? 0 : (m + skip - 1) / skip; }
SNF_HD int cons_skip(const snf_config_t& cfg, int64_t L) {
  return cfg.consensus_kmer_skip_base + (int)((double)L * cfg.consensus_kmer_skip_seqlen_mult);
}

// size class of a consensus call for the gfx950 workgroup kernels (snf_wave_cons.h): 1 SMALL (256-slot anchor table, 128
// sampled positions, 64 others, vote counters for 384 columns in LDS), 2 LARGE (1024 / 512 / 254, 8192 columns),
// 4 ROWS (1024 / 512 / 512, aligned rows in HBM), 0: fits none of them -> thread kernels e4/e5/e6
#define SNF_CONS_SMALL_L 384
#define SNF_CONS_LARGE_L 8192
SNF_HD int cons_class_of(int wave_path, int klen, int skip, int64_t L, int32_t n_others) {
  if (!wave_path || klen > 7 || klen < 1 || skip < 1 || L >= 65000) return 0;
  int64_t npos = cons_npos(L, klen, skip);
  if (npos <= 120 && n_others <= 64 && L <= SNF_CONS_SMALL_L && skip <= 7) return 1;   // (skip <= 7: a step's bytes fit the k-mer word, snf_wave_cons.h)
  if (npos <= 500 && n_others <= 254 && L <= SNF_CONS_LARGE_L) return 2;
  if (npos <= 500 && n_others <= 512) return 4;
  return 0;
}
SNF_HD int cons_class(const View& v, int64_t L, int32_t n_others) {
  if (v.cons_thread_only) return 0;
  return cons_class_of(v.wave_path, v.cfg.consensus_kmer_len, cons_skip(v.cfg, L), L, n_others);
}
// sampling steps of consensus call `cid` (consensus.py:280: `skip` for the other reads, `skip_repetitive` for the best read's anchors)
SNF_HD int cons_skip_reads(const View& v, int64_t cid, int64_t L) { return v.cons_skip_arr ? v.cons_skip_arr[cid] : cons_skip(v.cfg, L); }
SNF_HD int cons_skip_anchors(const View& v, int64_t cid, int64_t L) { return v.cons_skiprep_arr ? v.cons_skiprep_arr[cid] : cons_skip(v.cfg, L); }
SNF_HD bool cons_wave_eligible(const View& v, int64_t L, int32_t n_others) { return cons_class(v, L, n_others) != 0; }

// where this pass's ALT bytes live (pinned host memory when the total fits the batch's pinned buffer, else the HBM pool)
SNF_HD uint8_t* alt_base(const View& v) { return v.cnt->alt_in_pinned ? v.alt_pin : v.alt_pool; }
SNF_HD void alt_decide(const View& v, int64_t alt_total) {
  v.cnt->alt_in_pinned = (!(v.out_mode & SNF_OUT_DEVICE) && v.alt_pin && alt_total <= v.alt_pin_cap) ? 1 : 0;
}

SNF_HD void e2_best_body(int64_t i, const View& v) {
  const int64_t nc = v.cnt->n_calls;
  if (i == 0) {
    v.fN[nc] = 0; v.fL[nc] = 0; v.sz_tab[nc] = 0; v.sz_aln[nc] = 0; v.sz_rd[nc] = 0;
    for (int k = 0; k < 8; k++) v.cnt->n_cls[k] = 0;   // filled by e3_conslist (finalize may run more than once)
    v.cnt->n_cons_fallback = 0;
    if (nc == 0) alt_decide(v, 0);
  }
  if (i >= nc) return;
  v.fN[i] = 0; v.fL[i] = 0; v.sz_tab[i] = 0; v.sz_aln[i] = 0; v.sz_rd[i] = 0;
  snf_call_t& c = v.calls[i];
  CallX& x = v.callx[i];
  x.cons_id = -1;
  // best lead / number of other sequence-bearing leads: chosen by the call kernels (d2w_call, d2_call)
  if (c.svtype != SNF_INS || v.cfg.symbolic || v.t_status[c.task_index] != SNF_TASK_OK || x.best < 0) { x.best = -1; x.n_others = 0; x.do_cons = 0; return; }
  const int32_t best = x.best;
  const int64_t L = v.F_seq_len[best];
  c.alt_len = (int32_t)L;
  v.fN[i] = (uint32_t)L;
  v.fL[i] = 1u;  // listed for the ALT stage (consensus or verbatim copy of the best read)
  if (x.do_cons) {
    v.sz_aln[i] = (int64_t)x.n_others * L;
    v.sz_rd[i] = x.n_others;
    if (!cons_wave_eligible(v, L, x.n_others)) {   // anchor table in HBM only for the thread-kernel fallback
      int64_t npos = cons_npos(L, v.cfg.consensus_kmer_len, cons_skip(v.cfg, L));
      int64_t hs = 16; while (hs < 2 * npos + 2) hs <<= 1;
      v.sz_tab[i] = hs;
    }
  }
}

// E3: ALT work list (pN/pL/sc_* = exclusive scans over the calls of alt length, has-alt flag and the sizes)
// next free slot of the class list (order within a list is irrelevant).  On the GPU the lanes of a wave that append to
// the same class share one atomic: same-address atomics serialise in L2 at ~25 ns each, tens of thousands of them
// would dominate the kernel
SNF_HD int64_t class_list_slot(const View& v, int lid) {
#if defined(__HIP_DEVICE_COMPILE__)
  int64_t slot = 0;
  for (int c = 0; c < 8; c++) {
    const unsigned long long m = __ballot(lid == c);
    if (lid == c) {
      const int lane = (int)__lane_id();
      const int leader = __builtin_ctzll(m);
      unsigned long long base = 0;
      if (lane == leader) base = atomic_add_u64(&v.cnt->n_cls[c], (unsigned long long)__builtin_popcountll(m));
      base = __shfl(base, leader, 64);
      slot = (int64_t)base + __builtin_popcountll(m & ((1ull << lane) - 1ull));
    }
  }
  return slot;
#else
  return (int64_t)atomic_add_u64(&v.cnt->n_cls[lid], 1ull);
#endif
}

SNF_HD void e3_emit(int64_t i, const View& v);
SNF_HD void e3_conslist_body(int64_t i, const View& v) {
  const int64_t nc = v.cnt->n_calls;
  if (i == 0) {
    v.cnt->alt_total = v.pN[nc]; v.cnt->n_cons = v.pL[nc];
    v.cnt->tab_total = v.sc_tab[nc]; v.cnt->aln_total = v.sc_aln[nc]; v.cnt->n_cons_reads = v.sc_rd[nc];
    alt_decide(v, (int64_t)v.pN[nc]);
  }
  if (i >= nc) return;
  e3_emit(i, v);
}
// per-call part of E3 (offsets of call i are final in pN / pL / sc_*)
SNF_HD void e3_emit(int64_t i, const View& v) {
  snf_call_t& c = v.calls[i];
  CallX& x = v.callx[i];
  if (c.alt_len < 0) return;
  c.alt_off = v.pN[i]; x.alt_off = v.pN[i];
  uint32_t cid = v.pL[i];
  x.cons_id = (int32_t)cid;
  v.cons_call[cid] = (int32_t)i;
  v.cons_tab_off[cid] = v.sc_tab[i]; v.cons_tab_sz[cid] = v.sz_tab[i];
  v.cons_aln_off[cid] = v.sc_aln[i]; v.cons_read_off[cid] = v.sc_rd[i];
  if (v.sz_tab[i] > 0) atomic_add_u64((unsigned long long*)&v.cnt->n_cons_fallback, 1ull);   // rare
  if (!v.wave_path) return;
  // work item for the gfx950 workgroup kernels
  ConsDesc d;
  d.L = v.F_seq_len[x.best]; d.best_off = v.F_seq_off[x.best]; d.alt_off = x.alt_off; d.aln_off = v.sc_aln[i];
  d.read_off = x.flo;   // read list (crl_*) and kept flags (aln_kept_w) live in the refined cluster's own slot range
  d.n_others = x.n_others; d.skip = cons_skip(v.cfg, d.L);
  d.cls = x.do_cons ? (cons_class(v, d.L, x.n_others) ? cons_class(v, d.L, x.n_others) : 3) : 0;
  v.cdesc[cid] = d;
  // work list: LARGE calls are bucketed by work (others x length) and the kernel walks the heaviest bucket first, so
  // the few very long items do not end up as the tail of the launch
  int lid = d.cls;
  if (d.cls == 2) {
    const int64_t work = (int64_t)d.n_others * d.L;
    lid = work >= 32768 ? 2 : work >= 16384 ? 3 : work >= 8192 ? 4 : 5;
  } else if (d.cls == 3) lid = 6;
  else if (d.cls == 4) lid = 7;
  const int64_t slot = class_list_slot(v, lid);
  if (lid != 6) v.cls_list[lid][slot] = (int32_t)cid;   // list 6 (thread kernels e4/e5/e6) is only counted
}

SNF_HD uint64_t kmer_key(const uint8_t* s, int klen) {
  uint64_t k = 0;
  for (int i = 0; i < klen; i++) k = (k << 8) | s[i];
  return k;
}
// slot of a k-mer key in a table of hs (power of two, <= 2^24) entries: HIGH bits of the multiplicative hash - the low bits of a
// product depend only on the low bits of the key, i.e. on the first three or four bases of the k-mer, and k-mers that share
// them would all start probing at the same slot (measured: probe chains of tens of slots, the lookups were half of a read's time)
SNF_HD int64_t kmer_slot(uint64_t key, int64_t hs) { return (int64_t)((key * 0x9E3779B97F4A7C15ull) >> 40) & (hs - 1); }

// E4: anchor table of the best read: k-mers seen exactly once among the sampled positions (consensus.py:289-299)
SNF_HD void e4_anchor_body(int64_t cid, const View& v) {
  if (cid >= v.cnt->n_cons) return;
  int32_t ci = v.cons_call[cid];
  const CallX& x = v.callx[ci];
  int64_t L = v.F_seq_len[x.best];
  if (!x.do_cons) return;
  // (cid, read) work items
  int64_t r0 = v.cons_read_off[cid];
  for (int32_t r = 0; r < x.n_others; r++) { v.cr_call[r0 + r] = (int32_t)cid; v.cr_read[r0 + r] = r; }
  if (cons_wave_eligible(v, L, x.n_others)) return;  // e45w_consensus builds its table in LDS
  const uint8_t* B = v.pool + v.F_seq_off[x.best];
  int klen = v.cfg.consensus_kmer_len, skip = cons_skip_anchors(v, cid, L);
  int64_t t0 = v.cons_tab_off[cid], hs = v.cons_tab_sz[cid];
  uint64_t* key = v.tab_key + t0; int32_t* pos = v.tab_pos + t0; uint8_t* st = v.tab_state + t0;
  for (int64_t p = 0; p < hs; p++) st[p] = 0;
  for (int64_t i = 0; i < L - klen; i += skip) {
    uint64_t kk = kmer_key(B + i, klen);
    int64_t p = kmer_slot(kk, hs);
    while (st[p] && key[p] != kk) p = (p + 1) & (hs - 1);
    if (st[p] == 0) { st[p] = 1; key[p] = kk; pos[p] = (int32_t)i; }
    else st[p] = 2;
  }
}

// E5: align one other read against the best read's anchors -> aligned row (consensus.py:301-363)
SNF_HD void e5_align_body(int64_t j, const View& v) {
  if (j >= v.cnt->n_cons_reads) return;
  int32_t cid = v.cr_call[j], ridx = v.cr_read[j];
  int32_t ci = v.cons_call[cid];
  const CallX& x = v.callx[ci];
  if (cons_wave_eligible(v, v.F_seq_len[x.best], x.n_others)) return;
  // locate the ridx-th seq-bearing lead other than best, in cluster order
  int32_t slot = -1, seen = 0;
  for (int32_t k = 0; k < x.fn; k++) {
    int32_t s = v.FI[x.flo + k];
    if (v.F_seq_len[s] < 0 || s == x.best) continue;
    if (seen == ridx) { slot = s; break; }
    seen++;
  }
  int64_t L = v.F_seq_len[x.best];
  const uint8_t* B = v.pool + v.F_seq_off[x.best];
  const uint8_t* S = v.pool + v.F_seq_off[slot];
  int64_t SL = v.F_seq_len[slot];
  int klen = v.cfg.consensus_kmer_len, skip = cons_skip_reads(v, cid, L), maxshift = klen;
  int64_t t0 = v.cons_tab_off[cid], hs = v.cons_tab_sz[cid];
  const uint64_t* key = v.tab_key + t0; const int32_t* pos = v.tab_pos + t0; const uint8_t* st = v.tab_state + t0;
  uint8_t* row = v.aln + v.cons_aln_off[cid] + (int64_t)ridx * L;
  bool have_last = false; int64_t last_i = 0, last_j = 0, clen = 0, span = 0;
  for (int64_t jj = 0; jj < SL - klen; jj += skip) {
    uint64_t kk = kmer_key(S + jj, klen);
    int64_t p = kmer_slot(kk, hs);
    while (st[p] && key[p] != kk) p = (p + 1) & (hs - 1);
    if (st[p] != 1) continue;
    int64_t i = pos[p];
    if (iabs64(i - jj) > maxshift) continue;
    if (have_last && i <= last_i) continue;
    if (!have_last) {
      if (jj > 0) { for (int64_t q = 0; q < i; q++) row[q] = '-'; clen = i; }
    } else {
      int64_t fwd_i = i - last_i, fwd_j = jj - last_j;
      if (clen + fwd_j > L) fwd_j = L - clen;
      if (fwd_i == fwd_j && fwd_j > 0) {
        span += jj - last_j;
        int64_t m = 0;
        for (int64_t l = 1; l <= jj - last_j; l++) if (S[last_j + l] == B[last_i + l]) m++;
        double ident = (double)m / (double)(jj - last_j);
        if (ident >= 0.5) for (int64_t q = 0; q < fwd_j; q++) row[clen + q] = S[last_j + q];
        else for (int64_t q = 0; q < fwd_j; q++) row[clen + q] = '-';
        clen += fwd_j;
      } else if (fwd_j > 0) { for (int64_t q = 0; q < fwd_j; q++) row[clen + q] = '-'; clen += fwd_j; }
    }
    have_last = true; last_i = i; last_j = jj;
  }
  for (int64_t q = clen; q < L; q++) row[q] = '-';
  // keep a run only if it agrees with the best read: matches/len > 0.5 and matches > 5
  for (int64_t hh = 0; hh < L;) {
    if (row[hh] == '-') { hh++; continue; }
    int64_t h0 = hh, ident = 0;
    while (hh < L && row[hh] != '-') { ident += (B[hh] == row[hh]); hh++; }
    int64_t bl = hh - h0;
    if (!((double)ident / (double)bl > 0.5 && ident > 5)) for (int64_t q = h0; q < hh; q++) row[q] = '-';
  }
  v.aln_kept[j] = ((double)span / (double)L > 0.2) ? 1 : 0;
}

// E6: one output base per thread: column vote (consensus.py:365-380) or a verbatim copy of the best read
SNF_HD void e6_vote_body(int64_t col, const View& v) {
  if (col >= v.cnt->alt_total) return;
  // call owning this column: last call index with pN[i] <= col (pN = alt offsets over all calls)
  int64_t lo = 0, hi = v.cnt->n_calls;
  while (lo < hi) { int64_t mid = (lo + hi) >> 1; if ((int64_t)v.pN[mid + 1] <= col) lo = mid + 1; else hi = mid; }
  int64_t ci = lo;
  const CallX& x = v.callx[ci];
  if (v.wave_path && (!x.do_cons || cons_wave_eligible(v, v.F_seq_len[x.best], x.n_others))) return;  // e45w wrote it
  int64_t i = col - (int64_t)v.pN[ci];
  int64_t L = v.F_seq_len[x.best];
  uint8_t b = v.pool[v.F_seq_off[x.best] + i];
  uint8_t out = b;
  if (x.do_cons) {
    int32_t cid = x.cons_id;
    int64_t r0 = v.cons_read_off[cid];
    const uint8_t* rows = v.aln + v.cons_aln_off[cid];
    int64_t nkept = 0, nvotes = 0;
    for (int32_t r = 0; r < x.n_others; r++) if (v.aln_kept[r0 + r]) { nkept++; if (rows[(int64_t)r * L + i] != '-') nvotes++; }
    double maxal = (double)(1 + nkept);
    if (!(nvotes < 2 || (double)nvotes / maxal < 0.25)) {
      // util.most_common([best]+votes): (count, char) descending; replace iff top beats runner-up by >= 3
      int64_t c0 = -1, c1 = -1; int k0 = -1, k1 = -1, nd = 0;
      for (int32_t r = -1; r < x.n_others; r++) {
        uint8_t ch;
        if (r < 0) ch = b;
        else { if (!v.aln_kept[r0 + r]) continue; ch = rows[(int64_t)r * L + i]; if (ch == '-') continue; }
        bool seen = (r >= 0 && ch == b);  // first occurrence test among [best]+votes
        for (int32_t r2 = 0; r2 < r && !seen; r2++)
          if (v.aln_kept[r0 + r2] && rows[(int64_t)r2 * L + i] == ch) seen = true;
        if (seen) continue;
        int64_t cntc = (ch == b) ? 1 : 0;
        if (ch != '-')   // (the best read's own '-' is a character of the vote; a '-' in a row is a gap, never a vote)
          for (int32_t r2 = 0; r2 < x.n_others; r2++) if (v.aln_kept[r0 + r2] && rows[(int64_t)r2 * L + i] == ch) cntc++;
        nd++;
        if (cntc > c0 || (cntc == c0 && (int)ch > k0)) { c1 = c0; k1 = k0; c0 = cntc; k0 = ch; }
        else if (cntc > c1 || (cntc == c1 && (int)ch > k1)) { c1 = cntc; k1 = ch; }
      }
      if (nd > 1 && c0 - c1 >= 3) out = (uint8_t)k0;
    }
  }
  alt_base(v)[col] = out;
}

// Column vote of the LDS-vote consensus kernels (snf_wave_cons.h; consensus.py:365-380, util.most_common): `cnt4` = four
// 8-bit counters of the kept other reads' bases at this column (code (c >> 1) & 3: A 0, C 1, T 2, G 3), `esc` = the votes
// with any other byte as (column << 8 | byte), `bq` = the best read's base, `nkept` = kept other reads of the call.
// The output base is replaced iff at least 2 votes, votes / (1 + nkept) >= 0.25, more than one distinct character among
// [bq] + votes and the most common one ((count, char) descending) beats the runner-up by >= 3.
SNF_HD uint8_t vote_column(uint32_t cnt4, const uint32_t* esc, int n_esc, int q, uint8_t bq, int nkept) {
  const uint32_t ACTG = 0x47544341u;
  int ne = 0;
  for (int a = 0; a < n_esc; a++) ne += (int)(esc[a] >> 8) == q;
  const int nv = (int)(cnt4 & 0xffu) + (int)((cnt4 >> 8) & 0xffu) + (int)((cnt4 >> 16) & 0xffu) + (int)(cnt4 >> 24) + ne;
  if (nv < 2 || (double)nv / (double)(1 + nkept) < 0.25) return bq;
  const int cdb = (bq >> 1) & 3;
  const bool bq_plain = ((ACTG >> (8 * cdb)) & 0xffu) == bq;
  int c0 = -1, c1 = -1, k0 = -1, k1 = -1, nd = 0;
  for (int z = 0; z < 4; z++) {
    const int cntc = (int)((cnt4 >> (8 * z)) & 0xffu) + ((bq_plain && z == cdb) ? 1 : 0);
    if (!cntc) continue;
    const int c = (int)((ACTG >> (8 * z)) & 0xffu);
    nd++;
    if (cntc > c0 || (cntc == c0 && c > k0)) { c1 = c0; k1 = k0; c0 = cntc; k0 = c; }
    else if (cntc > c1 || (cntc == c1 && c > k1)) { c1 = cntc; k1 = c; }
  }
  if (ne > 0 || !bq_plain) {   // rare: characters outside A/C/G/T
    bool bq_seen = bq_plain;
    for (int a = 0; a < n_esc; a++) {
      if ((int)(esc[a] >> 8) != q) continue;
      const int c = (int)(esc[a] & 0xffu);
      bool first = true;
      for (int a2 = 0; a2 < a && first; a2++) if (esc[a2] == esc[a]) first = false;
      if (!first) continue;
      int cntc = (c == (int)bq) ? 1 : 0;
      if (c == (int)bq) bq_seen = true;
      for (int a2 = a; a2 < n_esc; a2++) cntc += esc[a2] == esc[a];
      nd++;
      if (cntc > c0 || (cntc == c0 && c > k0)) { c1 = c0; k1 = k0; c0 = cntc; k0 = c; }
      else if (cntc > c1 || (cntc == c1 && c > k1)) { c1 = cntc; k1 = c; }
    }
    if (!bq_seen) {
      const int c = (int)bq, cntc = 1;
      nd++;
      if (cntc > c0 || (cntc == c0 && c > k0)) { c1 = c0; k1 = k0; c0 = cntc; k0 = c; }
      else if (cntc > c1 || (cntc == c1 && c > k1)) { c1 = cntc; k1 = c; }
    }
  }
  (void)k1;
  return (nd > 1 && c0 - c1 >= 3) ? (uint8_t)k0 : bq;
}

}  // namespace snf
