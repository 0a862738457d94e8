// snf_stage_call.h - kernel bodies: per-cluster refinement and candidate calls.
//
// Reference semantics: merge_inner (cluster.py:85-122), resplit (cluster.py:125-161), resplit_bnd
// (cluster.py:164-216), Cluster.get_sa_count (cluster.py:79-82), sv.call_from / resolve_bnd /
// calculate_bounds (sv.py:484-639), util.center/trim/stdev/most_common_top (util.py:25-103).
#pragma once
#include <cstddef>
#include "snf_stage_cluster.h"
#include "snf_cov.h"

namespace snf {

SNF_HD bool lead_has_seq(const View& v, uint32_t o) { return v.in_seq_len[o] >= 0 && !v.seqnull[o]; }

// ------------------------------------------------------------------------------------------ D1
// (the refinement bodies read the cluster's leads from the packed records R = Lrec + lo: one 64-byte line per lead instead
// of a dozen columns gathered through L[] - and one array to stage when a wave keeps the cluster in LDS, x_big<0>)
struct LessQnameIdx {
  const LeadRec* R;
  SNF_HD bool operator()(int32_t a, int32_t b) const {
    uint32_t qa = R[a].qname, qb = R[b].qname;
    return qa != qb ? qa < qb : a < b;
  }
};
struct LessFaRefIdx {
  const LeadRec* R; const int32_t* fa;
  SNF_HD bool operator()(int32_t a, int32_t b) const {
    if (fa[a] != fa[b]) return fa[a] < fa[b];
    int32_t ra = R[a].ref_start, rb = R[b].ref_start;
    return ra != rb ? ra < rb : a < b;
  }
};
struct LessBinIdx {
  const int32_t* svlen; int32_t rb;
  SNF_HD int64_t bin(int32_t k) const { int64_t a = svlen[k] < 0 ? -(int64_t)svlen[k] : svlen[k]; return (a / rb) * rb; }
  SNF_HD bool operator()(int32_t a, int32_t b) const {
    int64_t ba = bin(a), bb = bin(b);
    return ba != bb ? ba < bb : a < b;
  }
};
struct LessIdentIdx {
  const LeadRec* R;
  SNF_HD bool operator()(int32_t a, int32_t b) const {
    if (R[a].mate_contig != R[b].mate_contig) return R[a].mate_contig < R[b].mate_contig;
    if (R[a].first != R[b].first) return R[a].first < R[b].first;
    return a < b;
  }
};
struct LessFaPosbinIdx {
  const LeadRec* R; const int32_t* fa; int32_t thr;
  SNF_HD int64_t pb(int32_t a) const { return thr > 0 ? ((int64_t)R[a].mate_pos / thr) * thr : 0; }
  SNF_HD bool operator()(int32_t a, int32_t b) const {
    if (fa[a] != fa[b]) return fa[a] < fa[b];
    int64_t pa = pb(a), pbb = pb(b);
    return pa != pbb ? pa < pbb : a < b;
  }
};

SNF_HD void rc_emit(const View& v, int32_t pos, int32_t n, int32_t c, bool keeplong) {
  v.rcflag[pos] = 1;
  v.rc_n_s[pos] = n;
  v.rc_cl_s[pos] = c;
  v.rc_keeplong_s[pos] = keeplong ? 1 : 0;
}

// room for a fused sequence behind the input sequences; in a wave that runs the body uniformly one lane reserves for all
SNF_HD unsigned long long pool_reserve(const View& v, unsigned long long nbytes) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (v.wave_uniform) {
    unsigned long long o = 0;
    if ((threadIdx.x & 63) == 0) o = atomicAdd(&v.cnt->pool_extra_used, nbytes);
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(o >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)o);
  }
#endif
  return atomic_add_u64(&v.cnt->pool_extra_used, nbytes);
}

// one merged cluster -> fused leads (F) + refined clusters in final lead order (FI)
SNF_HD void d1_refine_body(int64_t c, const View& v) {
  if (c >= v.cnt->n_clusters) return;
  const snf_config_t& cfg = v.cfg;
  int32_t h = v.cl_head[c];
  int32_t lo = v.seed_lo[h], hi = v.seed_hi[v.c_last[h]];
  int32_t n = hi - lo;
  if (n <= 0) return;
  if (v.wave_path && (n <= 64 || v.big_wave)) return;  // d1w_refine (snf_wave_refine.h) / x_big<0> (a wave of its own)
  int svtype = grp_svtype(v.seed_grp[h]);
  int32_t *a0 = v.w0 + lo, *a1 = v.w1 + lo, *a2 = v.w2 + lo, *a3 = v.w3 + lo, *a4 = v.w4 + lo, *a5 = v.w5 + lo,
          *a6 = v.w6 + lo, *stmp = v.w7 + lo;
  if (v.stage_w) {   // the calling wave holds the cluster's scratch rows in LDS
    const int32_t cap = v.stage_cap;
    a0 = v.stage_w; a1 = a0 + cap; a2 = a1 + cap; a3 = a2 + cap; a4 = a3 + cap; a5 = a4 + cap; a6 = a5 + cap; stmp = a6 + cap;
  }
  const bool uni = v.wave_uniform != 0;
  int32_t* Forig = v.F_orig + lo; int32_t* Fsvlen = v.F_svlen + lo; int32_t* Fseqlen = v.F_seq_len + lo;
  int64_t* Fseqoff = v.F_seq_off + lo;
  int32_t m = 0;

  const LeadRec* R = v.stage_R ? v.stage_R : v.Lrec + lo;   // the cluster's leads, cluster order (in LDS when staged)
  if (svtype == SNF_INS || svtype == SNF_DEL) {
    // ---- merge_inner: group by read (first appearance), sort by ref_start, fuse neighbours
    int thr = v.c_repeat[h] ? -1 : cfg.cluster_merge_pos;
    for (int32_t j = 0; j < n; j++) a0[j] = j;
    SNF_SORT(uni, a0, (int64_t)n, (LessQnameIdx{R}), stmp);
    for (int32_t x = 0; x < n;) {
      int32_t y = x; uint32_t q = R[a0[x]].qname;
      while (y < n && R[a0[y]].qname == q) { a1[a0[y]] = a0[x]; y++; }
      x = y;
    }
    for (int32_t j = 0; j < n; j++) a0[j] = j;
    SNF_SORT(uni, a0, (int64_t)n, (LessFaRefIdx{R, a1}), stmp);
    for (int32_t x = 0; x < n;) {
      int32_t y_end = x; while (y_end < n && a1[a0[y_end]] == a1[a0[x]]) y_end++;
      LeadRec rh = R[a0[x]];       // head of the part being fused
      int64_t cs = rh.svlen; bool seq_ok = rh.seq_len >= 0; int64_t seq_total = seq_ok ? rh.seq_len : 0;
      int32_t part_start = x;
      int32_t l_re = rh.ref_end, l_qe = rh.qry_end, l_rs = rh.ref_start, l_qs = rh.qry_start;
      for (int32_t y = x + 1; y <= y_end; y++) {
        bool flush = (y == y_end);
        LeadRec rt{};
        if (!flush) {
          rt = R[a0[y]];
          int32_t rs = rt.ref_start, qs = rt.qry_start;
          bool mg = (thr == -1) ||
                    (((iabs64((int64_t)rs - l_re) < thr || iabs64((int64_t)rs - l_rs) < thr) &&
                      (iabs64((int64_t)qs - l_qe) < thr || iabs64((int64_t)qs - l_qs) < thr)) &&
                     (rh.strand == rt.strand));
          if (mg) {
            cs += rt.svlen;
            if (rt.seq_len < 0 || !seq_ok) seq_ok = false; else seq_total += rt.seq_len;
          } else flush = true;
        }
        if (flush) {
          Forig[m] = (int32_t)rh.orig; Fsvlen[m] = (int32_t)cs; v.F_lpos[lo + m] = lo + a0[part_start];
          int32_t nparts = (y < y_end ? y : y_end) - part_start;
          if (!seq_ok) { Fseqlen[m] = -1; Fseqoff[m] = 0; }
          else if (nparts == 1) { Fseqlen[m] = rh.seq_len; Fseqoff[m] = rh.seq_off; }
          else {  // curr_lead.seq += to_merge.seq: new string in the fused part of the pool
            int64_t off = v.pool_extra_base + (int64_t)pool_reserve(v, (unsigned long long)seq_total);
            if (off + seq_total > v.pool_cap) { atomic_or_i32(&v.cnt->overflow, 1); Fseqlen[m] = -1; Fseqoff[m] = 0; }
            else {
              int64_t w = off;
              for (int32_t z = part_start; z < part_start + nparts; z++) {
                const LeadRec& rz = R[a0[z]];
                const uint8_t* src = v.pool + rz.seq_off;
#if defined(__HIP_DEVICE_COMPILE__)
                if (uni) {   // the wave runs this body in lock step: the lanes share the bytes (a serial load -> store chain per
                             // byte through the same array costs a memory round trip each)
                  for (int32_t b = (int32_t)(threadIdx.x & 63); b < rz.seq_len; b += 64) v.pool[w + b] = src[b];
                  w += rz.seq_len;
                } else
#endif
                for (int32_t b = 0; b < rz.seq_len; b++) v.pool[w++] = src[b];
              }
              Fseqlen[m] = (int32_t)seq_total; Fseqoff[m] = off;
            }
          }
          m++;
          if (y < y_end) {
            rh = rt; cs = rt.svlen; seq_ok = rt.seq_len >= 0; seq_total = seq_ok ? rt.seq_len : 0;
            part_start = y;
          }
        }
        if (y < y_end) { l_re = rt.ref_end; l_qe = rt.qry_end; l_rs = rt.ref_start; l_qs = rt.qry_start; }
      }
      x = y_end;
    }
  } else {
    for (int32_t j = 0; j < n; j++) {
      const LeadRec& r = R[j];
      Forig[j] = (int32_t)r.orig; Fsvlen[j] = r.svlen; v.F_lpos[lo + j] = lo + j;
      bool hs = r.seq_len >= 0;
      Fseqlen[j] = hs ? r.seq_len : -1; Fseqoff[j] = hs ? r.seq_off : 0;
    }
    m = n;
  }
  int32_t* FI = v.FI + lo;

  if (svtype == SNF_BND) {
    // ---- resplit_bnd
    if (m <= 1 || cfg.dev_no_resplit) {
      for (int32_t k = 0; k < m; k++) FI[k] = lo + k;
      rc_emit(v, lo, m, (int32_t)c, true);
      return;
    }
    int thr = cfg.cluster_merge_bnd;
    for (int32_t j = 0; j < m; j++) a0[j] = j;
    SNF_SORT(uni, a0, (int64_t)m, (LessIdentIdx{R}), stmp);
    for (int32_t x = 0; x < m;) {
      int32_t y = x; const LeadRec& rx = R[a0[x]];
      while (y < m) { const LeadRec& ry = R[a0[y]];
        if (ry.mate_contig != rx.mate_contig || ry.first != rx.first) break;
        a1[a0[y]] = a0[x]; y++; }
      x = y;
    }
    for (int32_t j = 0; j < m; j++) a0[j] = j;
    LessFaPosbinIdx lp{R, a1, thr};
    SNF_SORT(uni, a0, (int64_t)m, lp, stmp);
    int32_t start = 0; int64_t last_bin = lp.pb(a0[0]);
    for (int32_t x = 0; x < m; x++) {
      FI[x] = lo + a0[x];
      if (x > 0) {
        int64_t pb = lp.pb(a0[x]);
        bool brk = a1[a0[x]] != a1[a0[x - 1]] || (pb - last_bin > thr);
        if (brk) { rc_emit(v, lo + start, x - start, (int32_t)c, false); start = x; }
        last_bin = pb;
      }
    }
    rc_emit(v, lo + start, m - start, (int32_t)c, false);
    return;
  }

  // ---- resplit on svlen
  if (cfg.dev_no_resplit_repeat || cfg.dev_no_resplit) {
    for (int32_t k = 0; k < m; k++) FI[k] = lo + k;
    rc_emit(v, lo, m, (int32_t)c, true);
    return;
  }
  LessBinIdx lb{Fsvlen, cfg.cluster_resplit_binsize};
  for (int32_t k = 0; k < m; k++) a0[k] = k;
  SNF_SORT(uni, a0, (int64_t)m, lb, stmp);
  // segments = distinct bins: a1 seg start, a2 seg key, a3 surviving list, a4 head, a5 tail, a6 next
  int32_t nb = 0;
  for (int32_t x = 0; x < m; x++)
    if (x == 0 || lb.bin(a0[x]) != lb.bin(a0[x - 1])) { a1[nb] = x; a2[nb] = (int32_t)lb.bin(a0[x]); nb++; }
  for (int32_t s = 0; s < nb; s++) { a3[s] = s; a4[s] = s; a5[s] = s; a6[s] = -1; }
  int32_t cntc = nb, i = 1;
  while (cntc > 1 && i < cntc) {
    int32_t im1 = (i == 0) ? cntc - 1 : i - 1;  // Python negative index: new_clusters[-1]
    int64_t last = a2[a3[im1]], curr = a2[a3[i]];
    int64_t mn = curr < last ? curr : last;
    double t = (double)mn * cfg.cluster_merge_len;
    double thr = ((double)cfg.minsvlen >= t) ? (double)cfg.minsvlen : t;
    int64_t diff = curr > last ? curr - last : last - curr;
    if ((double)diff <= thr) {
      int32_t cb = a3[i], lbk = a3[im1];
      a6[a5[cb]] = a4[lbk];  // bins_leads[curr].extend(bins_leads[last])
      a5[cb] = a5[lbk];
      for (int32_t t2 = im1; t2 + 1 < cntc; t2++) a3[t2] = a3[t2 + 1];
      cntc--;
      i = (i - 2 > 0) ? i - 2 : 0;
    } else i++;
  }
  int32_t outp = 0;
  for (int32_t t2 = 0; t2 < cntc; t2++) {
    int32_t start = outp;
    for (int32_t s = a4[a3[t2]]; s >= 0; s = a6[s]) {
      int32_t xe = (s + 1 < nb) ? a1[s + 1] : m;
      for (int32_t x = a1[s]; x < xe; x++) FI[outp++] = lo + a0[x];
    }
    rc_emit(v, lo + start, outp - start, (int32_t)c, true);
  }
}

// D1b: dense refined-cluster table (rcscan = exclusive scan of rcflag over the F slot space)
SNF_HD void d1b_emit(int64_t pos, const View& v);
SNF_HD void d1b_rctable_body(int64_t pos, const View& v) {
  if (pos == 0) v.cnt->n_rc = v.rcscan[v.NS];
  d1b_emit(pos, v);
}
SNF_HD void d1b_emit(int64_t pos, const View& v) {
  if (pos < v.cnt->NF && v.rcflag[pos]) {
    uint32_t r = v.rcscan[pos];
    v.rc_lo[r] = (int32_t)pos; v.rc_n[r] = v.rc_n_s[pos]; v.rc_cluster[r] = v.rc_cl_s[pos];
    v.rc_keeplong[r] = v.rc_keeplong_s[pos];
  }
}

// ------------------------------------------------------------------------------------------ D2
SNF_HD int64_t distinct_sorted_i32(int32_t* a, int64_t n) {
  if (n == 0) return 0;
  int64_t k = 1;
  for (int64_t i = 1; i < n; i++) if (a[i] != a[k - 1]) a[k++] = a[i];
  return k;
}
SNF_HD bool contains_sorted_i32(const int32_t* a, int64_t n, int32_t x) {
  int64_t p = lower_bound_i32(a, 0, n, x);
  return p < n && a[p] == x;
}

// sv.call_from for one refined cluster -> candidate record (or nothing: minsvlen_screen)
SNF_HD void d2_call_body(int64_t r, const View& v) {
  if (r == 0) v.cdflag[v.NS] = 0;
  if (r >= v.cnt->n_rc) { v.cdflag[r] = 0; return; }
  const snf_config_t& cfg = v.cfg;
  int32_t flo = v.rc_lo[r], n = v.rc_n[r], c = v.rc_cluster[r];
  if (v.wave_path && (n <= 64 || v.big_wave)) return;  // d2w_call / x_big<1> (snf_wave_call.h)
  int32_t h = v.cl_head[c];
  int g = v.seed_grp[h], svtype = grp_svtype(g), task = grp_task(g);
  int32_t *a0 = v.w0 + flo, *a1 = v.w1 + flo, *a2 = v.w2 + flo, *a3 = v.w3 + flo, *stmp = v.w7 + flo;
  const bool uni = v.wave_uniform != 0;
  const int32_t* FI = v.FI + flo;
  // x_big<1> (uniform mode): the fill loops take every 64th lead per lane (SNF_LEADS), sums are wave reductions, and the rows
  // live in LDS when the cluster fits (the sorted read names are copied to their global row at the end: d3_rnames reads them)
#if defined(__HIP_DEVICE_COMPILE__)
  const int l0 = uni ? (int)(threadIdx.x & 63) : 0, lstep = uni ? 64 : 1;
  const bool lds_rows = uni && v.stage_w != nullptr && n <= v.stage_cap;
  int32_t* const a1_global = a1;
  if (lds_rows) { a0 = v.stage_w; a1 = a0 + v.stage_cap; a2 = a1 + v.stage_cap; a3 = a2 + v.stage_cap; stmp = a3 + v.stage_cap; }
#define SNF_LEADS(k) for (int32_t k = l0; k < n; k += lstep)
#define SNF_UNI_SYNC() do { if (uni) __syncthreads(); } while (0)
#define SNF_UNI_SUM(x) do { if (uni) { for (int d_ = 32; d_ >= 1; d_ >>= 1) x += __shfl_xor(x, d_, 64); } } while (0)
#else
  const bool lds_rows = false;
#define SNF_LEADS(k) for (int32_t k = 0; k < n; k++)
#define SNF_UNI_SYNC() do { } while (0)
#define SNF_UNI_SUM(x) do { } while (0)
#endif
  v.cdflag[r] = 0;
  SNF_LEADS(k) { a0[k] = v.F_svlen[FI[k]]; v.F_sel[FI[k]] = 1; }
  SNF_SORT(uni, a0, (int64_t)n, LessI32{}, stmp);
  int64_t svlen = center_sorted(a0, n);
  bool single = svtype == SNF_SINGLE_LEFT || svtype == SNF_SINGLE_RIGHT;
  if (!single && svtype != SNF_BND && iabs64(svlen) < cfg.minsvlen_screen) return;

  SNF_LEADS(k) a1[k] = (int32_t)v.in_qname[v.F_orig[FI[k]]];
  SNF_SORT(uni, a1, (int64_t)n, LessI32{}, stmp);
  int64_t nq = distinct_sorted_i32(a1, n);
  int64_t support = nq, support_long = 0;
  int32_t llo = v.seedL_lo[h], lhi = v.seedL_hi[v.c_last[h]];
  bool keeplong = v.rc_keeplong[r] && svtype == SNF_INS;
  if (svtype == SNF_INS && svlen >= cfg.long_ins_length) {
    for (int32_t x = llo; x < lhi; x++) {
      int32_t q = (int32_t)v.in_qname[v.LL[x]];
      bool first = true;
      for (int32_t y = llo; y < x; y++) if ((int32_t)v.in_qname[v.LL[y]] == q) { first = false; break; }
      if (first) { support_long++; if (!contains_sorted_i32(a1, nq, q)) support++; }
    }
  }
  SNF_UNI_SYNC();   // (distinct_sorted_i32 compacted a1 in place on every lane)
  SNF_LEADS(k) a2[k] = v.in_ref_start[v.F_orig[FI[k]]];
  SNF_SORT(uni, a2, (int64_t)n, LessI32{}, stmp);
  int64_t ref_start = center_sorted(a2, n);
  double stdev_pos = stdev_trim_sorted(a2, n);
  double stdev_len = NAN; bool precise;
  if (svtype != SNF_BND) { stdev_len = stdev_trim_sorted(a0, n); precise = (stdev_pos + stdev_len < (double)cfg.precise); }
  else precise = stdev_pos < (double)cfg.precise;
  int64_t svstart, svend;
  if (svtype == SNF_INS) { svstart = ref_start; svend = ref_start; }
  else if (svtype == SNF_DEL) { svstart = ref_start + svlen; svend = ref_start; }
  else { svstart = ref_start; svend = svstart + iabs64(svlen); }
  int64_t msum = 0, fwd = 0, sa = 0, src_noninline = 0; double nmsum = 0;
  SNF_LEADS(k) {
    uint32_t o = (uint32_t)v.F_orig[FI[k]];
    msum += v.in_mapq[o]; fwd += (v.in_strand[o] == 0); sa += v.in_is_sa[o];
    src_noninline += (v.in_source[o] != SNF_SRC_INLINE);
  }
  SNF_UNI_SUM(msum); SNF_UNI_SUM(fwd); SNF_UNI_SUM(sa); SNF_UNI_SUM(src_noninline);
  if (cfg.qc_nm_measure) for (int32_t k = 0; k < n; k++) nmsum += v.in_nm[(uint32_t)v.F_orig[FI[k]]];   // Python sum(): left to right
  int64_t n_all = n;
  if (keeplong) { for (int32_t x = llo; x < lhi; x++) sa += v.in_is_sa[v.LL[x]]; n_all += lhi - llo; }

  snf_call_t cc;
  memset(&cc, 0, sizeof(cc));
  cc.task_index = task; cc.svtype = svtype; cc.pos = (int32_t)svstart; cc.end = (int32_t)svend; cc.svlen = (int32_t)svlen;
  cc.support = (int32_t)support; cc.support_long = -1; cc.support_sa = -1;
  cc.qual = (int32_t)((double)msum / (double)n); cc.precise = precise; cc.fwd = (int32_t)fwd; cc.rev = (int32_t)(n - fwd);
  cc.qc = 1; cc.filter = SNF_F_PASS;
  cc.nm = cfg.qc_nm_measure ? nmsum / (double)n : -1.0;
  cc.stdev_pos = stdev_pos; cc.stdev_len = stdev_len;
  cc.sa_count = (int32_t)sa; cc.sa_frac = (double)sa / (double)n_all; cc.n_leads = n;
  cc.mate_contig = -1; cc.gt_hp = -1; cc.gt_ps = -1; cc.vaf = NAN; cc.alt_len = -1;
  cc.cluster_start = v.seed_start[h]; cc.cluster_end = v.c_end[h];
  cc.cluster_seed_index = v.prefilter ? -1 : v.seed_bin[h] - v.grp_first_bin[g];   // (counts ALL occupied bins: not known behind the prefilter)
  int64_t rn_len = support;
  if (svtype == SNF_BND) {  // resolve_bnd
    SNF_LEADS(k) a3[k] = v.in_mate_contig[v.F_orig[FI[k]]];
    SNF_SORT(uni, a3, (int64_t)n, LessI32{}, stmp);
    int32_t mc = a3[0]; int64_t bc = 0;
    for (int32_t x = 0; x < n;) { int32_t y = x; while (y < n && a3[y] == a3[x]) y++; if (y - x > bc) { bc = y - x; mc = a3[x]; } x = y; }
    int32_t ns = 0; int64_t nfirst = 0, nrev = 0;
    for (int32_t k = 0; k < n; k++) {
      uint32_t o = (uint32_t)v.F_orig[FI[k]];
      bool sel = v.in_mate_contig[o] == mc;
      v.F_sel[FI[k]] = sel ? 1 : 0;
      if (sel) { a3[ns] = v.in_mate_pos[o]; a2[ns] = (int32_t)v.in_qname[o]; nfirst += v.in_first[o]; nrev += v.in_rev[o]; ns++; }
    }
    SNF_SORT(uni, a3, (int64_t)ns, LessI32{}, stmp);
    SNF_SORT(uni, a2, (int64_t)ns, LessI32{}, stmp);
    // SUPPORT counts the reads of the selected leads; RNAMES stays the read set of the whole cluster (sv.py:555 is taken
    // before resolve_bnd narrows the leads, sv.py:636) - a1 / nq / rn_len are left as they are.  The two differ only when
    // a cluster mixes mate contigs, i.e. with --dev-no-resplit (resplit_bnd groups by mate contig).
    cc.support = (int32_t)distinct_sorted_i32(a2, ns);
    cc.mate_contig = mc; cc.mate_ref_start = center_sorted(a3, ns);
    cc.bnd_is_first = (nfirst > ns - nfirst) ? 1 : 0;   // most_common_top: ties -> False
    cc.bnd_is_reverse = (nrev > ns - nrev) ? 1 : 0;
    cc.n_leads = ns;
  } else if (svtype == SNF_INS) cc.support_long = (int32_t)support_long;
  else if (svtype == SNF_DEL) cc.support_sa = (int32_t)src_noninline;
  cc.rn_len = (int32_t)rn_len;
  cc.rn_off = nq;  // stash: number of distinct names already sorted in w1[flo..]
  v.cand[r] = cc;
  CallX x; x.rc = (int32_t)r; x.cluster = c; x.flo = flo; x.fn = n; x.best = -1; x.n_others = 0; x.do_cons = 0; x.cons_id = -1; x.alt_off = 0;
  x.rn_nq = (int32_t)nq; x._pad = 0;
  x.ag_valid = 0; x.ag_nstrands = x.ag_close_edge = x.ag_hp_val = x.ag_hp_support = x.ag_hp_other = x.ag_ps_val = x.ag_ps_support = x.ag_ps_other = x.ag_has_nm = 0; x.ag_nm_mean = 0.0;   // (collected by the finalize body itself)
  if (svtype == SNF_INS && !cfg.symbolic) {
    // best lead of annotate_sv (postprocessing.py:33-66): first argmin of |len(seq) - svlen| + |ref_start - pos| * 1.5
    int32_t best = -1, cnt = 0; double best_diff = 0; int32_t best_k = 0x7fffffff;
    SNF_LEADS(k) {
      int32_t s = FI[k];
      if (v.F_seq_len[s] < 0) continue;
      double d = (double)iabs64((int64_t)v.F_seq_len[s] - cc.svlen) +
                 (double)iabs64((int64_t)v.in_ref_start[(uint32_t)v.F_orig[s]] - cc.pos) * 1.5;
      if (best < 0 || d < best_diff) { best = s; best_diff = d; best_k = k; }
      cnt++;
    }
#if defined(__HIP_DEVICE_COMPILE__)
    if (uni) {   // the lanes' minima: smallest d, then the earliest lead (first argmin)
      for (int d_ = 32; d_ >= 1; d_ >>= 1) {
        const int32_t ob = __shfl_xor(best, d_, 64), ok = __shfl_xor(best_k, d_, 64); const double od = __shfl_xor(best_diff, d_, 64);
        if (ob >= 0 && (best < 0 || od < best_diff || (od == best_diff && ok < best_k))) { best = ob; best_diff = od; best_k = ok; }
      }
      SNF_UNI_SUM(cnt);
    }
#endif
    if (best >= 0) { x.best = best; x.n_others = cnt - 1; x.do_cons = (x.n_others >= cfg.consensus_min_reads && !cfg.no_consensus) ? 1 : 0; }
    if (best >= 0 && v.wave_path) {  // read list for the workgroup consensus kernel (see d2w_call)
#if defined(__HIP_DEVICE_COMPILE__)
      if (uni) {
        int32_t w = 0;
        for (int32_t base = 0; base < n; base += 64) {      // list order = lead order: ballots per 64 leads
          const int32_t k = base + l0;
          const int32_t s = k < n ? FI[k] : 0;
          const bool oth = k < n && v.F_seq_len[s] >= 0 && s != best;
          const unsigned long long om = __ballot(oth);
          if (oth) { const int32_t q = w + __builtin_popcountll(om & ((1ull << l0) - 1ull)); v.crl_off[flo + q] = v.F_seq_off[s]; v.crl_len[flo + q] = v.F_seq_len[s]; }
          w += __builtin_popcountll(om);
        }
      } else
#endif
      {
        int32_t w = 0;
        for (int32_t k = 0; k < n; k++) {
          int32_t s = FI[k];
          if (v.F_seq_len[s] < 0 || s == best) continue;
          v.crl_off[flo + w] = v.F_seq_off[s]; v.crl_len[flo + w] = v.F_seq_len[s]; w++;
        }
      }
    }
  }
#if defined(__HIP_DEVICE_COMPILE__)
  if (lds_rows) {   // the distinct sorted read names stay in w1 for d3_rnames
    __syncthreads();
    for (int32_t k = l0; k < (int32_t)nq; k += 64) a1_global[k] = a1[k];
  }
#endif
#undef SNF_LEADS
#undef SNF_UNI_SYNC
#undef SNF_UNI_SUM
  v.candx[r] = x;
  v.cdflag[r] = 1;
}

// D3a: compaction (cdscan = exclusive scan of cdflag) -> calls in candidate order
SNF_HD void d3_compact_emit(int64_t r, const View& v) {
  if (r < v.cnt->n_rc && v.cdflag[r]) { uint32_t i = v.cdscan[r]; v.calls[i] = v.cand[r]; v.callx[i] = v.candx[r]; }
}
SNF_HD void d3_compact_body(int64_t r, const View& v) {
  if (r == 0) { v.cnt->n_calls = v.cdscan[v.NS]; }
  d3_compact_emit(r, v);
}

// D3b: per task offsets into calls (calls are sorted by task), T+1 entries; task status / stale BND `end`
// (postprocessing.py:84-106); thread 0 publishes the counters to the pinned result block so that the host can size
// the ALT chain while the rest of the candidate stage is still running on the side stream
SNF_HD int64_t task_lower_bound(const View& v, int64_t t) {
  int64_t lo = 0, hi = v.cnt->n_calls;
  while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (v.calls[mid].task_index < t) lo = mid + 1; else hi = mid; }
  return lo;
}
// threads of the two launches that publish per-task words and the counters into the pinned result block (d3_taskoff, z1_results):
// at least 128, so that the ~60 words of the counters and the striped sums are one PCIe store / one short loop per thread whatever
// the number of tasks (with T + 1 threads a one-contig batch had two threads storing 30 words each, one after the other: 26 us
// at the end of a 0.5-ms pass)
SNF_HD int64_t tail_threads(const View& v) { return v.T + 1 > 128 ? (int64_t)v.T + 1 : 128; }
SNF_HD void d3_taskoff_body(int64_t t, const View& v) {
  const int64_t lo = t <= v.T ? task_lower_bound(v, t) : 0;
  if (t <= v.T) v.t_call_off[t] = lo;
  if (t < v.T) {
    const int64_t hi = task_lower_bound(v, t + 1);
    int64_t a = lo, b = hi;
    while (a < b) { int64_t mid = (a + b) >> 1; if (v.calls[mid].svtype < SNF_BND) a = mid + 1; else b = mid; }
    v.t_status[t] = SNF_TASK_OK; v.t_stale_end[t] = 0;
    if (a < hi && v.calls[a].svtype == SNF_BND) {
      if (a == lo) v.t_status[t] = SNF_TASK_ERR_UNBOUND_END;   // UnboundLocalError: local variable 'end'
      else {
        const snf_call_t& p = v.calls[a - 1];
        v.t_stale_end[t] = p.svtype == SNF_INS ? p.pos + 1 : (int32_t)((int64_t)p.pos + iabs64(p.svlen));
      }
    }
  }
  {  // the counters go to the pinned result block: the T + 1 threads of this launch share the words (one thread storing ~40 words over
     // PCIe one after the other was most of this kernel's 20 us, and the kernel sits on the critical path of the pass)
    const unsigned long long* s = (const unsigned long long*)v.cnt; unsigned long long* d = (unsigned long long*)v.res_cnt;
    for (size_t k = (size_t)t; k < sizeof(Counts) / 8; k += (size_t)tail_threads(v)) d[k] = s[k];
  }
}

// D3c: sv ids, rnames offsets helper
SNF_HD void d3_svid_body(int64_t i, const View& v) {
  if (i >= v.cnt->n_calls) { v.rnf[i] = 0; return; }
  snf_call_t& c = v.calls[i];
  int t = c.task_index;
  c.sv_id = v.t_sv_id_start[t] + (int32_t)(i - v.t_call_off[t]);
  v.rnf[i] = (uint32_t)c.rn_len;  // scanned into rn_off
  if (i == 0) v.rnf[v.NS] = 0;
}

// D3d: supporting read names (pN = exclusive scan of rn_len)
SNF_HD void d3_rnames_emit(int64_t i, const View& v);
SNF_HD void d3_rnames_body(int64_t i, const View& v) {
  if (i == 0) { v.cnt->rn_total = v.rnp[v.NS]; *v.res_rn_total = v.rnp[v.NS]; }
  d3_rnames_emit(i, v);
}
SNF_HD void d3_rnames_emit(int64_t i, const View& v) {
  if (i >= v.cnt->n_calls) return;
  snf_call_t& c = v.calls[i];
  const CallX& x = v.callx[i];
  int64_t nq = x.rn_nq;   // stashed by d2 (not in c.rn_off: this function may run again - late pass, repeated finalize)
  int64_t off = v.rnp[i];
  const int32_t* a1 = v.w1 + x.flo;
  for (int64_t k = 0; k < nq; k++) v.rnames[off + k] = (uint32_t)a1[k];
  int64_t w = nq;
  if (c.svtype == SNF_INS && c.rn_len > nq) {
    int32_t h = v.cl_head[x.cluster];
    int32_t llo = v.seedL_lo[h], lhi = v.seedL_hi[v.c_last[h]];
    for (int32_t p = llo; p < lhi; p++) {
      int32_t q = (int32_t)v.in_qname[v.LL[p]];
      bool first = true;
      for (int32_t y = llo; y < p; y++) if ((int32_t)v.in_qname[v.LL[y]] == q) { first = false; break; }
      if (first && !contains_sorted_i32(a1, nq, q)) v.rnames[off + w++] = (uint32_t)q;
    }
  }
  c.rn_off = off;
}

// ------------------------------------------------------------------------------------------ coverage
// coverage(x) of the reference's dense uint16 vector (leadprov.py:451,510) as rank queries:
// #(start <= x) - #(end <= x) over the task's reads, modulo 2^16
SNF_HD Reads reads_of(const View& v, int t) {
  Reads q;
  q.r_start = v.r_start; q.re_sorted = v.re_sorted; q.rs_top = v.rs_top; q.re_top = v.re_top;
  q.lo = v.t_read_off[t]; q.hi = v.t_read_off[t + 1]; q.L = v.t_contig_len[t];
  q.nm_start = v.nm_start; q.nm_end = v.nm_end;
  q.nm_lo = v.t_nm_off ? v.t_nm_off[t] : 0; q.nm_hi = v.t_nm_off ? v.t_nm_off[t + 1] : 0;
  return q;
}
SNF_HD bool view_masked(const View& v, int t, int64_t idx) { return v.t_nm_off && v.t_nm_off[t] < v.t_nm_off[t + 1] && cov_masked(reads_of(v, t), idx); }
SNF_HD bool cov_get(const View& v, int t, int64_t idx, int32_t* out) {
  int64_t len = v.t_contig_len[t];
  if (idx < -len || idx >= len) return false;  // IndexError: the field keeps its value
  if (idx < 0) idx += len;                      // numpy negative index
  int64_t lo = v.t_read_off[t], hi = v.t_read_off[t + 1];
  int64_t ns = bound_top_i32<true>(v.r_start, v.rs_top, lo, hi, idx) - lo;
  int64_t ne = bound_top_i32<true>(v.re_sorted, v.re_top, lo, hi, idx) - lo;
  *out = view_masked(v, t, idx) ? 0 : (int32_t)((uint64_t)(ns - ne) & 0xffffu);
  return true;
}

// the five samples of a call lie within a few hundred bp of each other: the first one pays the two-level search, the
// others gallop from where the previous one ended (same cache lines of r_start / re_sorted)
SNF_HD bool cov_get_near(const View& v, int t, int64_t idx, int32_t* out, int64_t* hs, int64_t* he) {
  int64_t len = v.t_contig_len[t];
  if (idx < -len || idx >= len) return false;  // IndexError: the field keeps its value
  if (idx < 0) idx += len;                      // numpy negative index
  int64_t lo = v.t_read_off[t], hi = v.t_read_off[t + 1];
  const int64_t ps = *hs < 0 ? bound_top_i32<true>(v.r_start, v.rs_top, lo, hi, idx) : upper_bound_hint_i32(v.r_start, lo, hi, idx, *hs);
  const int64_t pe = *he < 0 ? bound_top_i32<true>(v.re_sorted, v.re_top, lo, hi, idx) : upper_bound_hint_i32(v.re_sorted, lo, hi, idx, *he);
  *hs = ps; *he = pe;
  *out = view_masked(v, t, idx) ? 0 : (int32_t)((uint64_t)(ps - pe) & 0xffffu);
  return true;
}

// d4s_coverage: a thread per (call, sample) - 5 x 2 independent rank queries per call run side by side instead of one thread walking
// them in turn (a pass has ~94 k calls: 1.4 waves per SIMD with a thread per call, nothing to hide the chain behind) - and every query
// a 16-ary descent (snf_exact.h::rank_upper_16ary).  Same five samples, same quirks (postprocessing.py:69-130): numpy's negative
// index, an out-of-range sample keeps the field's value, a BND takes `end` from the last non-BND call before it.
SNF_HD void d4s_sample_body(int64_t q, const View& v) {
  const int64_t i = q / 5; const int k = (int)(q - 5 * i);
  snf_call_t& c = v.calls[i];
  const int t = c.task_index;
  if (v.t_status[t] != SNF_TASK_OK) return;
  const int64_t bs = v.cfg.coverage_binsize, ud = v.cfg.coverage_updown_bins;
  const int svtype = c.svtype;
  int64_t start = c.pos, end;
  if (svtype == SNF_INS) end = start + 1;
  else if (svtype == SNF_BND) { if (c.bnd_is_first) start -= 1; end = v.t_stale_end[t]; }
  else end = (int64_t)c.pos + iabs64(c.svlen);
  const bool point = svtype == SNF_INS || svtype == SNF_BND;
  int64_t idx;
  switch (k) {
    case 0: idx = start - bs * ud; break;
    case 1: idx = point ? start - bs : start; break;
    case 2: idx = point ? start : (start + end) / 2; break;
    case 3: idx = point ? end + bs : end - bs; break;
    default: idx = end + bs * ud; break;
  }
  const int64_t len = v.t_contig_len[t];
  if (idx < -len || idx >= len) return;        // IndexError: the field keeps its value
  if (idx < 0) idx += len;                      // numpy negative index
  const int64_t lo = v.t_read_off[t], hi = v.t_read_off[t + 1];
  const int64_t ps = rank_upper_16ary(v.r_start, v.rs_mid, v.rs_top, lo, hi, idx);
  const int64_t pe = rank_upper_16ary(v.re_sorted, v.re_mid, v.re_top, lo, hi, idx);
  c.cov[k] = view_masked(v, t, idx) ? 0 : (int32_t)((uint64_t)(ps - pe) & 0xffffu);
}

SNF_HD void d4_coverage_body(int64_t i, const View& v) {
  if (i >= v.cnt->n_calls) return;
  snf_call_t& c = v.calls[i];
  int t = c.task_index;
  if (v.t_status[t] != SNF_TASK_OK) return;
  int bs = v.cfg.coverage_binsize, ud = v.cfg.coverage_updown_bins;
  int64_t start = c.pos, end;
  if (c.svtype == SNF_INS) end = start + 1;
  else if (c.svtype == SNF_BND) { if (c.bnd_is_first) start -= 1; end = v.t_stale_end[t]; }
  else end = (int64_t)c.pos + iabs64(c.svlen);
  int64_t hs = -1, he = -1;
  int32_t cov[5] = {c.cov[0], c.cov[1], c.cov[2], c.cov[3], c.cov[4]};
  cov_get_near(v, t, start - (int64_t)bs * ud, &cov[0], &hs, &he);
  if (c.svtype == SNF_INS || c.svtype == SNF_BND) {
    cov_get_near(v, t, start - bs, &cov[1], &hs, &he);
    cov_get_near(v, t, start, &cov[2], &hs, &he);
    cov_get_near(v, t, end + bs, &cov[3], &hs, &he);
  } else {
    cov_get_near(v, t, start, &cov[1], &hs, &he);
    cov_get_near(v, t, (start + end) / 2, &cov[2], &hs, &he);
    cov_get_near(v, t, end - bs, &cov[3], &hs, &he);
  }
  cov_get_near(v, t, end + (int64_t)bs * ud, &cov[4], &hs, &he);
  for (int k = 0; k < 5; k++) c.cov[k] = cov[k];
}

// coverage.mean(): sum of clipped read lengths / contig_len (exact integer sum, one division)
#define SNF_COV_CHUNK 256
SNF_HD void d5_covsum_body(int64_t j, const View& v) {
  int64_t lo = j * SNF_COV_CHUNK, hi = lo + SNF_COV_CHUNK;
  if (hi > v.R) hi = v.R;
  if (lo >= hi) return;
  int t = v.r_task[lo];
  unsigned long long acc = 0;
  for (int64_t r = lo; r < hi; r++) {
    int tr = v.r_task[r];
    if (tr != t) { atomic_add_u64(&v.t_cov_sum[t], acc); acc = 0; t = tr; }
    if (v.t_cov_exact && v.t_cov_exact[tr]) continue;      // d5x_covexact forms this task's sum (mask / uint16 wrap)
    int64_t L = v.t_contig_len[tr];
    int64_t s = v.r_start[r], e = v.r_end[r];
    if (s < 0) s = 0; if (s > L) s = L;
    if (e < 0) e = 0; if (e > L) e = L;
    if (e > s) acc += (unsigned long long)(e - s);
  }
  atomic_add_u64(&v.t_cov_sum[t], acc);
}
// coverage.mean() of a task whose vector is masked or wraps: the exact sum of the masked uint16 vector, 4096 positions per thread
#define SNF_COVX_CHUNK 4096
struct CovExact { Reads q; unsigned long long* out; };
SNF_HD void d5x_covexact_body(int64_t j, const CovExact& p) {
  const int64_t a = j * SNF_COVX_CHUNK;
  if (a >= p.q.L) return;
  const int64_t b = a + SNF_COVX_CHUNK < p.q.L ? a + SNF_COVX_CHUNK : p.q.L;
  const uint64_t s = cov_range_sum(p.q, a, b);
  if (s) atomic_add_u64(p.out, (unsigned long long)s);
}
// largest depth of a task = the depth right at some read's start: decides at upload whether uint16 can wrap at all
struct MaxDepth { const int32_t *r_start, *re_sorted, *rs_top, *re_top; const int32_t* r_task; const int64_t* t_read_off; int32_t* t_max_depth; };
SNF_HD void r3_maxdepth_body(int64_t r, const MaxDepth& p) {
  const int t = p.r_task[r];
  const int64_t lo = p.t_read_off[t], hi = p.t_read_off[t + 1];
  const int64_t x = p.r_start[r];
  if (r + 1 < hi && p.r_start[r + 1] == x) return;     // (the last read of equal starts sees them all)
  const int64_t ns = r + 1 - lo;
  const int64_t ne = bound_top_i32<true>(p.re_sorted, p.re_top, lo, hi, x) - lo;
  const int32_t d = (int32_t)(ns - ne);
#if defined(__HIP_DEVICE_COMPILE__)
  // (millions of reads of a task would queue on one address: only a depth above what is already recorded goes to the atomic)
  if (d > __atomic_load_n(&p.t_max_depth[t], __ATOMIC_RELAXED)) atomicMax(&p.t_max_depth[t], d);
#else
  if (d > p.t_max_depth[t]) p.t_max_depth[t] = d;
#endif
}
SNF_HD void d5_covavg_body(int64_t t, const View& v) {
  int64_t L = v.t_contig_len[t];
  v.t_cov_avg[t] = L > 0 ? (double)v.t_cov_sum[t] / (double)L : NAN;
}

// SNFile.annotate_block_coverages (snf.py:249-267): the dense coverage vector zero-padded to a multiple of `binsize`,
// averaged per bin (float64 mean of exact integers) and rounded with Python's round() (half to even).  The bin sum is
// formed from the sparse read table: sum over x in [a, b) of #(start <= x) - #(end <= x); the starts (ends) below a
// count for the whole bin, those inside it from their position to the bin's end.
struct BlockCov {
  const int32_t *r_start, *re_sorted, *rs_top, *re_top;
  int64_t lo, hi;      // the task's reads
  int64_t L;           // contig length
  const int32_t *nm_start, *nm_end; int64_t nm_lo, nm_hi; int32_t exact, _pad;   // mask intervals; exact: sum by cov_range_sum
  int64_t first_bin;   // bin j covers [j * binsize, (j + 1) * binsize)
  int32_t binsize;
  int32_t* out;        // rounded mean depth, -1: bin beyond the padded vector (IndexError in the reference, bin skipped)
};
SNF_HD int64_t blockcov_side(const int32_t* a, const int32_t* top, int64_t lo, int64_t hi, int64_t x0, int64_t x1) {
  const int64_t p0 = bound_top_i32<true>(a, top, lo, hi, x0 - 1), p1 = bound_top_i32<true>(a, top, lo, hi, x1 - 1);
  int64_t s = (p0 - lo) * (x1 - x0);
  for (int64_t p = p0; p < p1; p++) s += x1 - (int64_t)a[p];
  return s;
}
SNF_HD void s1_blockcov_body(int64_t i, const BlockCov& q) {
  const int64_t bs = q.binsize, x0 = (q.first_bin + i) * bs;
  if (x0 >= q.L) { q.out[i] = -1; return; }
  const int64_t x1 = x0 + bs < q.L ? x0 + bs : q.L;
  int64_t sum;
  if (q.exact) {
    Reads r; r.r_start = q.r_start; r.re_sorted = q.re_sorted; r.rs_top = q.rs_top; r.re_top = q.re_top; r.lo = q.lo; r.hi = q.hi; r.L = q.L;
    r.nm_start = q.nm_start; r.nm_end = q.nm_end; r.nm_lo = q.nm_lo; r.nm_hi = q.nm_hi;
    sum = (int64_t)cov_range_sum(r, x0, x1);
  } else sum = blockcov_side(q.r_start, q.rs_top, q.lo, q.hi, x0, x1) - blockcov_side(q.re_sorted, q.re_top, q.lo, q.hi, x0, x1);
  const int64_t quo = sum / bs, rem = sum % bs;   // sum >= 0: a read ends after it starts
  q.out[i] = (int32_t)(quo + ((2 * rem > bs || (2 * rem == bs && (quo & 1))) ? 1 : 0));
}

// postprocessing.coverage (postprocessing.py:69-130) for calls that are NOT this batch's own candidates - the target SVs of
// GenotypeTask.execute (parallel.py:353).  Same five samples and the same quirks as d4_coverage: numpy's negative
// index, an out-of-range sample keeps the field's value, and a BND takes `end` from the last non-BND call before it in
// list order (UnboundLocalError when there is none: the calls from there on stay untouched, status 1).
struct CovCalls {
  const int32_t *r_start, *re_sorted, *rs_top, *re_top;
  int64_t lo, hi, L, n;
  const int32_t *nm_start, *nm_end; int64_t nm_lo, nm_hi;   // mask intervals of the task
  int32_t binsize, updown;
  const int32_t *svtype, *pos, *svlen; const uint8_t* bnd_is_first;
  int64_t* end;        // [n] scratch: `end` as the loop of the reference sees it at call i
  int32_t* cov;        // [5n] in/out: upstream, start, center, end, downstream
  int32_t* n_valid;    // [2]: calls annotated, status
};
SNF_HD void s2_covends_body(int64_t, const CovCalls& q) {   // one thread: `end` is carried from call to call
  int64_t end = 0; bool have = false; int64_t i = 0;
  for (; i < q.n; i++) {
    if (q.svtype[i] == SNF_INS) { end = (int64_t)q.pos[i] + 1; have = true; }
    else if (q.svtype[i] == SNF_BND) { if (!have) break; }
    else { end = (int64_t)q.pos[i] + iabs64(q.svlen[i]); have = true; }
    q.end[i] = end;
  }
  q.n_valid[0] = (int32_t)i; q.n_valid[1] = i < q.n ? 1 : 0;
}
SNF_HD void covcalls_sample(const CovCalls& q, int64_t idx, int32_t* out) {
  if (idx < -q.L || idx >= q.L) return;
  if (idx < 0) idx += q.L;
  const int64_t ns = bound_top_i32<true>(q.r_start, q.rs_top, q.lo, q.hi, idx) - q.lo;
  const int64_t ne = bound_top_i32<true>(q.re_sorted, q.re_top, q.lo, q.hi, idx) - q.lo;
  Reads r; r.nm_start = q.nm_start; r.nm_end = q.nm_end; r.nm_lo = q.nm_lo; r.nm_hi = q.nm_hi;
  *out = cov_masked(r, idx) ? 0 : (int32_t)((uint64_t)(ns - ne) & 0xffffu);
}
SNF_HD void s2_covcalls_body(int64_t i, const CovCalls& q) {
  if (i >= q.n_valid[0]) return;
  const int t = q.svtype[i];
  int64_t start = q.pos[i]; const int64_t end = q.end[i];
  if (t == SNF_BND && q.bnd_is_first[i]) start -= 1;
  int32_t* c = q.cov + 5 * i;
  const int64_t bs = q.binsize;
  if (t == SNF_INS || t == SNF_BND) {
    covcalls_sample(q, start - bs, &c[1]); covcalls_sample(q, start, &c[2]); covcalls_sample(q, end + bs, &c[3]);
  } else {
    covcalls_sample(q, start, &c[1]); covcalls_sample(q, (start + end) / 2, &c[2]); covcalls_sample(q, end - bs, &c[3]);
  }
  covcalls_sample(q, start - bs * q.updown, &c[0]);
  covcalls_sample(q, end + bs * q.updown, &c[4]);
}

// Z1: counters, task status / call offsets / coverage averages -> the pinned host result block (zero-copy stores)
SNF_HD void z1_results_body(int64_t t, const View& v) {
  if (t < v.T) { v.res_status[t] = v.t_status[t]; v.res_cov[t] = v.t_cov_avg[t]; }
  // offsets of the tasks' records: candidate list, or (after finalize) the block of the output stage
  if (t <= v.T) v.res_off[t] = v.out_valid ? (int64_t)v.o_scan[v.t_call_off[t]] : v.t_call_off[t];
  if (t == 0 && v.out_valid) *v.res_out = *v.out_hdr;
  // striped byte counters of the ALT kernels (snf_wave_cons.h): class c summed by thread c % (T + 1), straight into the pinned copy
  // (the copy of the other counters below skips these four words)
  const size_t cb0 = offsetof(Counts, cons_bytes) / 8;
  if (v.wave_path) for (int c = (int)t; c < 4; c += (int)tail_threads(v)) {
    unsigned long long sum = 0;
    for (int k = 0; k < 64; k++) sum += v.stripes[(c * 64 + k) * 16];
    v.cnt->cons_bytes[c] = sum;
    ((unsigned long long*)v.res_cnt)[cb0 + c] = sum;
  }
  // how many clusters / calls went to the one-wave-each kernels x_big<kind>: a handle whose pass found none skips those launches next time
  const size_t nb0 = offsetof(Counts, n_big) / 8;
  if (v.wave_path) for (int c = (int)t - 4; c >= 0 && c < 3; c += (int)tail_threads(v)) {   // (threads 4..6)
    long long sum = 0;
    for (int k = 0; k < 64; k++) sum += v.big_cnt[(c * 64 + k) * 16];
    v.cnt->n_big[c] = sum;
    ((long long*)v.res_cnt)[nb0 + c] = sum;
  }
  const unsigned long long* s = (const unsigned long long*)v.cnt; unsigned long long* d = (unsigned long long*)v.res_cnt;
  for (size_t k = (size_t)t; k < sizeof(Counts) / 8; k += (size_t)tail_threads(v))
    if (!(v.wave_path && ((k >= cb0 && k < cb0 + 4) || (k >= nb0 && k < nb0 + 3)))) d[k] = s[k];
}

}  // namespace snf
