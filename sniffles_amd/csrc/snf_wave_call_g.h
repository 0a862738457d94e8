// snf_wave_call_g.h - sv.call_from / resolve_bnd (sv.py:497-639) for SMALL refined clusters: several clusters per wave.
//
// d2w_call (snf_wave_call.h) gives every refined cluster a wave of its own, one lead per lane.  On a 30x genome 71 % of the
// refined clusters have at most 4 leads and 85 % at most 16 (the noise candidates QC drops later; the calls that pass have
// 15-30): 7 of 64 lanes worked on average, and the per-cluster chain of dependent loads was paid 94 000 times per pass.
// Here a wave is cut into 64 / G groups of G lanes (G = 8, then 32) and every group takes a cluster of at most G leads: the
// same arithmetic, every wave-wide primitive of d2w_call replaced by its group-wide form (ballots sliced per group, shuffles
// inside the group, LDS scratch indexed from the group's base lane).  A cluster that does not fit a group is handed on:
// d2g_call<8> -> list 0 -> d2g_call<32> -> list 1 -> d2w_call -> (more than 64 leads) x_big<1>.  Results are indexed by the
// refined cluster's id, so the order in which the kernels get to a cluster does not matter.
//
// Control flow is wave-uniform throughout: a group whose cluster is screened out, or that has no cluster, keeps executing with
// its predicate off (the collectives of the other groups need every lane).
#pragma once
#include "snf_wave_call.h"

namespace snf {

#ifdef SNF_CONS_PROFILE
#define D2G_PT_DECL unsigned long long dpt = __builtin_amdgcn_s_memtime();
#define D2G_PT(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0 && G == 8) atomicAdd(&v.cnt->dbg[9 + (k)], t_ - dpt); dpt = t_; } while (0)
#else
#define D2G_PT_DECL
#define D2G_PT(k) do { } while (0)
#endif

template <int G> SNF_D unsigned long long gballot(bool p, int gbase) {
  const unsigned long long m = __ballot(p);
  if (G == 64) return m;
  return (m >> gbase) & ((1ull << G) - 1ull);
}
template <int G> SNF_D int32_t gshfl_i32(int32_t x, int src, int gbase) { return __shfl(x, gbase + src, SNF_WAVE); }
template <int G> SNF_D uint64_t gshfl_u64(uint64_t x, int src, int gbase) {
  const uint32_t lo = (uint32_t)__shfl((int32_t)(uint32_t)x, gbase + src, SNF_WAVE), hi = (uint32_t)__shfl((int32_t)(uint32_t)(x >> 32), gbase + src, SNF_WAVE);
  return ((uint64_t)hi << 32) | lo;
}
// group-wide all-reduce.  G = 8: three DPP steps (lanes ^1 and ^2 inside the quad, then the mirror image inside the half row:
// lane i <- lane 7 - i, which lies in the other quad) instead of three round trips through the LDS crossbar
#define SNF_DPP_G8(STEP) STEP(0xB1) STEP(0x4E) STEP(0x141)   /* quad_perm:[1,0,3,2], quad_perm:[2,3,0,1], row_half_mirror */
template <int G> SNF_D int64_t gsum64(int64_t x) {
  if constexpr (G == 8) {
#define SNF_STEP(ctrl) { const uint32_t lo_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(uint64_t)x, ctrl, 0xf, 0xf, false), \
                                        hi_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)((uint64_t)x >> 32), ctrl, 0xf, 0xf, false); \
                         x += (int64_t)(((uint64_t)hi_ << 32) | lo_); }
    SNF_DPP_G8(SNF_STEP)
#undef SNF_STEP
  } else {
#pragma unroll
    for (int d = G / 2; d >= 1; d >>= 1) x += __shfl_xor(x, d, SNF_WAVE);
  }
  return x;
}
template <int G> SNF_D uint32_t gsum32(uint32_t x) {      // sums below 2^32: one DPP add per step
  if constexpr (G == 8) {
#define SNF_STEP(ctrl) x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, ctrl, 0xf, 0xf, false);
    SNF_DPP_G8(SNF_STEP)
#undef SNF_STEP
  } else {
#pragma unroll
    for (int d = G / 2; d >= 1; d >>= 1) x += (uint32_t)__shfl_xor((int)x, d, SNF_WAVE);
  }
  return x;
}
template <int G> SNF_D int gmax32(int x) {
  if constexpr (G == 8) {
#define SNF_STEP(ctrl) { const int y_ = __builtin_amdgcn_update_dpp(x, x, ctrl, 0xf, 0xf, false); x = y_ > x ? y_ : x; }
    SNF_DPP_G8(SNF_STEP)
#undef SNF_STEP
  } else {
#pragma unroll
    for (int d = G / 2; d >= 1; d >>= 1) { const int y = __shfl_xor(x, d, SNF_WAVE); if (y > x) x = y; }
  }
  return x;
}
// rank of `key` among the group's keys (inactive lanes hold ~0: never smaller); nmax = the largest cluster of the wave
// (four lanes' keys are fetched per step: a ds_bpermute takes ~100 cycles to come back and the compares depend on nothing else)
template <int G> SNF_D int grank(uint64_t key, int gbase, int nmax) {
  int r = 0;
  for (int i = 0; i < nmax; i += 4) {
    uint64_t k[4];
#pragma unroll
    for (int u = 0; u < 4; u++) k[u] = gshfl_u64<G>(key, (i + u) & (G - 1), gbase);
#pragma unroll
    for (int u = 0; u < 4; u++) r += (i + u < nmax && k[u] < key) ? 1 : 0;
  }
  return r;
}
// ascending sort of the group's active values; returns the value at sorted position gl (lanes >= n: the first one)
template <int G> SNF_D int32_t gsort_i32(int32_t x, bool act, int n, int gl, int gbase, int nmax, int32_t* lds) {
  const uint64_t key = act ? (((uint64_t)((uint32_t)x ^ 0x80000000u) << 8) | (uint32_t)gl) : ~0ull;
  const int rank = grank<G>(key, gbase, nmax);
  __syncthreads();
  if (act) lds[gbase + rank] = x;
  __syncthreads();
  return lds[gbase + (gl < n ? gl : 0)];
}
// util.median_modes == center (util.py:49-58) on lanes 0..n-1 of the group holding a sorted array; n == 0: 0
template <int G> SNF_D int32_t gcenter_sorted(int32_t s, int n, int gl, int gbase) {
  const int32_t p = __shfl_up(s, 1, SNF_WAVE);
  const bool start = gl < n && (gl == 0 || p != s);
  const unsigned long long smask = gballot<G>(start, gbase);
  const unsigned long long above = gl < G - 1 ? (smask >> (gl + 1)) : 0ull;
  const int len = start ? ((above ? gl + 1 + __builtin_ctzll(above) : n) - gl) : 0;
  const int maxc = gmax32<G>(len);
  const bool q = start && (maxc - len < 3);
  const unsigned long long qmask = gballot<G>(q, gbase);
  const int k = __builtin_popcountll(qmask), want = k / 2;
  unsigned long long m = qmask;
  for (int i = 0; i < want; i++) m &= m - 1;
  const int src = m ? __builtin_ctzll(m) : 0;
  return gshfl_i32<G>(s, src, gbase);
}
// util.stdev(util.trim(sorted)) (util.py:25-27,82-88)
template <int G> SNF_D double gstdev_trim_sorted(int32_t s, int n, int gl, int gbase) {
  const int trim_n = (int)((double)n / 100.0 * 25.0);
  const int lo = trim_n > 0 ? trim_n : 0, hi = trim_n > 0 ? n - trim_n : n;
  const int cnt = hi - lo;
  const int64_t x0 = gshfl_i32<G>(s, cnt >= 2 ? lo : 0, gbase);
  const bool in = cnt >= 2 && gl >= lo && gl < hi;
  const uint64_t d = in ? (uint64_t)((int64_t)s - x0) : 0;  // sorted: d >= 0, < 2^32
  // every group of the wave within 8191 (the usual case, wave-uniform test): 32-bit sums and one fp64 division (see wave_stdev_trim_sorted)
  if (__ballot(d >= (1ull << 13)) == 0ull) {
    const uint32_t d32 = (uint32_t)d;
    const uint64_t S1s = gsum32<G>(d32), S2s = gsum32<G>(d32 * d32);
    if (cnt < 2) return 0.0;
    const uint64_t num = (uint64_t)cnt * S2s - S1s * S1s;
    return sqrt((double)num / (double)((uint64_t)cnt * (uint64_t)(cnt - 1)));
  }
  const uint64_t d2 = d * d;
  const int64_t S1 = gsum64<G>((int64_t)d);
  const int64_t lo32 = gsum64<G>((int64_t)(d2 & 0xffffffffull)), hi32 = gsum64<G>((int64_t)(d2 >> 32));
  if (cnt < 2) return 0.0;
  const u128 S2 = ((u128)(uint64_t)hi32 << 32) + (u128)(uint64_t)lo32;
  return stdev_from_sums(cnt, (i128)S1, S2);
}

// the lead aggregates of wave_lead_agg (snf_wave_call.h), per group
template <int G, bool PHASE>
SNF_D void group_lead_agg(const snf_config_t& cfg, CallLds& lds, int gl, int gbase, int n, int nmax, bool sel, int strand, int hap, uint32_t rid,
                          int32_t ps, bool close, bool want_nm, double nm, CallX& x) {
  x.ag_valid = 1;
  x.ag_nstrands = (gballot<G>(sel && strand == 0, gbase) ? 1 : 0) + (gballot<G>(sel && strand != 0, gbase) ? 1 : 0);
  x.ag_close_edge = __builtin_popcountll(gballot<G>(sel && close, gbase));
  int hp_val = 0, hp_support = -1, hp_other = 0; int32_t ps_val = 0; int ps_support = -1, ps_other = 0;
  if (PHASE) {
    bool later = false;   // reads_phases = {read_id: (hap, ps)}: the last lead of a read wins
    const unsigned long long selmask = gballot<G>(sel, gbase);
    for (int k0 = 0; k0 < nmax; k0 += 4) {
      int32_t rk[4];
#pragma unroll
      for (int u = 0; u < 4; u++) rk[u] = gshfl_i32<G>((int32_t)rid, (k0 + u) & (G - 1), gbase);
#pragma unroll
      for (int u = 0; u < 4; u++) if (k0 + u < n && k0 + u > gl && ((selmask >> ((k0 + u) & 63)) & 1ull) && rk[u] == (int32_t)rid) later = true;
    }
    const bool contrib = sel && !later;
    int hc[3];
    for (int hh = 0; hh < 3; hh++) hc[hh] = __builtin_popcountll(gballot<G>(contrib && hap == hh, gbase));
    for (int hh = 0; hh < 3; hh++) if (hc[hh] > 0 && hc[hh] >= hp_support) { hp_support = hc[hh]; hp_val = hh; }
    for (int hh = 0; hh < 3; hh++) if (hh != hp_val) hp_other += hc[hh];
    const int np_ = __builtin_popcountll(gballot<G>(contrib, gbase));
    const uint64_t key = contrib ? (((uint64_t)((uint32_t)ps ^ 0x80000000u) << 8) | (uint32_t)gl) : ~0ull;
    const int rank = grank<G>(key, gbase, nmax);
    __syncthreads();
    if (contrib) lds.buf[gbase + rank] = ps;
    __syncthreads();
    const int32_t s_ps = lds.buf[gbase + (gl < np_ ? gl : 0)];
    const int32_t p_ps = __shfl_up(s_ps, 1, SNF_WAVE);
    const bool st = gl < np_ && (gl == 0 || p_ps != s_ps);
    const unsigned long long smask = gballot<G>(st, gbase);
    const unsigned long long above = gl < G - 1 ? (smask >> (gl + 1)) : 0ull;
    const int len = st ? ((above ? gl + 1 + __builtin_ctzll(above) : np_) - gl) : 0;
    const int maxc = gmax32<G>(len);
    const unsigned long long best = gballot<G>(st && len == maxc, gbase);
    const int bl = best ? 63 - __builtin_clzll(best) : 0;   // (count, value) descending: ties -> larger value
    ps_val = gshfl_i32<G>(s_ps, bl, gbase); ps_support = maxc;
    ps_other = (int)gsum32<G>((st && s_ps != ps_val && s_ps != SNF_PS_NULL_CODE) ? (uint32_t)len : 0u);
    __syncthreads();
  }
  x.ag_hp_val = hp_val; x.ag_hp_support = hp_support; x.ag_hp_other = hp_other;
  x.ag_ps_val = ps_val; x.ag_ps_support = ps_support; x.ag_ps_other = ps_other;
  x.ag_has_nm = 0; x.ag_nm_mean = 0.0;
  if (PHASE && want_nm) {
    // np.nanmean with numpy's pairwise summation for n <= 128 (see wave_lead_agg)
    const int cnt = __builtin_popcountll(gballot<G>(gl < n && nm == nm, gbase));
    lds.nm[gbase + gl] = (gl < n && nm == nm) ? nm : 0.0;
    __syncthreads();
    double res = 0.0;
    const int n8 = n - n % 8;
    double r = 0.0;
    if (n >= 8 && gl < 8) { r = lds.nm[gbase + gl]; for (int q = 8 + gl; q < n8; q += 8) r += lds.nm[gbase + q]; }
    double rq[8];
    for (int q = 0; q < 8; q++) rq[q] = __longlong_as_double((long long)gshfl_u64<G>((uint64_t)__double_as_longlong(r), q, gbase));
    if (n < 8) { for (int q = 0; q < n; q++) res += lds.nm[gbase + q]; }
    else {
      res = ((rq[0] + rq[1]) + (rq[2] + rq[3])) + ((rq[4] + rq[5]) + (rq[6] + rq[7]));
      for (int q = n8; q < n; q++) res += lds.nm[gbase + q];
    }
    x.ag_nm_mean = res / (double)cnt; x.ag_has_nm = 1;
    __syncthreads();
  }
}

// LIST: the items are the entries of hand-over list 0 (clusters d2g_call<8> handed on); otherwise every refined cluster
template <int G, int MINW, bool PHASE>
__global__ void __launch_bounds__(SNF_WAVE, MINW) d2g_call(const View v, int64_t n_unused) {
  IT_SCOPE(6)
  static_assert(G == 8 || G == 32, "group widths in use");
  constexpr int NG = SNF_WAVE / G;
  constexpr bool LIST = G != 8;
  __shared__ CallLds lds;
  const int lane = threadIdx.x, gl = lane & (G - 1), gbase = lane - gl, gi = lane / G;
  const snf_config_t& cfg = v.cfg;
  __shared__ int32_t d2pre[65];
  const int64_t n_items = LIST ? d2list_prefix(v, 0, lane, d2pre) : v.cnt->n_rc;
  const bool want_nm = PHASE && cfg.mode_call_sample;
  // software pipeline over the wave's items, as in d2w_call: the table entries of item k + 2 and the cluster head / lead order
  // of item k + 1 are requested while item k is worked on (an item then starts two round trips deep instead of five)
  struct Item { int32_t r, flo, n, c, h, slot; bool valid; };
  const int64_t stride = (int64_t)gridDim.x * NG;
  auto level1 = [&](int64_t base_, Item& t) {
    const int64_t idx = base_ + gi;
    t.valid = idx < n_items; t.r = 0; t.flo = 0; t.n = 0; t.c = 0; t.h = 0; t.slot = 0;
    if (t.valid) { t.r = LIST ? d2list_at(v, 0, idx, d2pre) : (int32_t)idx; t.flo = v.rc_lo[t.r]; t.n = v.rc_n[t.r]; t.c = v.rc_cluster[t.r]; }
  };
  auto level2 = [&](Item& t) { if (t.valid && t.n <= G) { t.h = v.cl_head[t.c]; if (gl < t.n) t.slot = v.FI[t.flo + gl]; } };
  Item nx, nx2;
  level1((int64_t)blockIdx.x * NG, nx);
  level1((int64_t)blockIdx.x * NG + stride, nx2);
  level2(nx);
  for (int64_t base = (int64_t)blockIdx.x * NG; base < n_items; base += stride) {
    const Item cur = nx;
    nx = nx2;
    level2(nx);
    level1(base + 2 * stride, nx2);
    bool valid = cur.valid;
    const int32_t r = cur.r;
    int32_t flo = cur.flo, n = cur.n, c = cur.c, h = cur.h;
    // clusters that do not fit a group go to the next kernel's list
    d2list_push(v, LIST ? 1 : 0, valid && n > G && gl == 0, r, lane, n);
    if (valid && n > G) valid = false;
    if (!valid) n = 0;
    int nmax = n;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const int y = __shfl_xor(nmax, d, SNF_WAVE); if (y > nmax) nmax = y; }
    nmax = __builtin_amdgcn_readfirstlane(nmax);
    if (nmax == 0) continue;                               // (wave-uniform: no group of this wave has a cluster)
    D2G_PT_DECL
    // everything the item reads that depends only on its ids is requested here, up front: the kernel is a chain of dependent
    // loads per item, and the sorts below then run while the second level is still in flight
    int g = 0; int32_t llo = 0, lhi = 0, c_last_h = 0, seed_start_h = 0, c_end_h = 0, seed_bin_h = 0, grp_first = 0; bool keeplong_rc = false;
    if (valid) {
      g = v.seed_grp[h]; llo = v.seedL_lo[h]; c_last_h = v.c_last[h]; seed_start_h = v.seed_start[h]; c_end_h = v.c_end[h];
      seed_bin_h = v.seed_bin[h]; keeplong_rc = v.rc_keeplong[r] != 0;
    }
    const int svtype = grp_svtype(g), task = grp_task(g);
    const bool act = gl < n;
    int32_t slot = 0; uint32_t o = 0; int32_t svl = 0, rs = 0; uint32_t qn = 0;
    int mapq = 0, strand = 0, is_sa = 0, noninline = 0; double nm = 0;
    int32_t mctg = 0, mpos = 0; int bfirst = 0, brev = 0;
    int hap = 0; uint32_t rid = 0; int32_t ps = SNF_PS_NULL_CODE; bool close = false;
    int32_t seq_len = -1; int64_t seq_off = 0;
    if (valid) { lhi = v.seedL_hi[c_last_h]; grp_first = v.grp_first_bin[g]; }
    if (act) {
      slot = cur.slot; svl = v.F_svlen[slot];
      seq_len = v.F_seq_len[slot]; seq_off = v.F_seq_off[slot];
      const LeadRec rr = v.Lrec[v.F_lpos[slot]];
      o = rr.orig; rs = rr.ref_start; qn = rr.qname; mapq = rr.mapq; strand = rr.strand; is_sa = rr.is_sa;
      noninline = rr.source != SNF_SRC_INLINE;
      if (PHASE) {
        hap = rr.hap; rid = rr.read_id;
        ps = (rr.ps == SNF_PS_NONE || rr.ps == v.t_ps_null[task]) ? SNF_PS_NULL_CODE : rr.ps;
      }
      close = (int64_t)rr.qry_start <= cfg.dev_min_close_edge_dist || iabs64((int64_t)rr.read_len - (int64_t)rr.qry_start) <= cfg.dev_min_close_edge_dist;
      if (cfg.qc_nm_measure || want_nm) nm = v.in_nm[o];
      mctg = rr.mate_contig; mpos = rr.mate_pos; bfirst = rr.first; brev = rr.rev;
      v.F_sel[slot] = 1;
    }
    if (valid && gl == 0) v.cdflag[r] = 0;
    const int32_t s_svl = gsort_i32<G>(svl, act, n, gl, gbase, nmax, lds.buf);
    const int64_t svlen = gcenter_sorted<G>(s_svl, n, gl, gbase);
    D2G_PT(0);
    const bool single = svtype == SNF_SINGLE_LEFT || svtype == SNF_SINGLE_RIGHT;
    bool alive = valid && !(!single && svtype != SNF_BND && iabs64(svlen) < cfg.minsvlen_screen);
    // distinct read names, sorted (kept in w1 for d3_rnames)
    const int32_t s_qn = gsort_i32<G>((int32_t)qn, act, n, gl, gbase, nmax, lds.buf);
    const int32_t p_qn = __shfl_up(s_qn, 1, SNF_WAVE);
    const bool qfirst = act && (gl == 0 || p_qn != s_qn);
    const unsigned long long qmask = gballot<G>(qfirst, gbase);
    int64_t nq = __builtin_popcountll(qmask);
    int32_t* a1 = v.w1 + flo;
    if (alive && qfirst) a1[__builtin_popcountll(qmask & ((1ull << gl) - 1ull))] = s_qn;
    const bool insl = alive && svtype == SNF_INS && svlen >= cfg.long_ins_length;
    if (__ballot(insl)) __syncthreads();     // (only the long-INS walk below reads a1 back: the stores need not be waited for otherwise)
    D2G_PT(1);
    int64_t support = nq, support_long = 0;
    const bool keeplong = alive && keeplong_rc && svtype == SNF_INS;
    // long INS (sv.py:578-590): distinct reads of the whole seed cluster's lead list - a walk that is quadratic in the list; the
    // WHOLE wave takes it for one group after the other (a group's few lanes would hold the wave up for its length)
    for (unsigned long long need = __ballot(insl && gl == 0); need; need &= need - 1ull) {
      const int src = __builtin_ctzll(need);
      const int32_t s_llo = __shfl(llo, src, SNF_WAVE), s_lhi = __shfl(lhi, src, SNF_WAVE), s_flo = __shfl(flo, src, SNF_WAVE);
      const int64_t s_nq = __shfl((int32_t)nq, src, SNF_WAVE);
      const int32_t* s_a1 = v.w1 + s_flo;
      int cl = 0, cu = 0;
      for (int32_t x = s_llo + lane; x < s_lhi; x += SNF_WAVE) {
        const int32_t q = (int32_t)v.in_qname[v.LL[x]];
        bool first = true;
        for (int32_t y = s_llo; y < x; y++) if ((int32_t)v.in_qname[v.LL[y]] == q) { first = false; break; }
        if (first) { cl++; if (!contains_sorted_i32(s_a1, s_nq, q)) cu++; }
      }
      const int64_t cls = wave_sum64(cl), cus = wave_sum64(cu);
      if (gbase == (src & ~(G - 1))) { support_long = cls; support += cus; }
    }
    const int32_t s_rs = gsort_i32<G>(rs, act, n, gl, gbase, nmax, lds.buf);
    const int64_t ref_start = gcenter_sorted<G>(s_rs, n, gl, gbase);
    const double stdev_pos = gstdev_trim_sorted<G>(s_rs, n, gl, gbase);
    const double stdev_len_all = gstdev_trim_sorted<G>(s_svl, n, gl, gbase);
    D2G_PT(2);
    double stdev_len = NAN; bool precise;
    if (svtype != SNF_BND) { stdev_len = stdev_len_all; precise = (stdev_pos + stdev_len < (double)cfg.precise); }
    else precise = stdev_pos < (double)cfg.precise;
    int64_t svstart, svend;
    if (svtype == SNF_INS) { svstart = ref_start; svend = ref_start; }
    else if (svtype == SNF_DEL) { svstart = ref_start + svlen; svend = ref_start; }
    else { svstart = ref_start; svend = svstart + iabs64(svlen); }
    const int64_t msum = gsum32<G>(act ? (uint32_t)mapq : 0u);      // (<= 64 x 255)
    const int64_t fwd = __builtin_popcountll(gballot<G>(act && strand == 0, gbase));
    int64_t sa = __builtin_popcountll(gballot<G>(act && is_sa, gbase));
    const int64_t src_noninline = __builtin_popcountll(gballot<G>(act && noninline, gbase));
    double nmsum = 0;
    if (cfg.qc_nm_measure) {  // Python sum(): left to right
      for (int i = 0; i < nmax; i++) {
        const uint64_t bits = gshfl_u64<G>((uint64_t)__double_as_longlong(nm), i, gbase);
        if (i < n) nmsum += __longlong_as_double((long long)bits);
      }
    }
    int64_t n_all = n;
    {
      int cs = 0;
      if (keeplong) for (int32_t x = llo + gl; x < lhi; x += G) cs += v.in_is_sa[v.LL[x]];
      const int64_t cst = gsum32<G>((uint32_t)cs);
      if (keeplong) { sa += cst; n_all += lhi - llo; }
    }
    snf_call_t cc;
    memset(&cc, 0, sizeof(cc));
    cc.task_index = task; cc.svtype = svtype; cc.pos = (int32_t)svstart; cc.end = (int32_t)svend; cc.svlen = (int32_t)svlen;
    cc.support = (int32_t)support; cc.support_long = -1; cc.support_sa = -1;
    cc.qual = (int32_t)((double)msum / (double)(n > 0 ? n : 1)); cc.precise = precise; cc.fwd = (int32_t)fwd; cc.rev = (int32_t)(n - fwd);
    cc.qc = 1; cc.filter = SNF_F_PASS;
    cc.nm = cfg.qc_nm_measure ? nmsum / (double)(n > 0 ? n : 1) : -1.0;
    cc.stdev_pos = stdev_pos; cc.stdev_len = stdev_len;
    cc.sa_count = (int32_t)sa; cc.sa_frac = (double)sa / (double)(n_all > 0 ? n_all : 1); cc.n_leads = n;
    cc.mate_contig = -1; cc.gt_hp = -1; cc.gt_ps = -1; cc.vaf = NAN; cc.alt_len = -1;
    if (valid) {
      cc.cluster_start = seed_start_h; cc.cluster_end = c_end_h;
      cc.cluster_seed_index = v.prefilter ? -1 : seed_bin_h - grp_first;
    }
    int64_t rn_len = support;
    bool sel_final = act;     // the leads the call keeps (resolve_bnd narrows them)
    D2G_PT(3);
    if (__ballot(alive && svtype == SNF_BND)) {  // resolve_bnd (sv.py:625-639), for the groups that hold a BND cluster
      const bool isb = alive && svtype == SNF_BND;
      const bool actb = act && isb;
      const int nb = isb ? n : 0;
      const int32_t s_mc = gsort_i32<G>(mctg, actb, nb, gl, gbase, nmax, lds.buf);
      const int32_t p_mc = __shfl_up(s_mc, 1, SNF_WAVE);
      const bool st = actb && (gl == 0 || p_mc != s_mc);
      const unsigned long long smask = gballot<G>(st, gbase);
      const unsigned long long above = gl < G - 1 ? (smask >> (gl + 1)) : 0ull;
      const int len = st ? ((above ? gl + 1 + __builtin_ctzll(above) : nb) - gl) : 0;
      const int maxc = gmax32<G>(len);
      const unsigned long long best = gballot<G>(st && len == maxc, gbase);
      const int32_t mc = gshfl_i32<G>(s_mc, best ? __builtin_ctzll(best) : 0, gbase);  // most_common_top: ties -> smallest value
      const bool sel = actb && mctg == mc;
      if (isb) sel_final = sel;
      if (actb) v.F_sel[slot] = sel ? 1 : 0;
      const int ns = __builtin_popcountll(gballot<G>(sel, gbase));
      const int64_t nfirst = __builtin_popcountll(gballot<G>(sel && bfirst, gbase)), nrev = __builtin_popcountll(gballot<G>(sel && brev, gbase));
      // selected values packed to the front by sorting with +inf for the rest
      const int32_t s_mp = gsort_i32<G>(sel ? mpos : INT32_MAX, actb, nb, gl, gbase, nmax, lds.buf);
      const int64_t mate_pos = gcenter_sorted<G>(s_mp, ns, gl, gbase);
      const int32_t s_q2 = gsort_i32<G>(sel ? (int32_t)qn : INT32_MAX, actb, nb, gl, gbase, nmax, lds.buf);
      const int32_t p_q2 = __shfl_up(s_q2, 1, SNF_WAVE);
      const bool qf2 = gl < ns && (gl == 0 || p_q2 != s_q2);
      const unsigned long long qm2 = gballot<G>(qf2, gbase);
      if (isb) {
        // SUPPORT: reads of the selected leads; RNAMES (a1, nq, rn_len) stays the read set of the whole cluster (sv.py:555)
        cc.support = (int32_t)__builtin_popcountll(qm2);
        cc.mate_contig = mc; cc.mate_ref_start = (int32_t)mate_pos;
        cc.bnd_is_first = (nfirst > ns - nfirst) ? 1 : 0;   // most_common_top: ties -> False
        cc.bnd_is_reverse = (nrev > ns - nrev) ? 1 : 0;
        cc.n_leads = ns;
      }
    }
    if (svtype == SNF_INS) cc.support_long = (int32_t)support_long;
    else if (svtype == SNF_DEL) cc.support_sa = (int32_t)src_noninline;
    cc.rn_len = (int32_t)rn_len;
    cc.rn_off = nq;  // stash: number of distinct names already sorted in w1[flo..]
    D2G_PT(4);
    int32_t best_slot = -1, n_others = 0;
    if (__ballot(alive && svtype == SNF_INS && !cfg.symbolic)) {
      // best lead of annotate_sv (postprocessing.py:33-66): first argmin of |len(seq) - svlen| + |ref_start - pos| * 1.5
      const bool isi = alive && svtype == SNF_INS && !cfg.symbolic;
      const int32_t sl = (act && isi) ? seq_len : -1;
      const bool has = act && isi && sl >= 0;
      const double d = (double)iabs64((int64_t)sl - svlen) + (double)iabs64((int64_t)rs - svstart) * 1.5;
      const unsigned long long key = has ? (unsigned long long)__double_as_longlong(d) : ~0ull;
      unsigned long long mn = key;
#pragma unroll
      for (int dd = G / 2; dd >= 1; dd >>= 1) { const unsigned long long o2 = __shfl_xor(mn, dd, SNF_WAVE); if (o2 < mn) mn = o2; }
      const unsigned long long hm = gballot<G>(has, gbase);
      const unsigned long long bm = gballot<G>(has && key == mn, gbase);
      const int bl = bm ? __builtin_ctzll(bm) : 0;
      const int32_t bs = gshfl_i32<G>(slot, bl, gbase);
      const bool oth = has && gl != bl;
      const unsigned long long om = gballot<G>(oth, gbase);
      if (hm) {
        best_slot = bs;
        n_others = __builtin_popcountll(hm) - 1;
        // read list of the consensus kernel: the other sequence-bearing leads in cluster order, in the cluster's own slot range
        if (oth) {
          const int w = __builtin_popcountll(om & ((1ull << gl) - 1ull));
          v.crl_off[flo + w] = seq_off; v.crl_len[flo + w] = sl;
        }
      }
    }
    D2G_PT(5);
    CallX x; x.rc = r; x.cluster = c; x.flo = flo; x.fn = n; x.best = best_slot; x.n_others = n_others;
    x.do_cons = (best_slot >= 0 && n_others >= cfg.consensus_min_reads && !cfg.no_consensus) ? 1 : 0; x.cons_id = -1; x.alt_off = 0;
    x.rn_nq = (int32_t)nq; x._pad = 0;
    group_lead_agg<G, PHASE>(cfg, lds, gl, gbase, n, nmax, sel_final, strand, hap, rid, ps, close, want_nm, nm, x);
    D2G_PT(6);
    if (alive && gl == 0) {
      v.cand[r] = cc;
      v.candx[r] = x;
      v.cdflag[r] = 1;
    }
    __syncthreads();
  }
}

}  // namespace snf
