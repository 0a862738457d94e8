// snf_wave_call.h - gfx950 wave-per-cluster implementations of sv.call_from / resolve_bnd (sv.py:497-639) and of
// the lead aggregates Task.finalize_candidates needs (strand set, phase majorities; postprocessing.py:626-654).
// One refined cluster (<= 64 leads) per wave, one lead per lane; sorts are rank sorts in registers, medians /
// modes / trimmed variances are ballots and butterfly reductions over the sorted lanes.  Larger clusters use the
// thread-per-cluster bodies (d2_call_body / e1_finalize_body), which are also what the host emulation runs.
#pragma once
#include "snf_wave_refine.h"

namespace snf {

struct CallLds { int32_t buf[SNF_WAVE]; double nm[SNF_WAVE]; };

SNF_D int64_t wave_sum64(int64_t x) { return wave_last64(wave_incl_scan64(x, 0)); }     // (DPP scans, snf_wave_refine.h)
// the same for sums that stay below 2^32 (counts, MAPQ sums, the small-spread variance sums): six DPP adds instead of six 64-bit steps
// of two moves and an add-with-carry each.  The call kernels are bound by instruction issue (d2w_call: ~2 850 instructions per cluster
// at five waves per SIMD is its 21 us per cluster), so what they do is only worth what it costs in instructions.
SNF_D uint32_t wave_sum32(uint32_t x) { return (uint32_t)__builtin_amdgcn_readlane(wave_incl_scan((int32_t)x, 0), 63); }
SNF_D int wave_max32(int x) { return (int)__builtin_amdgcn_readlane((uint32_t)wave_incl_max(x), 63); }

// ascending sort of the active lanes' values; returns the value at sorted position `lane` (lanes >= n: garbage)
SNF_D int32_t wave_sort_i32(int32_t x, bool act, int n, int lane, int32_t* lds) {
  const uint64_t key = act ? (((uint64_t)((uint32_t)x ^ 0x80000000u) << 8) | (uint32_t)lane) : ~0ull;
  const int rank = wave_rank(key, n);
  __syncthreads();
  if (act) lds[rank] = x;
  __syncthreads();
  return lds[lane < n ? lane : 0];
}

// util.median_modes == center on lanes 0..n-1 holding a sorted array (util.py:49-58)
SNF_D int32_t wave_center_sorted(int32_t s, int n, int lane) {
  const int32_t p = __shfl_up(s, 1, SNF_WAVE);
  const bool start = lane < n && (lane == 0 || p != s);
  const unsigned long long smask = __ballot(start);
  const unsigned long long above = lane < 63 ? (smask >> (lane + 1)) : 0ull;
  const int len = start ? ((above ? lane + 1 + __builtin_ctzll(above) : n) - lane) : 0;
  const int maxc = wave_max32(len);
  const bool q = start && (maxc - len < 3);
  const unsigned long long qmask = __ballot(q);
  const int k = __builtin_popcountll(qmask), want = k / 2;
  // the want-th set bit of qmask
  unsigned long long m = qmask;
  for (int i = 0; i < want; i++) m &= m - 1;
  const int src = __builtin_ctzll(m);
  return __shfl(s, src, SNF_WAVE);
}

// util.stdev(util.trim(sorted)) (util.py:25-27,82-88): exact 128-bit variance, one division, sqrt
SNF_D double wave_stdev_trim_sorted(int32_t s, int n, int lane) {
  const int trim_n = (int)((double)n / 100.0 * 25.0);
  const int lo = trim_n > 0 ? trim_n : 0, hi = trim_n > 0 ? n - trim_n : n;
  const int cnt = hi - lo;
  if (cnt < 2) return 0.0;
  const int64_t x0 = __shfl(s, lo, SNF_WAVE);
  const bool in = lane >= lo && lane < hi;
  const uint64_t d = in ? (uint64_t)((int64_t)s - x0) : 0;  // sorted: d >= 0, < 2^32
  // the usual cluster: the values lie within 8191 of each other (sorted: the last one is the largest).  Then sum(d) < 2^19 and
  // sum(d^2) < 2^32 - two 32-bit sums -, n * S2 - S1^2 < 2^38, and the exact ratio is ONE fp64 division of two exactly
  // representable integers: what stdev_from_sums / ratio_to_double return on their first branch, without the 128-bit arithmetic
  // in front of it
  const uint32_t dmax = (uint32_t)(__builtin_amdgcn_readlane(s, __builtin_amdgcn_readfirstlane(hi - 1)) - (int32_t)x0);
  if (dmax < (1u << 13)) {
    const uint32_t d32 = (uint32_t)d;
    const uint64_t S1s = wave_sum32(d32), S2s = wave_sum32(d32 * d32);
    const uint64_t num = (uint64_t)cnt * S2s - S1s * S1s;
    return sqrt((double)num / (double)((uint64_t)cnt * (uint64_t)(cnt - 1)));
  }
  const uint64_t d2 = d * d;
  const int64_t S1 = wave_sum64((int64_t)d);
  const int64_t lo32 = wave_sum64((int64_t)(d2 & 0xffffffffull)), hi32 = wave_sum64((int64_t)(d2 >> 32));
  const u128 S2 = ((u128)(uint64_t)hi32 << 32) + (u128)(uint64_t)lo32;
  return stdev_from_sums(cnt, (i128)S1, S2);
}

// ------------------------------------------------------------------------------------------ striped hand-over lists (View::d2_list)
// consumer side: prefix of the 64 stripes' counts in LDS (pre[0..64]); item i of the flat index space is entry i - pre[s] of stripe s
SNF_D int64_t d2list_prefix(const View& v, int k, int lane, int32_t* pre) {
  int32_t cnt = (int32_t)v.d2cnt[(k * 64 + lane) * 16];
  if (cnt > (int32_t)v.d2cap) cnt = (int32_t)v.d2cap;       // (overflowed stripe: flagged by the producer, the fetch fails)
  const int32_t inc = wave_incl_scan(cnt, lane);
  if (lane == 0) pre[0] = 0;
  pre[lane + 1] = inc;
  __syncthreads();
  return pre[64];
}
SNF_D int32_t d2list_at(const View& v, int k, int64_t i, const int32_t* pre) {
  int lo = 0, hi = 63;   // last stripe s with pre[s] <= i
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (pre[mid] <= (int32_t)i) lo = mid; else hi = mid - 1; }
  return v.d2_list[k][(int64_t)lo * v.d2cap + (i - pre[lo])];
}
// producer side: the lanes with `hand` append their cluster to list k, one atomic per wave on the workgroup's stripe.
// `nleads`: the item's size.  The kernel that walks the list starts its workgroups in index order and takes a few items each: a
// cluster of 60 leads costs several times what one of 10 does, and when such items sit anywhere in the list the kernel ends with a
// handful of waves working through them (d1w_refine: 4096 waves in flight for 40 us, fewer than 30 for the following 45 us).  Items
// above View::heavy_n go to stripes 0..15 - the front of the consumer's index space -, the others to stripes 16..63.
SNF_D void d2list_push(const View& v, int k, bool hand, int32_t r, int lane, int nleads = 0) {
  const bool split = v.heavy_n > 0;
  for (int cls = 0; cls < 2; cls++) {
    const bool mine = hand && (!split ? cls == 0 : (nleads > v.heavy_n) == (cls == 0));
    const unsigned long long hm = __ballot(mine);
    if (!hm) continue;
    const int stripe = !split ? (int)(blockIdx.x & 63) : cls == 0 ? (int)(blockIdx.x & 15) : 16 + (int)(blockIdx.x % 48u);
    const int leader = __builtin_ctzll(hm);
    uint32_t at = 0;
    if (lane == leader) at = atomicAdd(&v.d2cnt[(k * 64 + stripe) * 16], (uint32_t)__builtin_popcountll(hm));
    at = (uint32_t)__builtin_amdgcn_readlane((int)at, leader);
    if (mine) {
      const int64_t slot = (int64_t)at + __builtin_popcountll(hm & ((1ull << lane) - 1ull));
      if (slot < v.d2cap) v.d2_list[k][(int64_t)stripe * v.d2cap + slot] = r;
      else atomicOr(&v.cnt->overflow, 2);      // (a stripe holds a 64th of all LEADS + 64 entries; the items of either class are at least 9 times fewer)
    }
  }
}

// ------------------------------------------------------------------------------------------ lead aggregates of a call
// len(set(strands)), leads close to a read edge (postprocessing.py:574-577), the HP / PS majorities of phase_sv over distinct
// reads (the last lead of a read wins, postprocessing.py:626-654), np.nanmean of the NM ratios (rescue_phasing) - over the
// SELECTED leads of a cluster of n <= 64 leads, one lead per lane.  All lanes must call it.
template <bool PHASE>
SNF_D void wave_lead_agg(const snf_config_t& cfg, CallLds& lds, int lane, int n, bool sel, int strand, int hap, uint32_t rid, int32_t ps,
                         bool close, bool want_nm, double nm, CallX& x) {
  x.ag_valid = 1;
  x.ag_nstrands = (__ballot(sel && strand == 0) ? 1 : 0) + (__ballot(sel && strand != 0) ? 1 : 0);
  x.ag_close_edge = __builtin_popcountll(__ballot(sel && close));
  int hp_val = 0, hp_support = -1, hp_other = 0; int32_t ps_val = 0; int ps_support = -1, ps_other = 0;
  if (PHASE) {
    // reads_phases = {read_id: (hap, ps)}: the last lead of a read wins
    bool later = false;
    for (int k = 0; k < n; k++) {
      const uint32_t rk = (uint32_t)wave_bcast_i32((int32_t)rid, k);
      const bool sk = wave_bcast_i32(sel ? 1 : 0, k) != 0;
      if (k > lane && sk && rk == rid) later = true;
    }
    const bool contrib = sel && !later;
    int hc[3];
    for (int hh = 0; hh < 3; hh++) hc[hh] = __builtin_popcountll(__ballot(contrib && hap == hh));
    for (int hh = 0; hh < 3; hh++) if (hc[hh] > 0 && hc[hh] >= hp_support) { hp_support = hc[hh]; hp_val = hh; }
    for (int hh = 0; hh < 3; hh++) if (hh != hp_val) hp_other += hc[hh];
    const int np_ = __builtin_popcountll(__ballot(contrib));
    // phase sets of the contributing reads, sorted (non-contributors sort behind: rank sort on a wider key)
    const uint64_t key = contrib ? (((uint64_t)((uint32_t)ps ^ 0x80000000u) << 8) | (uint32_t)lane) : ~0ull;
    const int rank = wave_rank(key, n);
    __syncthreads();
    if (contrib) lds.buf[rank] = ps;
    __syncthreads();
    const int32_t s_ps = lds.buf[lane < np_ ? lane : 0];
    const int32_t p_ps = __shfl_up(s_ps, 1, SNF_WAVE);
    const bool st = lane < np_ && (lane == 0 || p_ps != s_ps);
    const unsigned long long smask = __ballot(st);
    const unsigned long long above = lane < 63 ? (smask >> (lane + 1)) : 0ull;
    const int len = st ? ((above ? lane + 1 + __builtin_ctzll(above) : np_) - lane) : 0;
    const int maxc = wave_max32(len);
    const unsigned long long best = __ballot(st && len == maxc);
    const int bl = 63 - __builtin_clzll(best);   // (count, value) descending: ties -> larger value
    ps_val = __shfl(s_ps, bl, SNF_WAVE); ps_support = maxc;
    ps_other = (int)wave_sum32((st && s_ps != ps_val && s_ps != SNF_PS_NULL_CODE) ? (uint32_t)len : 0u);
    __syncthreads();
  }
  x.ag_hp_val = hp_val; x.ag_hp_support = hp_support; x.ag_hp_other = hp_other;
  x.ag_ps_val = ps_val; x.ag_ps_support = ps_support; x.ag_ps_other = ps_other;
  x.ag_has_nm = 0; x.ag_nm_mean = 0.0;
  if (PHASE && want_nm) {
    // np.nanmean(nm of the leads, list order) with numpy's pairwise summation for n <= 128 (snf_exact.h::np_pairwise_sum):
    // fewer than 8 values: left to right; otherwise eight accumulators r[q] over the positions q, q + 8, ... below
    // n - n % 8, combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the tail left to right.  NaN counts as 0 in the sum.
    const int cnt = __builtin_popcountll(__ballot(lane < n && nm == nm));
    lds.nm[lane] = (lane < n && nm == nm) ? nm : 0.0;
    __syncthreads();
    double res = 0.0;
    if (n < 8) { for (int q = 0; q < n; q++) res += lds.nm[q]; }
    else {
      const int n8 = n - n % 8;
      double r = 0.0;
      if (lane < 8) { r = lds.nm[lane]; for (int q = 8 + lane; q < n8; q += 8) r += lds.nm[q]; }
      double rq[8];
      for (int q = 0; q < 8; q++) rq[q] = __shfl(r, q, SNF_WAVE);
      res = ((rq[0] + rq[1]) + (rq[2] + rq[3])) + ((rq[4] + rq[5]) + (rq[6] + rq[7]));
      for (int q = n8; q < n; q++) res += lds.nm[q];
    }
    x.ag_nm_mean = res / (double)cnt; x.ag_has_nm = 1;
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------ d2w: call_from
// PHASE: config.phase (a launch-time constant: the phase majorities and the NM mean are only formed, and their inputs only
// loaded, by the instance that needs them - the other one must not pay registers for them)
template <int MINW, bool PHASE>
__global__ void __launch_bounds__(SNF_WAVE, MINW) d2w_call(const View v, int64_t n_unused) {
  IT_SCOPE(7)
  __shared__ CallLds lds;
  const int lane = threadIdx.x;
  const snf_config_t& cfg = v.cfg;
  // the refined clusters of this launch: all of them, or (View::d2_from_list) the ones the grouped kernels handed on
  // (more than 32 leads, snf_wave_call_g.h)
  __shared__ int32_t d2pre[65];
  const bool from_list = v.d2_from_list != 0;      // 1 + the list's number
  const int64_t n_rc = from_list ? d2list_prefix(v, v.d2_from_list - 1, lane, d2pre) : v.cnt->n_rc;
  auto RC = [&](int64_t it) -> int32_t { return from_list ? d2list_at(v, v.d2_from_list - 1, it, d2pre) : (int32_t)it; };
  // software pipeline over this wave's refined clusters (the kernel is a chain of dependent loads: table entry -> cluster head
  // -> group, lead order -> fused lead -> record): the table entry of item k+2, the cluster head and the lead order of item
  // k+1 are requested while item k is worked on, so an item starts two round trips deep instead of six
  const int64_t stride = gridDim.x;
  int32_t flo1 = 0, n1 = 0, c1 = 0, flo2 = 0, n2 = 0, c2 = 0, h1 = 0, slot1 = 0, r1 = 0, r2 = 0;
  if ((int64_t)blockIdx.x < n_rc) { r1 = RC(blockIdx.x); flo1 = v.rc_lo[r1]; n1 = v.rc_n[r1]; c1 = v.rc_cluster[r1]; }
  if ((int64_t)blockIdx.x + stride < n_rc) { r2 = RC(blockIdx.x + stride); flo2 = v.rc_lo[r2]; n2 = v.rc_n[r2]; c2 = v.rc_cluster[r2]; }
  if ((int64_t)blockIdx.x < n_rc) { h1 = v.cl_head[c1]; if (lane < n1 && n1 <= SNF_WAVE) slot1 = v.FI[flo1 + lane]; }
  for (int64_t it = blockIdx.x; it < n_rc; it += stride) {
    const int32_t r = r1;
    const int32_t flo = flo1, n = n1, c = c1, h = h1; const int32_t slot_pre = slot1;
    flo1 = flo2; n1 = n2; c1 = c2; r1 = r2;
    if (it + stride < n_rc) { h1 = v.cl_head[c1]; if (lane < n1 && n1 <= SNF_WAVE) slot1 = v.FI[flo1 + lane]; }
    if (it + 2 * stride < n_rc) { r2 = RC(it + 2 * stride); flo2 = v.rc_lo[r2]; n2 = v.rc_n[r2]; c2 = v.rc_cluster[r2]; }
    if (n > SNF_WAVE) { if (lane == 0) big_push(v, 1, (int32_t)r); continue; }  // x_big<1>
    const int g = v.seed_grp[h], svtype = grp_svtype(g), task = grp_task(g);
    const bool act = lane < n;
    int32_t slot = 0; uint32_t o = 0; int32_t svl = 0, rs = 0; uint32_t qn = 0;
    int mapq = 0, strand = 0, is_sa = 0, noninline = 0; double nm = 0;
    int32_t mctg = 0, mpos = 0; int bfirst = 0, brev = 0;
    int hap = 0; uint32_t rid = 0; int32_t ps = SNF_PS_NULL_CODE; bool close = false;    // for the lead aggregates (wave_lead_agg)
    const bool want_nm = PHASE && cfg.mode_call_sample;     // rescue_phasing may ask for the mean NM ratio of the leads
    if (act) {
      slot = slot_pre; svl = v.F_svlen[slot];
      const LeadRec r = v.Lrec[v.F_lpos[slot]];
      o = r.orig; rs = r.ref_start; qn = r.qname; mapq = r.mapq; strand = r.strand; is_sa = r.is_sa;
      noninline = r.source != SNF_SRC_INLINE;
      if (PHASE) {
        hap = r.hap; rid = r.read_id;
        ps = (r.ps == SNF_PS_NONE || r.ps == v.t_ps_null[task]) ? SNF_PS_NULL_CODE : r.ps;
      }
      close = (int64_t)r.qry_start <= cfg.dev_min_close_edge_dist || iabs64((int64_t)r.read_len - (int64_t)r.qry_start) <= cfg.dev_min_close_edge_dist;
      if (cfg.qc_nm_measure || want_nm) nm = v.in_nm[o];
      mctg = r.mate_contig; mpos = r.mate_pos; bfirst = r.first; brev = r.rev;
      v.F_sel[slot] = 1;
    }
    if (lane == 0) v.cdflag[r] = 0;
    const int32_t s_svl = wave_sort_i32(svl, act, n, lane, lds.buf);
    const int64_t svlen = wave_center_sorted(s_svl, n, lane);
    const bool single = svtype == SNF_SINGLE_LEFT || svtype == SNF_SINGLE_RIGHT;
    if (!single && svtype != SNF_BND && iabs64(svlen) < cfg.minsvlen_screen) continue;
    // distinct read names, sorted (kept in w1 for d3_rnames)
    const int32_t s_qn = wave_sort_i32((int32_t)qn, act, n, lane, lds.buf);
    const int32_t p_qn = __shfl_up(s_qn, 1, SNF_WAVE);
    const bool qfirst = act && (lane == 0 || p_qn != s_qn);
    const unsigned long long qmask = __ballot(qfirst);
    int64_t nq = __builtin_popcountll(qmask);
    int32_t* a1 = v.w1 + flo;
    if (qfirst) a1[__builtin_popcountll(qmask & ((1ull << lane) - 1ull))] = s_qn;
    __syncthreads();
    int64_t support = nq, support_long = 0;
    const int32_t llo = v.seedL_lo[h], lhi = v.seedL_hi[v.c_last[h]];
    const bool keeplong = v.rc_keeplong[r] && svtype == SNF_INS;
    if (svtype == SNF_INS && svlen >= cfg.long_ins_length) {
      int cl = 0, cu = 0;
      for (int32_t x0 = llo; x0 < lhi; x0 += SNF_WAVE) {
        const int32_t x = x0 + lane;
        if (x < lhi) {
          const int32_t q = (int32_t)v.in_qname[v.LL[x]];
          bool first = true;
          for (int32_t y = llo; y < x; y++) if ((int32_t)v.in_qname[v.LL[y]] == q) { first = false; break; }
          if (first) { cl++; if (!contains_sorted_i32(a1, nq, q)) cu++; }
        }
      }
      support_long = wave_sum32((uint32_t)cl); support += wave_sum32((uint32_t)cu);
    }
    const int32_t s_rs = wave_sort_i32(rs, act, n, lane, lds.buf);
    const int64_t ref_start = wave_center_sorted(s_rs, n, lane);
    const double stdev_pos = wave_stdev_trim_sorted(s_rs, n, lane);
    double stdev_len = NAN; bool precise;
    if (svtype != SNF_BND) { stdev_len = wave_stdev_trim_sorted(s_svl, n, lane); precise = (stdev_pos + stdev_len < (double)cfg.precise); }
    else precise = stdev_pos < (double)cfg.precise;
    int64_t svstart, svend;
    if (svtype == SNF_INS) { svstart = ref_start; svend = ref_start; }
    else if (svtype == SNF_DEL) { svstart = ref_start + svlen; svend = ref_start; }
    else { svstart = ref_start; svend = svstart + iabs64(svlen); }
    const int64_t msum = wave_sum32(act ? (uint32_t)mapq : 0u);      // (<= 64 x 255)
    const int64_t fwd = __builtin_popcountll(__ballot(act && strand == 0));
    int64_t sa = __builtin_popcountll(__ballot(act && is_sa));
    const int64_t src_noninline = __builtin_popcountll(__ballot(act && noninline));
    double nmsum = 0;
    if (cfg.qc_nm_measure) {  // Python sum(): left to right
      for (int i = 0; i < n; i++) {
        const uint64_t bits = wave_bcast_u64((uint64_t)__double_as_longlong(nm), i);
        nmsum += __longlong_as_double((long long)bits);
      }
    }
    int64_t n_all = n;
    if (keeplong) {
      int cs = 0;
      for (int32_t x = llo + lane; x < lhi; x += SNF_WAVE) cs += v.in_is_sa[v.LL[x]];
      sa += wave_sum32((uint32_t)cs); n_all += lhi - llo;
    }
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
    cc.cluster_seed_index = v.prefilter ? -1 : v.seed_bin[h] - v.grp_first_bin[g];
    int64_t rn_len = support;
    bool sel_final = act;     // the leads the call keeps (resolve_bnd narrows them)
    if (svtype == SNF_BND) {  // resolve_bnd (sv.py:625-639)
      const int32_t s_mc = wave_sort_i32(mctg, act, n, lane, lds.buf);
      const int32_t p_mc = __shfl_up(s_mc, 1, SNF_WAVE);
      const bool st = act && (lane == 0 || p_mc != s_mc);
      const unsigned long long smask = __ballot(st);
      const unsigned long long above = lane < 63 ? (smask >> (lane + 1)) : 0ull;
      const int len = st ? ((above ? lane + 1 + __builtin_ctzll(above) : n) - lane) : 0;
      const int maxc = wave_max32(len);
      const unsigned long long best = __ballot(st && len == maxc);
      const int32_t mc = __shfl(s_mc, __builtin_ctzll(best), SNF_WAVE);  // most_common_top: ties -> smallest value
      const bool sel = act && mctg == mc;
      sel_final = sel;
      if (act) v.F_sel[slot] = sel ? 1 : 0;
      const unsigned long long selmask = __ballot(sel);
      const int ns = __builtin_popcountll(selmask);
      const int64_t nfirst = __builtin_popcountll(__ballot(sel && bfirst)), nrev = __builtin_popcountll(__ballot(sel && brev));
      // selected values packed to the front by sorting with +inf for the rest
      const int32_t s_mp = wave_sort_i32(sel ? mpos : INT32_MAX, act, n, lane, lds.buf);
      const int64_t mate_pos = wave_center_sorted(s_mp, ns, lane);
      const int32_t s_q2 = wave_sort_i32(sel ? (int32_t)qn : INT32_MAX, act, n, lane, lds.buf);
      const int32_t p_q2 = __shfl_up(s_q2, 1, SNF_WAVE);
      const bool qf2 = lane < ns && (lane == 0 || p_q2 != s_q2);
      const unsigned long long qm2 = __ballot(qf2);
      // SUPPORT: reads of the selected leads; RNAMES (a1, nq, rn_len) stays the read set of the whole cluster (sv.py:555
      // is taken before resolve_bnd narrows the leads) - they differ only when a cluster mixes mate contigs (--dev-no-resplit)
      cc.support = (int32_t)__builtin_popcountll(qm2);
      cc.mate_contig = mc; cc.mate_ref_start = (int32_t)mate_pos;
      cc.bnd_is_first = (nfirst > ns - nfirst) ? 1 : 0;   // most_common_top: ties -> False
      cc.bnd_is_reverse = (nrev > ns - nrev) ? 1 : 0;
      cc.n_leads = ns;
    } else if (svtype == SNF_INS) cc.support_long = (int32_t)support_long;
    else if (svtype == SNF_DEL) cc.support_sa = (int32_t)src_noninline;
    cc.rn_len = (int32_t)rn_len;
    cc.rn_off = nq;  // stash: number of distinct names already sorted in w1[flo..]
    int32_t best_slot = -1, n_others = 0;
    if (svtype == SNF_INS && !cfg.symbolic) {
      // best lead of annotate_sv (postprocessing.py:33-66): first argmin of |len(seq) - svlen| + |ref_start - pos| * 1.5
      // over the sequence-bearing leads in cluster (= lane) order.  d >= 0, so its bit pattern orders like the value
      const int32_t sl = act ? v.F_seq_len[slot] : -1;
      const bool has = act && sl >= 0;
      const double d = (double)iabs64((int64_t)sl - svlen) + (double)iabs64((int64_t)rs - svstart) * 1.5;
      unsigned long long key = has ? (unsigned long long)__double_as_longlong(d) : ~0ull;
      unsigned long long mn = key;
#pragma unroll
      for (int dd = 32; dd >= 1; dd >>= 1) { const unsigned long long o2 = __shfl_xor(mn, dd, SNF_WAVE); if (o2 < mn) mn = o2; }
      const unsigned long long hm = __ballot(has);
      if (hm) {
        const int bl = __builtin_ctzll(__ballot(has && key == mn));
        best_slot = __shfl(slot, bl, SNF_WAVE);
        n_others = __builtin_popcountll(hm) - 1;
        // read list of the consensus kernel: the other sequence-bearing leads in cluster order, stored in this refined
        // cluster's own slot range [flo, flo + n) (no scan needed to place it)
        const bool oth = has && lane != bl;
        const unsigned long long om = __ballot(oth);
        if (oth) {
          const int w = __builtin_popcountll(om & ((1ull << lane) - 1ull));
          v.crl_off[flo + w] = v.F_seq_off[slot]; v.crl_len[flo + w] = sl;
        }
      }
    }
    CallX x; x.rc = (int32_t)r; x.cluster = c; x.flo = flo; x.fn = n; x.best = best_slot; x.n_others = n_others;
    x.do_cons = (best_slot >= 0 && n_others >= cfg.consensus_min_reads && !cfg.no_consensus) ? 1 : 0; x.cons_id = -1; x.alt_off = 0;
    x.rn_nq = (int32_t)nq; x._pad = 0;
    wave_lead_agg<PHASE>(cfg, lds, lane, n, sel_final, strand, hap, rid, ps, close, want_nm, nm, x);
    if (lane == 0) {
      v.cand[r] = cc;
      v.candx[r] = x;
      v.cdflag[r] = 1;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------ e1w: finalize
// The scalar tail of Task.finalize_candidates - QC, genotype, phase filters, rescue - one call per lane, View::e1_batch calls
// per wave (the tail is a chain of dependent loads and double arithmetic; more calls per wave would leave too few waves to
// hide it).  The aggregates over a call's leads come with the call (CallX::ag_*, formed by d2w_call); calls without them
// (clusters of more than 64 leads) go to x_big<2>.
#define SNF_E1_BATCH_MAX 32
template <int MINW>
__global__ void __launch_bounds__(SNF_WAVE, MINW) e1w_finalize(const View v, int64_t n_unused) {
  IT_SCOPE(9)
  const int E1B = v.e1_batch;      // calls per wave (the launch uses the same number)
  const int lane = threadIdx.x;
  const int64_t n_calls = v.cnt->n_calls;
  for (int64_t base = (int64_t)blockIdx.x * E1B; base < n_calls; base += (int64_t)gridDim.x * E1B) {
    const int64_t i = base + lane;
    if (lane >= E1B || i >= n_calls) continue;
    snf_call_t c = v.calls[i];
    if (v.t_status[c.task_index] != SNF_TASK_OK) continue;
    const CallX x = v.callx[i];
    if (x.fn > SNF_WAVE || !x.ag_valid) { big_push(v, 2, (int32_t)i); continue; }  // x_big<2>
    LeadAgg g;
    g.nstrands = x.ag_nstrands; g.close_edge = x.ag_close_edge;
    g.hp_val = x.ag_hp_val; g.hp_support = x.ag_hp_support; g.hp_other = x.ag_hp_other;
    g.ps_val = x.ag_ps_val; g.ps_support = x.ag_ps_support; g.ps_other = x.ag_ps_other;
    g.nm_row = nullptr; g.has_nm_mean = x.ag_has_nm; g.nm_mean = x.ag_nm_mean;
    finalize_call<1>(v, c, x, g, c.task_index);   // x.fn <= 64 leads
    store_final_fields(v.calls[i], c);
  }
}

// ------------------------------------------------------------------------------------------ d4s: the five coverage samples of every call
// (snf_stage_call.h::d4s_sample_body) over a grid that does not depend on the number of calls (known on the device only)
__global__ void __launch_bounds__(256) d4s_coverage(const View v, int64_t n_unused) {
  IT_SCOPE(12)
  const int64_t n = 5 * (int64_t)v.cnt->n_calls;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (int64_t)gridDim.x * 256) d4s_sample_body(q, v);
}

// ------------------------------------------------------------------------------------------ d5w: coverage.mean()
// numerator of coverage.mean() (exact integer sum of the clipped read lengths per task): coalesced loads and one
// atomic per block.  Reads are grouped by task, so a 4096-read chunk almost always belongs to one task; the few
// chunks that straddle a task boundary fall back to per-read atomics.
__global__ void __launch_bounds__(256) d5w_covsum(const View v, int64_t n_unused) {
  __shared__ unsigned long long part[4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t CH = 4096;
  for (int64_t base = (int64_t)blockIdx.x * CH; base < v.R; base += (int64_t)gridDim.x * CH) {
    const int64_t hi = base + CH < v.R ? base + CH : v.R;
    const int t0 = v.r_task[base], t1 = v.r_task[hi - 1];
    for (int t = t0; t <= t1; t++) {  // one pass per task present in the chunk (one, rarely two)
      if (v.t_cov_exact && v.t_cov_exact[t]) continue;   // d5x_covexact forms this task's sum (mask / uint16 wrap; uniform for the block)
      const int64_t L = v.t_contig_len[t];
      unsigned long long acc = 0;
      for (int64_t r = base + tid; r < hi; r += 256) {
        if (t0 != t1 && v.r_task[r] != t) continue;
        int64_t s = v.r_start[r], e = v.r_end[r];
        if (s < 0) s = 0; if (s > L) s = L;
        if (e < 0) e = 0; if (e > L) e = L;
        if (e > s) acc += (unsigned long long)(e - s);
      }
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
      if (lane == 0) part[wid] = acc;
      __syncthreads();
      if (tid == 0) { const unsigned long long tot = part[0] + part[1] + part[2] + part[3]; if (tot) atomicAdd(&v.t_cov_sum[t], tot); }
      __syncthreads();
    }
  }
}

// Items with more than 64 leads (rare at 30x, a quarter of the clusters at 60x): the serial bodies of the thread kernels,
// but ONE item per wave (lane 0) instead of one per lane - 64 different clusters in one wave diverge on every branch and
// cost the sum of their times - and with the wave running one body uniformly the sorts inside it become cooperative rank sorts
// (SNF_SORT, snf_exact.h).  KIND 0: d1_refine_body (clusters), 1: d2_call_body (refined clusters), 2: e1_finalize_body.
#define SNF_BIG_STAGE_CAP 160   // leads of a cluster x_big<0> keeps in LDS (11 KB of records + 5 KB of scratch rows per wave)
template <int KIND>
__global__ void __launch_bounds__(SNF_WAVE) x_big(View v, int64_t n_unused) {
  __shared__ alignas(16) LeadRec s_rec[KIND == 0 ? SNF_BIG_STAGE_CAP : 1];     // (filled with 16-byte stores)
  __shared__ alignas(16) int32_t s_scr[KIND == 0 ? 8 * SNF_BIG_STAGE_CAP : 5 * SNF_BIG_FINAL_CAP];
  const bool stage = v.stage_cap != 0;    // host switch (SNF_NO_BIG_STAGE=1 clears it): keep staged clusters in LDS
  v.big_wave = 0;   // the bodies below are the ones that skip big items when it is set
  // all 64 lanes run the serial body in lock step on the same data (identical stores, no atomics except the one pool
  // reservation, which one lane makes): the body's sorts are then done by the whole wave, and (refinement) the cluster's
  // lead records and scratch rows sit in LDS, so the body's dependent loads cost LDS latency instead of L2 / HBM latency
  v.wave_uniform = 1;
  const int lane = threadIdx.x;
  const int stripe = blockIdx.x & 63, per = (int)(gridDim.x >> 6);
  const uint32_t cnt = v.big_cnt[(KIND * 64 + stripe) * 16];
  const int32_t* list = v.big_list + ((int64_t)KIND * 64 + stripe) * v.big_cap;
  for (uint32_t k = blockIdx.x >> 6; k < cnt; k += (uint32_t)per) {
    const int64_t item = list[k];
    if (KIND == 0) {
      const int32_t h = v.cl_head[item];
      const int32_t lo = v.seed_lo[h], n = v.seed_hi[v.c_last[h]] - lo;
      __syncthreads();                       // the previous cluster is through with the LDS rows
      if (stage && n <= SNF_BIG_STAGE_CAP) {
        static_assert(sizeof(LeadRec) % 8 == 0, "LeadRec is copied in 8-byte words");
        const uint2* src = (const uint2*)(v.Lrec + lo); uint2* dst = (uint2*)s_rec;      // 8 bytes per lane and step, coalesced
        for (int q = lane; q < n * (int)(sizeof(LeadRec) / 8); q += SNF_WAVE) dst[q] = src[q];
        __syncthreads();
        v.stage_R = s_rec; v.stage_w = s_scr; v.stage_cap = SNF_BIG_STAGE_CAP;
      } else { v.stage_R = nullptr; v.stage_w = nullptr; }
      d1_refine_body(item, v);
    }
    else {
      __syncthreads();                       // the previous item is through with the LDS rows
      if (stage) { v.stage_w = s_scr; v.stage_cap = SNF_BIG_FINAL_CAP; } else v.stage_w = nullptr;   // rows of d2_call_body / collect_agg_wave
      if (KIND == 1) d2_call_body(item, v); else e1_finalize_body(item, v);
    }
  }
}

}  // namespace snf
