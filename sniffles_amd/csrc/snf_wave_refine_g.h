// snf_wave_refine_g.h - the refinement stage (merge_inner, resplit, resplit_bnd; cluster.py:85-216) for SMALL merged clusters:
// eight clusters per wave.
//
// d1w_refine (snf_wave_refine.h) gives every merged cluster a wave, one lead per lane; a 30x genome has 92 k clusters of 7.4 leads
// on average - one lane in eight had a lead.  Here a wave is cut into 64 / G groups of G = 8 lanes and every group takes a cluster of
// at most G leads, with the arithmetic of d1w_refine and every wave-wide primitive in its group-wide form (the pattern of
// d2g_call<8>, snf_wave_call_g.h): ballots sliced per group, shuffles inside the group, the sequential bin-merge state machine of
// resplit stepped in lock-step by all groups that still have work.  Clusters with more leads go to hand-over list 2, which
// d1w_refine walks next (it passes those beyond 64 leads on to x_big<0>).  Both kernels write the same F / FI / refined-cluster tables.
//
// Control flow is wave-uniform throughout: a group that is done, or has no cluster, keeps executing with its predicate off.
#pragma once
#include "snf_wave_call_g.h"

namespace snf {

// inclusive prefix sums inside the groups of G lanes
template <int G> SNF_D int64_t gscan64(int64_t x, int gl) {
#pragma unroll
  for (int d = 1; d < G; d <<= 1) { const int64_t y = __shfl_up(x, d, SNF_WAVE); if (gl >= d) x += y; }
  return x;
}
template <int G> SNF_D int32_t gscan32(int32_t x, int gl) {
#pragma unroll
  for (int d = 1; d < G; d <<= 1) { const int32_t y = __shfl_up(x, d, SNF_WAVE); if (gl >= d) x += y; }
  return x;
}
template <int G> SNF_D int64_t gshfl_i64(int64_t x, int src, int gbase) { return (int64_t)gshfl_u64<G>((uint64_t)x, src, gbase); }

struct GroupLds { int32_t perm[SNF_WAVE], seg_start[SNF_WAVE], seg_key[SNF_WAVE]; };

template <int G>
__global__ void __launch_bounds__(SNF_WAVE) d1g_refine(const View v, int64_t n_unused) {
  IT_SCOPE(4)
  static_assert(G == 8, "group width in use");
  constexpr int NG = SNF_WAVE / G;
  __shared__ GroupLds lds;
  const int lane = threadIdx.x, gl = lane & (G - 1), gbase = lane - gl, gi = lane / G;
  const snf_config_t& cfg = v.cfg;
  const int64_t n_clusters = v.cnt->n_clusters;
  const int64_t stride = (int64_t)gridDim.x * NG;
  // software pipeline over the wave's items: the header of item k + 2 and the lead records of item k + 1 are in flight while item
  // k is processed
  auto header = [&](int64_t base_) -> ClusterHdr { const int64_t c = base_ + gi; return c < n_clusters ? v.chdr[c] : ClusterHdr{}; };
  auto record = [&](const ClusterHdr& h) -> LeadRec { LeadRec r{}; if (gl < h.n && h.n <= G) r = v.Lrec[h.lo + gl]; return r; };
  ClusterHdr hd_cur = header((int64_t)blockIdx.x * NG), hd_nxt = header((int64_t)blockIdx.x * NG + stride);
  LeadRec rec_cur = record(hd_cur);
  int64_t slice_used = 0;     // bytes of this wave's private fused-sequence slice that are taken
  for (int64_t base = (int64_t)blockIdx.x * NG; base < n_clusters; base += stride) {
    const ClusterHdr hd = hd_cur; const LeadRec rec = rec_cur;
    hd_cur = hd_nxt;
    rec_cur = record(hd_cur);
    hd_nxt = header(base + 2 * stride);
    const int64_t c = base + gi;
    bool valid = c < n_clusters && hd.n > 0;
    // clusters that do not fit a group: d1w_refine, from list 2
    d2list_push(v, 2, valid && hd.n > G && gl == 0, (int32_t)c, lane, hd.n);
    if (hd.n > G) valid = false;
    const int32_t lo = hd.lo, n = valid ? hd.n : 0;
    const int nmax = __builtin_amdgcn_readfirstlane(wave_max32(n));
    if (nmax == 0) continue;
    const int svtype = grp_svtype(hd.grp);
    const bool act = gl < n;
    // ---- one lead per lane
    uint32_t o = 0;
    int32_t ref_start = 0, ref_end = 0, qry_start = 0, qry_end = 0, svlen = 0, seq_len = -1, mate_pos = 0, mate_contig = 0;
    uint32_t qname = 0; int64_t seq_off = 0; int strand = 0, is_first = 0;
    if (act) {
      o = rec.orig; ref_start = rec.ref_start; svlen = rec.svlen; seq_len = rec.seq_len; seq_off = rec.seq_off;
      ref_end = rec.ref_end; qry_start = rec.qry_start; qry_end = rec.qry_end; qname = rec.qname; strand = rec.strand;
      mate_pos = rec.mate_pos; mate_contig = rec.mate_contig; is_first = rec.first;
    }
    int m = n;                 // number of leads after fusion
    int32_t f_orig = (int32_t)o, f_svlen = svlen, f_seq_len = seq_len, f_lp = gl; int64_t f_seq_off = seq_off;

    // ---- merge_inner (INS / DEL clusters)
    const bool indel = valid && (svtype == SNF_INS || svtype == SNF_DEL);
    int fa = -1;  // first appearance of this read's qname in cluster order
    if (__ballot(indel)) {
      for (int i = 0; i < nmax; i++) {
        const uint32_t qi = (uint32_t)gshfl_i32<G>((int32_t)qname, i, gbase);
        if (fa < 0 && i < n && qi == qname) fa = i;
      }
    }
    // Every read of a cluster appears once (the usual case): the (read, ref_start) order is the cluster order and nothing can fuse.
    const bool fuse = indel && gballot<G>(act && fa != gl, gbase) != 0ull;      // (group-uniform)
    if (__ballot(fuse)) {
      const bool fact0 = act && fuse;
      const int thr = hd.repeat ? -1 : cfg.cluster_merge_pos;
      const uint64_t key = fact0 ? (((uint64_t)(uint32_t)fa << 40) | ((uint64_t)((uint32_t)ref_start ^ 0x80000000u) << 8) | (uint32_t)gl) : ~0ull;
      const int rank = grank<G>(key, gbase, nmax);
      __syncthreads();
      if (fact0) lds.perm[gbase + rank] = gl;
      __syncthreads();
      const int src_ = gbase + (fuse ? lds.perm[gbase + (gl < n ? gl : 0)] : gl);
      // everything below is in sorted order: lane r of the group holds the r-th lead of the (read, ref_start) order
      const int s_fa = __shfl(fa, src_, SNF_WAVE);
      const int32_t s_rs = __shfl(ref_start, src_, SNF_WAVE), s_re = __shfl(ref_end, src_, SNF_WAVE), s_qs = __shfl(qry_start, src_, SNF_WAVE), s_qe = __shfl(qry_end, src_, SNF_WAVE);
      const int32_t s_svlen = __shfl(svlen, src_, SNF_WAVE), s_seq_len = __shfl(seq_len, src_, SNF_WAVE);
      const int64_t s_seq_off = __shfl(seq_off, src_, SNF_WAVE);
      const int s_strand = __shfl(strand, src_, SNF_WAVE);
      const uint32_t s_o = __shfl(o, src_, SNF_WAVE); const int s_lp = __shfl(gl, src_, SNF_WAVE);
      // neighbour r - 1
      const int p_fa = __shfl_up(s_fa, 1, SNF_WAVE);
      const int32_t p_rs = __shfl_up(s_rs, 1, SNF_WAVE), p_re = __shfl_up(s_re, 1, SNF_WAVE);
      const int32_t p_qs = __shfl_up(s_qs, 1, SNF_WAVE), p_qe = __shfl_up(s_qe, 1, SNF_WAVE);
      const int p_strand = __shfl_up(s_strand, 1, SNF_WAVE);
      bool mg = false;
      if (fact0 && gl > 0 && p_fa == s_fa) {
        mg = (thr == -1) ||
             (((iabs64((int64_t)s_rs - p_re) < thr || iabs64((int64_t)s_rs - p_rs) < thr) &&
               (iabs64((int64_t)s_qs - p_qe) < thr || iabs64((int64_t)s_qs - p_qs) < thr)) &&
              (p_strand == s_strand));  // == head strand: every member of a fused run shares it
      }
      const bool start = fact0 && !mg;
      const unsigned long long smask = gballot<G>(start, gbase);
      const unsigned long long above = (gl < G - 1) ? (smask >> (gl + 1)) : 0ull;
      const int seg_end = above ? gl + __builtin_ctzll(above) : n - 1;      // (group-relative)
      const int64_t ps_svlen = gscan64<G>(fact0 ? (int64_t)s_svlen : 0, gl);
      const int64_t ps_seq = gscan64<G>((fact0 && s_seq_len >= 0) ? (int64_t)s_seq_len : 0, gl);
      const int32_t ps_has = gscan32<G>((fact0 && s_seq_len >= 0) ? 1 : 0, gl);
      const int se = seg_end < 0 ? 0 : seg_end;
      const int64_t e_svlen = gshfl_i64<G>(ps_svlen, se, gbase), e_seq = gshfl_i64<G>(ps_seq, se, gbase);
      const int32_t e_has = gshfl_i32<G>(ps_has, se, gbase);
      const int64_t x_svlen = ps_svlen - (fact0 ? s_svlen : 0), x_seq = ps_seq - ((fact0 && s_seq_len >= 0) ? s_seq_len : 0);
      const int32_t x_has = ps_has - ((fact0 && s_seq_len >= 0) ? 1 : 0);
      const int nparts = seg_end - gl + 1;
      const int64_t tot_svlen = e_svlen - x_svlen, tot_seq = e_seq - x_seq;
      const bool seq_ok = (e_has - x_has) == nparts;
      // fused sequence: concatenation in the pool's fused region (curr_lead.seq += to_merge.seq); the reservation and the byte
      // copies are the wave's (all groups together), as in d1w_refine
      int64_t new_off = 0; bool need_copy = start && seq_ok && nparts > 1;
      if (__ballot(need_copy)) {
        const int64_t mine = need_copy ? tot_seq : 0;
        const int64_t incl = wave_incl_scan64(mine, lane);
        const int64_t wave_total = __shfl(incl, 63, SNF_WAVE);
        int64_t pbase;
        if (slice_used + wave_total <= (v.pool_slice >> 1)) {   // (wave-uniform) this wave's own slice - its first half, the second is d1w_refine's: no atomic
          pbase = v.pool_len + (int64_t)blockIdx.x * v.pool_slice + slice_used;
          slice_used += wave_total;
        } else {
          int64_t got = 0;
          if (lane == 0) got = (int64_t)atomicAdd(&v.cnt->pool_extra_used, (unsigned long long)wave_total);
          pbase = v.pool_extra_base + __shfl(got, 0, SNF_WAVE);
        }
        new_off = pbase + (incl - mine);
        if (need_copy && new_off + tot_seq > v.pool_cap) { atomicOr(&v.cnt->overflow, 1); need_copy = false; }
      }
      {  // the parts that are copied, as in d1w_refine (the copy itself is the wave's: all groups' parts, four at a time)
        const unsigned long long upto = (2ull << gl) - 1ull;
        const int my_start = (fact0 && (smask & upto)) ? 63 - __builtin_clzll(smask & upto) : 0;      // (group-relative)
        const bool p_act = fact0 && __shfl((int)need_copy, gbase + my_start, SNF_WAVE) != 0;
        const int64_t p_dst = __shfl(new_off, gbase + my_start, SNF_WAVE) + (x_seq - __shfl(x_seq, gbase + my_start, SNF_WAVE));
        wave_copy_parts(v, lane, p_act, s_seq_off, p_act ? s_seq_len : 0, p_dst);
      }
      const bool ok_seq = start && seq_ok && (nparts == 1 || need_copy);
      // compact the fused leads (start lanes) to the front of the group
      const int mm = __builtin_popcountll(smask);
      const int kidx = __builtin_popcountll(smask & ((1ull << gl) - 1ull));
      __syncthreads();
      if (start) lds.perm[gbase + kidx] = gl;
      __syncthreads();
      const int src2 = gbase + (fuse ? lds.perm[gbase + (gl < mm ? gl : 0)] : gl);
      const int32_t t_seq_len = ok_seq ? (nparts == 1 ? s_seq_len : (int32_t)tot_seq) : -1;
      const int64_t t_seq_off = ok_seq ? (nparts == 1 ? s_seq_off : new_off) : 0;
      const int32_t g_orig = (int32_t)__shfl(s_o, src2, SNF_WAVE), g_lp = __shfl(s_lp, src2, SNF_WAVE);
      const int32_t g_svlen = (int32_t)__shfl(tot_svlen, src2, SNF_WAVE);
      const int32_t g_seq_len = __shfl(t_seq_len, src2, SNF_WAVE);
      const int64_t g_seq_off = __shfl(t_seq_off, src2, SNF_WAVE);
      if (fuse) { m = mm; f_orig = g_orig; f_lp = g_lp; f_svlen = g_svlen; f_seq_len = g_seq_len; f_seq_off = g_seq_off; }
      __syncthreads();
    }
    const bool fact = gl < m;
    if (fact) {
      v.F_orig[lo + gl] = f_orig; v.F_svlen[lo + gl] = f_svlen; v.F_lpos[lo + gl] = lo + f_lp;
      v.F_seq_len[lo + gl] = f_seq_len; v.F_seq_off[lo + gl] = f_seq_off;
    }
    bool done = !valid;

    // ---- resplit_bnd: group by (mate_contig, is_first) in first-appearance order, chain 1-kb bins
    const bool bnd = valid && svtype == SNF_BND;
    if (bnd && (m <= 1 || cfg.dev_no_resplit)) {
      if (fact) v.FI[lo + gl] = lo + gl;
      if (gl == 0) rc_emit(v, lo, m, (int32_t)c, true);
      done = true;
    }
    const bool bnd2 = bnd && !done;
    if (__ballot(bnd2)) {
      const int thr = cfg.cluster_merge_bnd;
      int fb = -1;
      for (int i = 0; i < nmax; i++) {
        const int32_t mc = gshfl_i32<G>(mate_contig, i, gbase); const int fi = gshfl_i32<G>(is_first, i, gbase);
        if (fb < 0 && i < m && mc == mate_contig && fi == is_first) fb = i;
      }
      const bool ab = fact && bnd2;
      const int64_t pb = thr > 0 ? ((int64_t)mate_pos / thr) * thr : 0;
      const uint64_t key = ab ? (((uint64_t)(uint32_t)fb << 48) | ((uint64_t)(uint32_t)((int64_t)pb + 0x80000000ll) << 8) | (uint32_t)gl) : ~0ull;
      const int rank = grank<G>(key, gbase, nmax);
      __syncthreads();
      if (ab) lds.perm[gbase + rank] = gl;
      __syncthreads();
      const int src_ = gbase + (bnd2 ? lds.perm[gbase + (gl < m ? gl : 0)] : gl);
      const int s_fb = __shfl(fb, src_, SNF_WAVE); const int64_t s_pb = __shfl(pb, src_, SNF_WAVE); const int s_j = __shfl(gl, src_, SNF_WAVE);
      const int p_fb = __shfl_up(s_fb, 1, SNF_WAVE); const int64_t p_pb = __shfl_up(s_pb, 1, SNF_WAVE);
      const bool brk = ab && (gl == 0 || p_fb != s_fb || (s_pb - p_pb > thr));
      const unsigned long long bmask = gballot<G>(brk, gbase);
      if (ab) v.FI[lo + gl] = lo + s_j;
      if (brk) {
        const unsigned long long above = (gl < G - 1) ? (bmask >> (gl + 1)) : 0ull;
        const int end = above ? gl + 1 + __builtin_ctzll(above) : m;  // exclusive: position of the next chain start
        rc_emit(v, lo + gl, end - gl, (int32_t)c, false);
      }
      if (bnd2) done = true;
      __syncthreads();
    }

    // ---- resplit on |svlen| bins of 20 (cluster.py:125-161)
    if (!done && (cfg.dev_no_resplit_repeat || cfg.dev_no_resplit)) {
      if (fact) v.FI[lo + gl] = lo + gl;
      if (gl == 0) rc_emit(v, lo, m, (int32_t)c, true);
      done = true;
    }
    const int rb = cfg.cluster_resplit_binsize;
    const uint32_t av = f_svlen < 0 ? (uint32_t)(-(int64_t)f_svlen) : (uint32_t)f_svlen;      // (32-bit: a 64-bit division is ~150 instructions here)
    const int32_t bin = (int32_t)((av / (uint32_t)rb) * (uint32_t)rb);
    // one bin: the cluster stays whole, in its order
    const int32_t bin0 = gshfl_i32<G>(bin, 0, gbase);
    if (!done && gballot<G>(fact && bin != bin0, gbase) == 0ull) {
      if (fact) v.FI[lo + gl] = lo + gl;
      if (gl == 0) rc_emit(v, lo, m, (int32_t)c, true);
      done = true;
    }
    if (__ballot(!done) == 0ull) continue;
    {
      const bool rs = !done;                    // (group-uniform) this group's cluster is split by |svlen| bins
      const bool ar = fact && rs;
      const uint64_t key = ar ? (((uint64_t)(uint32_t)bin << 8) | (uint32_t)gl) : ~0ull;
      const int rank = grank<G>(key, gbase, nmax);
      __syncthreads();
      if (ar) lds.perm[gbase + rank] = gl;
      __syncthreads();
      const int src_ = gbase + (rs ? lds.perm[gbase + (gl < m ? gl : 0)] : gl);
      const int32_t s_bin = __shfl(bin, src_, SNF_WAVE); const int s_k = __shfl(gl, src_, SNF_WAVE);
      const int32_t p_bin = __shfl_up(s_bin, 1, SNF_WAVE);
      const bool sstart = ar && (gl == 0 || p_bin != s_bin);
      const unsigned long long smask = gballot<G>(sstart, gbase);
      const int nb = __builtin_popcountll(smask);
      const int sidx = __builtin_popcountll(smask & ((2ull << gl) - 1ull)) - 1;  // segment of this position
      __syncthreads();
      if (sstart) { lds.seg_start[gbase + sidx] = gl; lds.seg_key[gbase + sidx] = s_bin; }
      __syncthreads();
      // The sequential bin-merge state machine with the reference's index quirks, its arrays one element per lane of the group (lane
      // s: segment s, lane t: surviving bin t); all groups that still merge step together.
      const int32_t K = (rs && gl < nb) ? lds.seg_key[gbase + gl] : 0, ST = (rs && gl < nb) ? lds.seg_start[gbase + gl] : m;
      int32_t NC = gl, HEAD = gl, TAIL = gl, NXT = -1, SEGOUT = 0, RCS = 0, RCL = 0;
      int cntc = rs ? nb : 0, i = 1;
      while (__ballot(cntc > 1 && i < cntc)) {
        const bool on = cntc > 1 && i < cntc;
        const int im1 = (i == 0) ? cntc - 1 : i - 1;  // Python negative index: new_clusters[-1]
        const int ii = on ? i : 0, jj = on ? im1 : 0;
        const int lb = gshfl_i32<G>(NC, jj, gbase), cb = gshfl_i32<G>(NC, ii, gbase);
        const int64_t last = gshfl_i32<G>(K, lb & (G - 1), gbase), curr = gshfl_i32<G>(K, cb & (G - 1), gbase);
        const int64_t mn = curr < last ? curr : last;
        const double t = (double)mn * cfg.cluster_merge_len;
        const double thr = ((double)cfg.minsvlen >= t) ? (double)cfg.minsvlen : t;
        const int64_t diff = curr > last ? curr - last : last - curr;
        const bool mrg = on && (double)diff <= thr;
        const int tcb = gshfl_i32<G>(TAIL, cb & (G - 1), gbase), hlb = gshfl_i32<G>(HEAD, lb & (G - 1), gbase), tlb = gshfl_i32<G>(TAIL, lb & (G - 1), gbase);
        const int32_t up = __shfl_down(NC, 1, SNF_WAVE);      // new_clusters.pop(im1)
        if (mrg) {
          if (gl == tcb) NXT = hlb;
          if (gl == cb) TAIL = tlb;
          if (gl >= im1 && gl + 1 < cntc) NC = up;
          cntc--;
          i = (i - 2 > 0) ? i - 2 : 0;
        } else if (on) i++;
      }
      // output order: the surviving bins in order, each the chain of its segments
      int outp = 0, t2 = 0, sg = -1, start = 0;
      while (__ballot(t2 < cntc)) {
        const bool on = t2 < cntc;
        const int nc_t2 = gshfl_i32<G>(NC, on ? t2 : 0, gbase);
        const int hd_t2 = gshfl_i32<G>(HEAD, nc_t2 & (G - 1), gbase);
        if (on && sg < 0) { start = outp; sg = hd_t2; }
        const int sgi = (on && sg >= 0) ? sg : 0;
        const int st_a = gshfl_i32<G>(ST, sgi, gbase), st_b = gshfl_i32<G>(ST, (sgi + 1) & (G - 1), gbase), nx = gshfl_i32<G>(NXT, sgi, gbase);
        if (on) {
          if (gl == sg) SEGOUT = outp;
          outp += (sg + 1 < nb ? st_b : m) - st_a;
          sg = nx;
          if (sg < 0) { if (gl == t2) { RCS = start; RCL = outp - start; } t2++; }
        }
      }
      const int sx = sidx < 0 ? 0 : sidx;
      const int32_t my_out = gshfl_i32<G>(SEGOUT, sx, gbase), my_st = gshfl_i32<G>(ST, sx, gbase);
      if (ar) v.FI[lo + my_out + (gl - my_st)] = lo + s_k;
      if (rs && gl < cntc) rc_emit(v, lo + RCS, RCL, (int32_t)c, true);
      __syncthreads();
    }
  }
}

}  // namespace snf
