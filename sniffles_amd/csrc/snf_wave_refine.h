// snf_wave_refine.h - gfx950 wave-per-cluster implementation of the refinement stage (merge_inner,
// resplit, resplit_bnd; cluster.py:85-216) for clusters of at most 64 leads (one lead per lane).
//
// Everything stays in registers: sorts are rank sorts over packed 64-bit keys (n broadcasts of one
// lane's key via v_readlane), the same-read fusion is a segmented scan over the sorted order, and the
// sequential resplit bin-merge state machine (k <= a handful of bins) keeps its arrays one element per lane and walks them with
// v_readlane.  The common shapes leave early: every read once in the cluster (nothing to sort or fuse), one |svlen| bin.  Clusters
// with more than 64 leads take the thread-per-cluster path (d1_refine_body), which is also what the
// host emulation executes; both write the same F / FI / refined-cluster tables.
#pragma once
#include "snf_stage_call.h"

namespace snf {

#define SNF_WAVE 64

SNF_D uint64_t wave_bcast_u64(uint64_t x, int src) {
  uint32_t lo = __builtin_amdgcn_readlane((uint32_t)x, src), hi = __builtin_amdgcn_readlane((uint32_t)(x >> 32), src);
  return ((uint64_t)hi << 32) | lo;
}
SNF_D int32_t wave_bcast_i32(int32_t x, int src) { return (int32_t)__builtin_amdgcn_readlane((uint32_t)x, src); }

// Wave-wide scans and reductions through DPP (row shifts inside the rows of 16 lanes, then row_bcast:15 / :31 for the row
// totals): six dependent VALU operations of a few cycles each.  The `__shfl_up` / `__shfl_xor` forms these replace compile to
// ds_bpermute - six dependent round trips through the LDS crossbar (~100 cycles each) per scan, and the wave kernels are
// chains of them.  All 64 lanes must be active (the wave kernels' control flow is wave-uniform; idle lanes carry zeros).
#define SNF_DPP_STEPS(STEP) STEP(0x111, 0xf) STEP(0x112, 0xf) STEP(0x114, 0xf) STEP(0x118, 0xf) STEP(0x142, 0xa) STEP(0x143, 0xc)
SNF_D int32_t wave_incl_scan(int32_t x, int lane) {   // inclusive prefix sum across the wave
  (void)lane;
#define SNF_STEP(ctrl, rows) x += __builtin_amdgcn_update_dpp(0, x, ctrl, rows, 0xf, false);
  SNF_DPP_STEPS(SNF_STEP)
#undef SNF_STEP
  return x;
}
SNF_D int64_t wave_incl_scan64(int64_t x, int lane) {
  (void)lane;
#define SNF_STEP(ctrl, rows) { const uint32_t lo_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(uint64_t)x, ctrl, rows, 0xf, false), \
                                              hi_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)((uint64_t)x >> 32), ctrl, rows, 0xf, false); \
                               x += (int64_t)(((uint64_t)hi_ << 32) | lo_); }
  SNF_DPP_STEPS(SNF_STEP)
#undef SNF_STEP
  return x;
}
SNF_D int32_t wave_incl_max(int32_t x) {
#define SNF_STEP(ctrl, rows) { const int32_t y_ = __builtin_amdgcn_update_dpp(x, x, ctrl, rows, 0xf, false); x = y_ > x ? y_ : x; }
  SNF_DPP_STEPS(SNF_STEP)
#undef SNF_STEP
  return x;
}
// the value lane 63 holds, as a wave-uniform value
SNF_D int64_t wave_last64(int64_t x) {
  const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)(uint64_t)x, 63), hi = __builtin_amdgcn_readlane((uint32_t)((uint64_t)x >> 32), 63);
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

// rank of `key` among the first n lanes' keys (keys are distinct: the lane index is part of the key)
SNF_D int wave_rank(uint64_t key, int n) {
  int r = 0;
  for (int i = 0; i < n; i++) r += (wave_bcast_u64(key, i) < key) ? 1 : 0;
  return r;
}

// an item with more than 64 leads: remembered for x_big (called by one lane)
SNF_D void big_push(const View& v, int kind, int32_t item) {
  const int s = item & 63;
  const uint32_t slot = atomicAdd(&v.big_cnt[(kind * 64 + s) * 16], 1u);
  v.big_list[((int64_t)kind * 64 + s) * v.big_cap + slot] = item;
}

// merge_inner's fused sequences (curr_lead.seq += to_merge.seq): every part - one per lane that holds one - is copied from its place in
// the input pool to its place in the fused lead's slice, all 64 lanes on one part at a time.  The parts are taken FOUR at a time: their
// loads are requested together and stored together.  (One part after the other - load, store, next part - was a dependent round trip per
// part: ~100 us per fusing cluster, a third of a wave's time in d1w_refine; source and destination never overlap - the slices lie behind
// the input sequences.)  Measured and not kept (round 6, profiles/ab_r06_17.log): clusters with more than eight parts leaving their copies
// to a copy kernel of their own behind d1w_refine (a wave per part) - the one workgroup d1w_refine ends with goes from 72 to 46-55 us,
// the kernel from 72 to 63-67, and the copy kernel's launch costs the chain the 11 us back: 0.932-0.934 against 0.935-0.937 ms per step.
SNF_D void wave_copy_parts(const View& v, int lane, bool p_act, int64_t p_src, int32_t p_len, int64_t p_dst) {
  typedef uint4 __attribute__((aligned(1))) u128_any;
  unsigned long long pm = __ballot(p_act && p_len > 0);
  while (pm) {
    int z[4] = {0, 0, 0, 0}, np = 0;
    while (pm && np < 4) { z[np++] = __builtin_ctzll(pm); pm &= pm - 1ull; }
    int64_t so[4], dt[4]; int32_t sl[4]; uint4 buf[4]; uint8_t tb[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      so[k] = __shfl(p_src, z[k], SNF_WAVE); dt[k] = __shfl(p_dst, z[k], SNF_WAVE);
      const int32_t l = __shfl(p_len, z[k], SNF_WAVE);
      sl[k] = k < np ? l : 0;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int32_t nfull = sl[k] & ~15;
      buf[k] = make_uint4(0, 0, 0, 0); tb[k] = 0;
      if (lane * 16 < nfull) buf[k] = *(const u128_any*)(v.pool + so[k] + lane * 16);
      if (nfull + lane < sl[k]) tb[k] = v.pool[so[k] + nfull + lane];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int32_t nfull = sl[k] & ~15;
      if (lane * 16 < nfull) *(u128_any*)(v.pool + dt[k] + lane * 16) = buf[k];
      if (nfull + lane < sl[k]) v.pool[dt[k] + nfull + lane] = tb[k];
      for (int32_t bb = SNF_WAVE * 16 + lane * 16; bb < nfull; bb += SNF_WAVE * 16)      // (sequences beyond 1 KB: the rest, 1 KB per step)
        *(u128_any*)(v.pool + dt[k] + bb) = *(const u128_any*)(v.pool + so[k] + bb);
    }
  }
}

// hand-over list 2 (clusters d1g_refine<8> left to d1w_refine): 64 stripes with a counter each, as the lists of the call kernels
// (snf_wave_call.h); consumer side: prefix of the stripes' counts in LDS, item i = entry i - pre[s] of stripe s
SNF_D int64_t d1list_prefix(const View& v, int lane, int32_t* pre) {
  int32_t cnt = (int32_t)v.d2cnt[(2 * 64 + lane) * 16];
  if (cnt > (int32_t)v.d2cap) cnt = (int32_t)v.d2cap;       // (overflowed stripe: flagged by the producer, the fetch fails)
  const int32_t inc = wave_incl_scan(cnt, lane);
  if (lane == 0) pre[0] = 0;
  pre[lane + 1] = inc;
  __syncthreads();
  return pre[64];
}
SNF_D int32_t d1list_at(const View& v, int64_t i, const int32_t* pre) {
  int lo = 0, hi = 63;   // last stripe s with pre[s] <= i
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (pre[mid] <= (int32_t)i) lo = mid; else hi = mid - 1; }
  return v.d2_list[2][(int64_t)lo * v.d2cap + (i - pre[lo])];
}

struct WaveLds {
  int32_t perm[SNF_WAVE];      // scatter target for permutations
  int32_t seg_start[SNF_WAVE]; // resplit: segment (distinct bin) start position in sorted order
  int32_t seg_key[SNF_WAVE];   //          its bin
};

// sorted-order gather: lane r receives the value held by the lane whose rank is r
#define SNF_PERMUTE_SETUP(rank_, active_)         \
  __syncthreads();                                \
  if (active_) lds.perm[rank_] = lane;            \
  __syncthreads();                                \
  const int src_ = lds.perm[lane < n_ ? lane : 0];
#define SNF_GATHER(x) __shfl((x), src_, SNF_WAVE)

#ifdef SNF_CONS_PROFILE
#define SNF_RT_DECL unsigned long long rt_acc[6] = {0, 0, 0, 0, 0, 0}; unsigned long long rt_t = __builtin_amdgcn_s_memtime();
#define SNF_RT(k) do { const unsigned long long rt_n = __builtin_amdgcn_s_memtime(); rt_acc[k] += rt_n - rt_t; rt_t = rt_n; } while (0)
#define SNF_RT_FLUSH() do { if (lane == 0) for (int pk = 0; pk < 6; pk++) atomicAdd(&v.cnt->dbg[24 + pk], rt_acc[pk]); } while (0)
#else
#define SNF_RT_DECL
#define SNF_RT(k) do { } while (0)
#define SNF_RT_FLUSH() do { } while (0)
#endif
// one block = one wave = one merged cluster per loop iteration (grid-stride over clusters)
__global__ void __launch_bounds__(SNF_WAVE) d1w_refine(const View v, int64_t n_unused) {
  IT_SCOPE(5)
  __shared__ WaveLds lds;
  const int lane = threadIdx.x;
  const snf_config_t& cfg = v.cfg;
  // the clusters of this launch: all of them, or (View::d1_from_list) those d1g_refine<8> handed on through list 2
  __shared__ int32_t d1pre[65];
  const bool from_list = v.d1_from_list != 0;
  const int64_t n_clusters = from_list ? d1list_prefix(v, lane, d1pre) : v.cnt->n_clusters;
  auto CL = [&](int64_t it) -> int64_t { return from_list ? (int64_t)d1list_at(v, it, d1pre) : it; };
  // software pipeline over this wave's clusters: the header of cluster k+2 and the lead records of cluster k+1 are
  // in flight while cluster k is processed (the kernel is bound by dependent global-load latency, not bandwidth)
  const int64_t stride = gridDim.x;
  int64_t c_cur = blockIdx.x < n_clusters ? CL(blockIdx.x) : 0, c_nxt = blockIdx.x + stride < n_clusters ? CL(blockIdx.x + stride) : 0;
  ClusterHdr hd_cur = blockIdx.x < n_clusters ? v.chdr[c_cur] : ClusterHdr{};
  ClusterHdr hd_nxt = blockIdx.x + stride < n_clusters ? v.chdr[c_nxt] : ClusterHdr{};
  LeadRec rec_cur{};
  if (lane < hd_cur.n && hd_cur.n <= SNF_WAVE) rec_cur = v.Lrec[hd_cur.lo + lane];
  SNF_RT_DECL
  // bytes of this wave's private fused-sequence slice that are taken (behind d1g_refine<8> the second half of the slice: the
  // first is that kernel's)
  int64_t slice_used = from_list ? (v.pool_slice >> 1) : 0;
  for (int64_t it = blockIdx.x; it < n_clusters; it += stride) {
    SNF_RT(5);   // tail of the previous cluster (stores, resplit)
    const ClusterHdr hd = hd_cur; const LeadRec rec = rec_cur;
    const int64_t c = c_cur;
    hd_cur = hd_nxt; c_cur = c_nxt;
    if (it + stride < n_clusters && lane < hd_cur.n && hd_cur.n <= SNF_WAVE) rec_cur = v.Lrec[hd_cur.lo + lane];
    if (it + 2 * stride < n_clusters) { c_nxt = CL(it + 2 * stride); hd_nxt = v.chdr[c_nxt]; }
    const int32_t lo = hd.lo, n = hd.n;
    if (n > SNF_WAVE) { if (lane == 0) big_push(v, 0, (int32_t)c); continue; }   // big clusters: x_big<0>
    if (n <= 0) continue;
    const int svtype = grp_svtype(hd.grp);
    const bool act = lane < n;
    // ---- load one lead per lane
    uint32_t o = 0;
    int32_t ref_start = 0, ref_end = 0, qry_start = 0, qry_end = 0, svlen = 0, seq_len = -1, mate_pos = 0, mate_contig = 0;
    uint32_t qname = 0; int64_t seq_off = 0; int strand = 0, is_first = 0;
    if (act) {
      const LeadRec& r = rec;
      o = r.orig; ref_start = r.ref_start; svlen = r.svlen; seq_len = r.seq_len; seq_off = r.seq_off;
      ref_end = r.ref_end; qry_start = r.qry_start; qry_end = r.qry_end; qname = r.qname; strand = r.strand;
      mate_pos = r.mate_pos; mate_contig = r.mate_contig; is_first = r.first;
    }
    int m = n;                 // number of leads after fusion
    int32_t f_orig = (int32_t)o, f_svlen = svlen, f_seq_len = seq_len, f_lp = lane; int64_t f_seq_off = seq_off;

    if (svtype == SNF_INS || svtype == SNF_DEL) {
      // ---- merge_inner
      const int thr = hd.repeat ? -1 : cfg.cluster_merge_pos;
      int fa = -1;  // first appearance of this read's qname in cluster order
      for (int i = 0; i < n; i++) {
        uint32_t qi = (uint32_t)wave_bcast_i32((int32_t)qname, i);
        if (fa < 0 && qi == qname) fa = i;
      }
      SNF_RT(0);   // records in registers, first appearance
      // Every read of the cluster appears once (the usual case): the (read, ref_start) order is the cluster order and nothing can
      // fuse - the leads stay where they are (m = n, f_* as loaded).  Wave-uniform.
      if (__ballot(act && fa != lane) != 0ull) {
      uint64_t key = act ? (((uint64_t)(uint32_t)fa << 40) | ((uint64_t)((uint32_t)ref_start ^ 0x80000000u) << 8) | (uint32_t)lane) : ~0ull;
      const int rank = wave_rank(key, n);
      const int n_ = n;
      SNF_PERMUTE_SETUP(rank, act)
      SNF_RT(1);   // rank + permute
      // everything below is in sorted order: lane r holds the r-th lead of the (read, ref_start) order
      const int s_fa = SNF_GATHER(fa);
      const int32_t s_rs = SNF_GATHER(ref_start), s_re = SNF_GATHER(ref_end), s_qs = SNF_GATHER(qry_start), s_qe = SNF_GATHER(qry_end);
      const int32_t s_svlen = SNF_GATHER(svlen), s_seq_len = SNF_GATHER(seq_len);
      const int64_t s_seq_off = SNF_GATHER(seq_off);
      const int s_strand = SNF_GATHER(strand);
      const uint32_t s_o = SNF_GATHER(o); const int s_lp = SNF_GATHER(lane);
      // neighbour r-1
      const int p_fa = __shfl_up(s_fa, 1, SNF_WAVE);
      const int32_t p_rs = __shfl_up(s_rs, 1, SNF_WAVE), p_re = __shfl_up(s_re, 1, SNF_WAVE);
      const int32_t p_qs = __shfl_up(s_qs, 1, SNF_WAVE), p_qe = __shfl_up(s_qe, 1, SNF_WAVE);
      const int p_strand = __shfl_up(s_strand, 1, SNF_WAVE);
      bool mg = false;
      if (act && lane > 0 && p_fa == s_fa) {
        mg = (thr == -1) ||
             (((iabs64((int64_t)s_rs - p_re) < thr || iabs64((int64_t)s_rs - p_rs) < thr) &&
               (iabs64((int64_t)s_qs - p_qe) < thr || iabs64((int64_t)s_qs - p_qs) < thr)) &&
              (p_strand == s_strand));  // == head strand: every member of a fused run shares it
      }
      const bool start = act && !mg;
      const unsigned long long smask = __ballot(start);
      // segment end for a start lane: next start - 1
      unsigned long long above = (lane < 63) ? (smask >> (lane + 1)) : 0ull;
      const int seg_end = above ? lane + __builtin_ctzll(above) : n - 1;
      const int64_t ps_svlen = wave_incl_scan64(act ? (int64_t)s_svlen : 0, lane);
      const int64_t ps_seq = wave_incl_scan64((act && s_seq_len >= 0) ? (int64_t)s_seq_len : 0, lane);
      const int32_t ps_has = wave_incl_scan((act && s_seq_len >= 0) ? 1 : 0, lane);
      const int64_t e_svlen = __shfl(ps_svlen, seg_end, SNF_WAVE), e_seq = __shfl(ps_seq, seg_end, SNF_WAVE);
      const int32_t e_has = __shfl(ps_has, seg_end, SNF_WAVE);
      const int64_t x_svlen = ps_svlen - s_svlen, x_seq = ps_seq - (s_seq_len >= 0 ? s_seq_len : 0);
      const int32_t x_has = ps_has - (s_seq_len >= 0 ? 1 : 0);
      const int nparts = seg_end - lane + 1;
      const int64_t tot_svlen = e_svlen - x_svlen, tot_seq = e_seq - x_seq;
      const bool seq_ok = (e_has - x_has) == nparts;
      // fused sequence: concatenation in the pool's fused region (curr_lead.seq += to_merge.seq)
      SNF_RT(2);   // gathers, fuse decisions, scans
      int64_t new_off = 0; bool need_copy = start && seq_ok && nparts > 1;
      if (__ballot(need_copy)) {  // one atomic per wave (same-address atomics serialise in L2): lanes take consecutive slices
        const int64_t mine = need_copy ? tot_seq : 0;
        const int64_t incl = wave_incl_scan64(mine, lane);
        const int64_t wave_total = __shfl(incl, 63, SNF_WAVE);
        int64_t base;
        if (slice_used + wave_total <= v.pool_slice) {        // (wave-uniform) this wave's own slice: no atomic
          base = v.pool_len + (int64_t)blockIdx.x * v.pool_slice + slice_used;
          slice_used += wave_total;
        } else {
          int64_t got = 0;
          if (lane == 0) got = (int64_t)atomicAdd(&v.cnt->pool_extra_used, (unsigned long long)wave_total);
          base = v.pool_extra_base + __shfl(got, 0, SNF_WAVE);
        }
        new_off = base + (incl - mine);
        if (need_copy && new_off + tot_seq > v.pool_cap) { atomicOr(&v.cnt->overflow, 1); need_copy = false; }
      }
      {  // the parts of the fused leads that are copied: every sorted lane knows its source, its length and - the start lane's slice + the
         // lengths of the parts before it in the same fused lead - its destination
        const unsigned long long upto = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
        const int my_start = (act && (smask & upto)) ? 63 - __builtin_clzll(smask & upto) : 0;
        const bool p_act = act && __shfl((int)need_copy, my_start, SNF_WAVE) != 0;
        const int64_t p_dst = __shfl(new_off, my_start, SNF_WAVE) + (x_seq - __shfl(x_seq, my_start, SNF_WAVE));
        wave_copy_parts(v, lane, p_act, s_seq_off, p_act ? s_seq_len : 0, p_dst);
      }
      const bool ok_seq = start && seq_ok && (nparts == 1 || need_copy);
      // compact the fused leads (start lanes) to lanes 0..m-1
      m = __builtin_popcountll(smask);
      const int kidx = __builtin_popcountll(smask & ((1ull << lane) - 1ull));
      __syncthreads();
      if (start) lds.perm[kidx] = lane;
      __syncthreads();
      const int src2 = lds.perm[lane < m ? lane : 0];
      f_orig = (int32_t)__shfl(s_o, src2, SNF_WAVE); f_lp = __shfl(s_lp, src2, SNF_WAVE);
      f_svlen = (int32_t)__shfl(tot_svlen, src2, SNF_WAVE);
      const int32_t t_seq_len = ok_seq ? (nparts == 1 ? s_seq_len : (int32_t)tot_seq) : -1;
      const int64_t t_seq_off = ok_seq ? (nparts == 1 ? s_seq_off : new_off) : 0;
      f_seq_len = __shfl(t_seq_len, src2, SNF_WAVE);
      f_seq_off = __shfl(t_seq_off, src2, SNF_WAVE);
      }
    }
    SNF_RT(3);   // pool reservation, byte copies, compaction
    const bool fact = lane < m;
    if (fact) {
      v.F_orig[lo + lane] = f_orig; v.F_svlen[lo + lane] = f_svlen; v.F_lpos[lo + lane] = lo + f_lp;
      v.F_seq_len[lo + lane] = f_seq_len; v.F_seq_off[lo + lane] = f_seq_off;
    }

    if (svtype == SNF_BND) {
      // ---- resplit_bnd: group by (mate_contig, is_first) in first-appearance order, chain 1-kb bins
      if (m <= 1 || cfg.dev_no_resplit) {
        if (fact) v.FI[lo + lane] = lo + lane;
        if (lane == 0) rc_emit(v, lo, m, (int32_t)c, true);
        continue;
      }
      const int thr = cfg.cluster_merge_bnd;
      int fa = -1;
      for (int i = 0; i < m; i++) {
        const int32_t mc = wave_bcast_i32(mate_contig, i); const int fi = wave_bcast_i32(is_first, i);
        if (fa < 0 && mc == mate_contig && fi == is_first) fa = i;
      }
      const int64_t pb = thr > 0 ? ((int64_t)mate_pos / thr) * thr : 0;
      uint64_t key = fact ? (((uint64_t)(uint32_t)fa << 48) | ((uint64_t)(uint32_t)((int64_t)pb + 0x80000000ll) << 8) | (uint32_t)lane) : ~0ull;
      const int rank = wave_rank(key, m);
      const int n_ = m;
      SNF_PERMUTE_SETUP(rank, fact)
      const int s_fa = SNF_GATHER(fa); const int64_t s_pb = SNF_GATHER(pb); const int s_j = SNF_GATHER(lane);
      const int p_fa = __shfl_up(s_fa, 1, SNF_WAVE); const int64_t p_pb = __shfl_up(s_pb, 1, SNF_WAVE);
      const bool brk = fact && (lane == 0 || p_fa != s_fa || (s_pb - p_pb > thr));
      const unsigned long long bmask = __ballot(brk);
      if (fact) v.FI[lo + lane] = lo + s_j;
      if (brk) {
        unsigned long long above = (lane < 63) ? (bmask >> (lane + 1)) : 0ull;
        const int end = above ? lane + 1 + __builtin_ctzll(above) : m;  // exclusive: position of the next chain start
        rc_emit(v, lo + lane, end - lane, (int32_t)c, false);
      }
      continue;
    }

    SNF_RT(4);   // F stores
    // ---- resplit on |svlen| bins of 20 (cluster.py:125-161)
    if (cfg.dev_no_resplit_repeat || cfg.dev_no_resplit) {
      if (fact) v.FI[lo + lane] = lo + lane;
      if (lane == 0) rc_emit(v, lo, m, (int32_t)c, true);
      continue;
    }
    {
      const int rb = cfg.cluster_resplit_binsize;
      const uint32_t av = f_svlen < 0 ? (uint32_t)(-(int64_t)f_svlen) : (uint32_t)f_svlen;      // (32-bit: a 64-bit division is ~150 instructions here)
      const int32_t bin = (int32_t)((av / (uint32_t)rb) * (uint32_t)rb);
      // one bin (wave-uniform test): the cluster stays whole, in its order
      if (__ballot(fact && bin != __builtin_amdgcn_readfirstlane(bin)) == 0ull) {
        if (fact) v.FI[lo + lane] = lo + lane;
        if (lane == 0) rc_emit(v, lo, m, (int32_t)c, true);
        continue;
      }
      uint64_t key = fact ? (((uint64_t)(uint32_t)bin << 8) | (uint32_t)lane) : ~0ull;
      const int rank = wave_rank(key, m);
      const int n_ = m;
      SNF_PERMUTE_SETUP(rank, fact)
      const int32_t s_bin = SNF_GATHER(bin); const int s_k = SNF_GATHER(lane);
      const int32_t p_bin = __shfl_up(s_bin, 1, SNF_WAVE);
      const bool sstart = fact && (lane == 0 || p_bin != s_bin);
      const unsigned long long smask = __ballot(sstart);
      const int nb = __builtin_popcountll(smask);
      const int sidx = __builtin_popcountll(smask & ((2ull << lane) - 1ull)) - 1;  // segment of this position
      __syncthreads();
      if (sstart) { lds.seg_start[sidx] = lane; lds.seg_key[sidx] = s_bin; }
      __syncthreads();
      // Sequential bin-merge state machine with the reference's index quirks.  Its arrays live ONE ELEMENT PER LANE (lane s: segment s,
      // lane t: surviving bin t) and the walk reads them with v_readlane at wave-uniform indices: a few cycles per access where the
      // LDS form (one lane walking arrays in LDS) paid a round trip of ~100 cycles for each of its ~50 dependent accesses.
      const int32_t K = lane < nb ? lds.seg_key[lane] : 0, ST = lane < nb ? lds.seg_start[lane] : m;
      int32_t NC = lane, HEAD = lane, TAIL = lane, NXT = -1, SEGOUT = 0, RCS = 0, RCL = 0;
      int cntc = nb, i = 1;
      while (cntc > 1 && i < cntc) {
        const int im1 = (i == 0) ? cntc - 1 : i - 1;  // Python negative index: new_clusters[-1]
        const int lb = wave_bcast_i32(NC, im1), cb = wave_bcast_i32(NC, i);
        const int64_t last = wave_bcast_i32(K, lb), curr = wave_bcast_i32(K, cb);
        const int64_t mn = curr < last ? curr : last;
        const double t = (double)mn * cfg.cluster_merge_len;
        const double thr = ((double)cfg.minsvlen >= t) ? (double)cfg.minsvlen : t;
        const int64_t diff = curr > last ? curr - last : last - curr;
        if ((double)diff <= thr) {
          const int tcb = wave_bcast_i32(TAIL, cb), hlb = wave_bcast_i32(HEAD, lb), tlb = wave_bcast_i32(TAIL, lb);
          if (lane == tcb) NXT = hlb;
          if (lane == cb) TAIL = tlb;
          const int32_t up = __shfl_down(NC, 1, SNF_WAVE);      // new_clusters.pop(im1)
          if (lane >= im1 && lane + 1 < cntc) NC = up;
          cntc--;
          i = (i - 2 > 0) ? i - 2 : 0;
        } else i++;
      }
      int outp = 0;
      for (int t2 = 0; t2 < cntc; t2++) {
        const int start = outp;
        for (int sg = wave_bcast_i32(HEAD, wave_bcast_i32(NC, t2)); sg >= 0; sg = wave_bcast_i32(NXT, sg)) {
          if (lane == sg) SEGOUT = outp;
          outp += (sg + 1 < nb ? wave_bcast_i32(ST, sg + 1) : m) - wave_bcast_i32(ST, sg);
        }
        if (lane == t2) { RCS = start; RCL = outp - start; }
      }
      const int32_t my_out = __shfl(SEGOUT, sidx < 0 ? 0 : sidx, SNF_WAVE), my_st = __shfl(ST, sidx < 0 ? 0 : sidx, SNF_WAVE);
      if (fact) v.FI[lo + my_out + (lane - my_st)] = lo + s_k;
      if (lane < cntc) rc_emit(v, lo + RCS, RCL, (int32_t)c, true);
      __syncthreads();
    }
  }
  SNF_RT_FLUSH();
}

}  // namespace snf
