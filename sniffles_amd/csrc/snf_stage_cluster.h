// snf_stage_cluster.h - kernel bodies for lead binning, seed clusters and the adaptive merge scan.
//
// Reference semantics: LeadProvider.record_lead (leadprov.py:400-418), cluster.resolve seed stage
// (cluster.py:219-275), Cluster.compute_metrics (cluster.py:48-61), merge scan (cluster.py:278-308).
#pragma once
#include "snf_exact.h"
#include "snf_view.h"

namespace snf {

#define SNF_KEY_INVALID (~0ull)
SNF_HD int key_grp(uint64_t key) { return (int)(key >> 32); }  // task*8 + svtype
SNF_HD int grp_svtype(int g) { return g & 7; }
SNF_HD int grp_task(int g) { return g >> 3; }

SNF_HD bool lead_is_long(const View& v, uint32_t o) {  // INS lead with svlen None -> leads_long (cluster.py:248-250)
  return v.in_svtype[o] == SNF_INS && v.in_svlen[o] == SNF_SVLEN_NONE;
}

// ------------------------------------------------------------------------------------------ stage A
// A1: sort key (task, svtype, bin); leads outside the task region are dropped (leadprov.py:464-468)
// With the occupancy prefilter (View::prefilter) the key goes to pf_key and the lead marks its (task, svtype, bin) cell:
// bit 0 "a lead was here", bit 1 "a second one was" - a0_keep / a0_emit then hand only the leads of cells with two or
// more leads to the sort, in arrival order.  Leads that arrive together are close on the genome (BAM order), so the
// marks of a wave fall into a handful of cache lines.
SNF_HD int64_t pf_cell(const View& v, int t, int svtype, uint64_t bin) {
  const int64_t c0 = v.t_cell_off[t], nb = (v.t_cell_off[t + 1] - c0) / SNF_NTYPES;
  return c0 + (int64_t)svtype * nb + (int64_t)bin;
}
// Leads that arrive together lie within a read length of each other, i.e. in a few hundred neighbouring bins: packed 16 cells
// to a word their marks hit the same two or three cache lines.  Experiment (SNF_PF_SPREAD=1, off by default): pf_slot moves
// the low five bits of the cell index to the top of a 2^19-cell block, so that neighbouring cells end up 4 KB apart (another L2
// channel each).  Measured on the 30x genome: the marking kernel gets SLOWER (0.146 ms against 0.111) - the adjacent marks are
// not queueing on a channel, they share cache lines that the spread version has to fetch one by one.
SNF_HD int64_t pf_slot(const View& v, int64_t cell) {
  if (!v.pf_spread) return cell;
  const int64_t in = cell & (((int64_t)1 << 19) - 1);
  return (cell - in) | ((in & 31) << 14) | (in >> 5);
}
SNF_HD void a1_keys_body(int64_t i, const View& v) {
  int t = v.lead_task[i];
  int64_t rs = v.in_ref_start[i];
  bool valid = rs >= 0 && rs < v.t_contig_len[t];
  uint64_t bin = valid ? (uint64_t)(rs / v.cfg.cluster_binsize) : 0;
  const int svtype = v.in_svtype[i];
  const uint64_t k = valid ? (((uint64_t)(t * 8 + svtype)) << v.key_bin_bits | bin) : (1ull << v.key_nbits);
  if (v.prefilter) {
    if (v.key32) ((uint32_t*)v.pf_key)[i] = (uint32_t)k; else v.pf_key[i] = k;
    if (valid) {
      const int64_t cell = pf_slot(v, pf_cell(v, t, svtype, bin));
      uint32_t* w = v.pf_bm + (cell >> 4);
      const int sh = (int)(cell & 15) * 2;
      const uint32_t old = atomic_fetch_or_u32(w, 1u << sh);
      if (((old >> sh) & 3u) == 1u) atomic_fetch_or_u32(w, 2u << sh);
    }
    return;
  }
  if (v.key32) ((uint32_t*)v.key_in)[i] = (uint32_t)k; else v.key_in[i] = k;
  v.val_in[i] = (uint32_t)i;
  v.seqnull[i] = 0;
}
// A0 (prefilter): keep flag per input lead | (scan) | keys and indices of the kept leads, compacted in arrival order; the
// marks of this pass are cleared again (plain stores: nobody reads the bitmap after a0_keep)
SNF_HD uint64_t pf_key_of(const View& v, int64_t i) { return v.key32 ? (uint64_t)((const uint32_t*)v.pf_key)[i] : v.pf_key[i]; }
SNF_HD bool pf_cell_of_key(const View& v, uint64_t k, int64_t* cell) {
  if (k >> v.key_nbits) return false;
  const uint64_t g = k >> v.key_bin_bits, bin = k & ((1ull << v.key_bin_bits) - 1ull);
  *cell = pf_slot(v, pf_cell(v, (int)(g >> 3), (int)(g & 7), bin));
  return true;
}
SNF_HD void a0_keep_body(int64_t i, const View& v) {
  int64_t cell;
  uint32_t keep = 0;
  if (pf_cell_of_key(v, pf_key_of(v, i), &cell)) keep = (v.pf_bm[cell >> 4] >> ((int)(cell & 15) * 2 + 1)) & 1u;
  v.pf_keep[i] = keep;
  if (i == 0) v.pf_keep[v.N] = 0;
}
SNF_HD void a0_emit(int64_t i, const View& v) {
  const uint64_t k = pf_key_of(v, i);
  int64_t cell;
  if (pf_cell_of_key(v, k, &cell)) v.pf_bm[cell >> 4] = 0;
  if (!v.pf_keep[i]) return;
  const uint32_t q = v.pf_scan[i];
  if (v.key32) ((uint32_t*)v.key_in)[q] = (uint32_t)k; else v.key_in[q] = k;
  v.val_in[q] = (uint32_t)i;
  v.seqnull[i] = 0;
}
SNF_HD void a0_compact_body(int64_t i, const View& v) {
  if (i == 0) v.cnt->n_kept = v.pf_scan[v.N];
  a0_emit(i, v);
}

// sorted key at position p in the canonical form grp << 32 | bin (SNF_KEY_INVALID for leads outside their contig)
SNF_HD uint64_t sorted_key(const View& v, int64_t p) {
  const uint64_t k = v.key32 ? (uint64_t)((const uint32_t*)v.key_out)[p] : v.key_out[p];
  if (k >> v.key_nbits) return SNF_KEY_INVALID;
  return ((k >> v.key_bin_bits) << 32) | (k & ((1ull << v.key_bin_bits) - 1ull));
}

// A2: bin heads in (stably) sorted order
SNF_HD void a2_heads_body(int64_t p, const View& v) {
  uint64_t k = sorted_key(v, p);
  bool valid = k != SNF_KEY_INVALID;
  v.headflag[p] = (valid && (p == 0 || sorted_key(v, p - 1) != k)) ? 1u : 0u;
  if (valid && (p + 1 == v.NS || sorted_key(v, p + 1) == SNF_KEY_INVALID)) v.cnt->n_valid = p + 1;
  if (p == 0) v.headflag[v.NS] = 0;
}

// A3: bin table (headscan = exclusive scan of headflag; headscan[N] = #bins)
SNF_HD void a3_emit(int64_t p, const View& v) {
  if (v.headflag[p]) {
    uint32_t b = v.headscan[p];
    v.bin_lo[b] = (int32_t)p;
    v.bin_key[b] = sorted_key(v, p);
  }
}
SNF_HD void a3_bins_body(int64_t p, const View& v) {
  if (p == 0) {
    int64_t nb = v.headscan[v.NS];
    v.cnt->n_bins = nb;
    v.bin_lo[nb] = (int32_t)v.cnt->n_valid;
  }
  a3_emit(p, v);
}

// A4: per bin: record_lead side effects (seq cap, hap counters) and seed eligibility (cluster.py:262)
SNF_HD void a4_binstats_body(int64_t b, const View& v) {
  if (b >= v.cnt->n_bins) { v.eligflag[b] = 0; return; }
  int32_t lo = v.bin_lo[b], hi = v.bin_lo[b + 1];
  int n_normal = 0;
  uint32_t hc[3] = {0, 0, 0};
  for (int32_t p = lo; p < hi; p++) {
    uint32_t o = v.val_out[p];
    if (p - lo + 1 > v.cfg.consensus_max_reads_bin) v.seqnull[o] = 1;
    uint8_t h = v.in_hap[o];
    if (hc[h] < 65535u) hc[h]++;  // array('H') OverflowError is swallowed by the reference
    if (!lead_is_long(v, o)) n_normal++;
  }
  for (int h = 0; h < 3; h++) v.bin_hap[3 * b + h] = (uint16_t)hc[h];
  bool el = n_normal >= v.cfg.dev_min_leads_cluster;
  v.bin_elig[b] = el;
  v.eligflag[b] = el ? 1u : 0u;
  int g = key_grp(v.bin_key[b]);
  if (b == 0 || key_grp(v.bin_key[b - 1]) != g) v.grp_first_bin[g] = (int32_t)b;
}

// A5: per sorted lead: member of a seed's `leads` (fN) or `leads_long` (fL)
SNF_HD void a5_leadflags_body(int64_t p, const View& v) {
  uint32_t fn = 0, fl = 0;
  if (p < v.cnt->n_valid) {
    uint32_t b = v.headscan[p + 1] - 1;
    if (v.bin_elig[b]) {
      if (lead_is_long(v, v.val_out[p])) fl = 1; else fn = 1;
    }
  }
  v.fN[p] = fn;
  v.fL[p] = fl;
  if (p == 0) { v.fN[v.NS] = 0; v.fL[v.NS] = 0; v.eligflag[v.NS] = 0; }
}

// A6: scatter into L / LL (pN/pL = exclusive scans of fN/fL)
SNF_HD void a6_emit(int64_t p, const View& v);
SNF_HD void a6_scatter_body(int64_t p, const View& v) {
  if (p == 0) { v.cnt->NF = v.pN[v.NS]; v.cnt->NLL = v.pL[v.NS]; v.cnt->n_seeds = v.eligscan[v.NS]; }
  a6_emit(p, v);
}
SNF_HD void a6_emit(int64_t p, const View& v) {
  if (p >= v.cnt->n_valid) return;
  uint32_t o = v.val_out[p];
  if (v.fN[p]) {
    const uint32_t q = v.pN[p];
    v.L[q] = o;
    LeadRec r = v.in_rec[o];
    if (r.seq_len >= 0 && v.seqnull[o]) { r.seq_len = -1; r.seq_off = 0; }  // 11th+ lead of a bin: Lead.seq = None
    v.Lrec[q] = r;
  }
  if (v.fL[p]) v.LL[v.pL[p]] = o;
}

// A7: seed table (eligscan = exclusive scan of eligflag)
SNF_HD void a7_seeds_body(int64_t b, const View& v) {
  if (b >= v.cnt->n_bins || !v.bin_elig[b]) return;
  uint32_t s = v.eligscan[b];
  int32_t lo = v.bin_lo[b], hi = v.bin_lo[b + 1];
  v.seed_bin[s] = (int32_t)b;
  v.seed_lo[s] = (int32_t)v.pN[lo];
  v.seed_hi[s] = (int32_t)v.pN[hi];
  v.seedL_lo[s] = (int32_t)v.pL[lo];
  v.seedL_hi[s] = (int32_t)v.pL[hi];
  v.seed_start[s] = (int32_t)((uint32_t)v.bin_key[b]) * v.cfg.cluster_binsize;
  v.seed_grp[s] = key_grp(v.bin_key[b]);
}

// ------------------------------------------------------------------------------------------ stage B
// Cluster.compute_metrics over L[lo,hi) (cluster.py:48-61): n = min(len,100) samples every int(len/n),
// mean divides by n (not the sample count), stdev = statistics.stdev of the sampled ref_starts
// sums of compute_metrics in 128-bit integers (positions further apart than 2^27: never inside one contig's cluster in practice)
SNF_HD void compute_metrics_wide(const View& v, int32_t lo, int64_t len, int64_t step, i128* S1, u128* S2) {
  const int64_t x0 = v.Lrec[lo].ref_start;
  i128 a = 0; u128 b = 0;
  for (int64_t i = 0; i < len; i += step) {
    const int64_t d = (int64_t)v.Lrec[lo + i].ref_start - x0;
    a += d; b += (u128)((i128)d * d);
  }
  *S1 = a; *S2 = b;
}
SNF_HD void compute_metrics(const View& v, int32_t lo, int32_t hi, double* mean, double* sd, ClusterSums* out = nullptr) {
  int64_t len = hi - lo;
  int64_t n = len < 100 ? len : 100;
  if (out) { out->sum = 0; out->s1 = 0; out->s2 = ~0ull; out->x0 = 0; out->hi = hi; }
  if (n == 0) { *mean = 0; *sd = 0; return; }
  if (n == 1) {
    const LeadRec& r = v.Lrec[lo];
    *mean = (double)r.svlen; *sd = 0;
    if (out) { out->sum = r.svlen; out->s2 = 0; out->x0 = r.ref_start; }
    return;
  }
  int64_t step = len / n;
  int64_t sum = 0, cnt = 0;
  // The sums of the (at most 199) sampled deviations d = ref_start - x0 and of their squares stay in 64 bits while |d| < 2^27
  // (d^2 < 2^54): 128-bit arithmetic per element cost ~15 instructions, and c1_mergeruns executes this loop for a whole wave
  // whenever one of its 64 runs merges (a lone wave, every instruction at full latency: the loop was most of that kernel).
  int64_t S1 = 0; uint64_t S2 = 0, wide = 0;
  int64_t x0 = 0;     // (the first record's ref_start: arrives with the first batch)
  // MCH records are requested before the first is used (a loop that waits for every record costs a round trip per lead).  A lane
  // whose range does not reach that far asks for record 0 of the table - the same line for every such lane of the wave, one access
  // for all of them (clamped to its own range, a seed of six leads sent 32 requests of its own per batch through the L1).
  constexpr int MCH = 32;
  for (int64_t i = 0; i < len; i += MCH * step) {
    int32_t sv[MCH], rs[MCH];
#pragma unroll
    for (int u = 0; u < MCH; u++) {
      const int64_t q = i + u * step;
      const LeadRec& r = v.Lrec[q < len ? lo + q : 0];
      sv[u] = r.svlen; rs[u] = r.ref_start;
    }
    // every request of the batch is issued before the first value is used: left alone the compiler sinks each load to its use
    // (fewer live registers) and the batch becomes 32 dependent round trips (seen in the ISA; b1k_seedmetrics 32.5 -> 26.0 us)
    asm volatile("" ::: "memory");
    if (i == 0) x0 = rs[0];
#pragma unroll
    for (int u = 0; u < MCH; u++) {
      if (i + u * step >= len) break;
      sum += sv[u];
      const int64_t d = (int64_t)rs[u] - x0;
      S1 += d; S2 += (uint64_t)(d * d);
      wide |= (uint64_t)(d < 0 ? -d : d) >> 27;
      cnt++;
    }
  }
  *mean = (double)sum / (double)n;
  if (out && step == 1 && !wide) { out->sum = sum; out->s1 = S1; out->s2 = S2; out->x0 = (int32_t)x0; }     // (every lead was read: the sums add up under merges)
  if (!wide) { *sd = stdev_from_sums(cnt, (i128)S1, (u128)S2); return; }
  i128 W1; u128 W2;
  compute_metrics_wide(v, lo, len, step, &W1, &W2);
  *sd = stdev_from_sums(cnt, W1, W2);
}

SNF_HD bool seed_first_of_group(const View& v, int64_t s) { return s == 0 || v.seed_grp[s - 1] != v.seed_grp[s]; }

SNF_HD void b1_seedmetrics_body(int64_t s, const View& v) {
  if (s == 0) v.runflag[v.NS] = 0;
  if (s >= v.cnt->n_seeds) { v.runflag[s] = 0; v.clflag[s] = 0; return; }
  double mean, sd;
  ClusterSums cs;
  compute_metrics(v, v.seed_lo[s], v.seed_hi[s], &mean, &sd, &cs);
  v.c_ms[s] = cs;
  int g = v.seed_grp[s], t = grp_task(g);
  int32_t seed = v.seed_start[s];
  bool within_tr = false;
  if (v.t_has_tr[t]) {
    // monotone sweep of cluster.py:240-246 == first TR whose end >= seed (else the last TR)
    int64_t lo = v.t_tr_off[t], hi = v.t_tr_off[t + 1];
    int64_t k = lower_bound_i32(v.tr_pmax, lo, hi, seed);
    if (k >= hi) k = hi - 1;
    within_tr = v.tr_start[k] < seed && seed < v.tr_end[k];
  }
  uint8_t rep = (within_tr || v.cfg.repeat) ? 1 : 0;
  v.s_mean0[s] = mean; v.s_stdev0[s] = sd; v.s_repeat0[s] = rep;
  v.c_mean[s] = mean; v.c_stdev[s] = sd; v.c_repeat[s] = rep;
  v.c_last[s] = (int32_t)s;
  v.c_end[s] = seed + v.cfg.cluster_binsize;
  bool first = seed_first_of_group(v, s);
  bool last = (s + 1 == v.cnt->n_seeds) || v.seed_grp[s + 1] != g;
  v.prv[s] = first ? -1 : (int32_t)(s - 1);
  v.nxt[s] = last ? -1 : (int32_t)(s + 1);
  v.clflag[s] = 1;
  if (first) v.grp_seed_lo[g] = (int32_t)s;
  if (last) v.grp_seed_hi[g] = (int32_t)(s + 1);
  // run cut: a gap no merge criterion can bridge unless a cluster's start-stdev is huge (validated later)
  bool cut = first;
  if (!first && v.run_gap >= 0) {
    int64_t inner = (int64_t)seed - ((int64_t)v.seed_start[s - 1] + v.cfg.cluster_binsize);
    cut = inner > v.run_gap;
  }
  v.runflag[s] = cut ? 1u : 0u;
}

SNF_HD void b2_emit(int64_t s, const View& v) {
  if (s < v.cnt->n_seeds && v.runflag[s]) v.run_first[v.runscan[s]] = (int32_t)s;
}
SNF_HD void b2_runs_body(int64_t s, const View& v) {
#ifdef SNF_C1_PROFILE
  if (s == 0) for (int k = 0; k < 16; k++) v.cnt->c1p[k] = 0;
#endif
  if (s == 0) {
    int64_t nr = v.runscan[v.NS];
    v.cnt->n_runs = nr;
    v.run_first[nr] = (int32_t)v.cnt->n_seeds;
  }
  b2_emit(s, v);
}

// ------------------------------------------------------------------------------------------ stage C
SNF_HD bool merge_criterion(const View& v, int svtype, int64_t inner, int64_t outer, double sd_a, double sd_b,
                            double mean_a, double mean_b, bool rep_a, bool rep_b) {
  double ms = sd_a < sd_b ? sd_a : sd_b;
  bool merge = (double)inner <= ms * v.cfg.cluster_r;
  if (!merge && (v.cfg.repeat || rep_a || rep_b)) {
    double h = (fabs(mean_a) + fabs(mean_b)) * v.cfg.cluster_repeat_h;
    double lim = h < v.cfg.cluster_repeat_h_max ? h : v.cfg.cluster_repeat_h_max;
    merge = (double)outer <= lim;
  }
  if (!merge && svtype == SNF_BND) merge = inner <= v.cfg.cluster_merge_bnd;
  return merge;
}

// Sequential adaptive merge scan (cluster.py:278-308) over the seeds [s0, s_end) of one group, as a
// linked list.  `group_first`: s0 is the first cluster of its (task, svtype) sequence, i.e. the scan's
// index rule `i = max(0, i-2) + 1` clamps at list index 0.  For a run that starts later the cluster
// before s0 never merges with it (validated by c2_validate), so after a merge at the run's first
// cluster the scan steps back to that boundary, fails, and returns: the position stays put.
// The walk is a chain of dependent loads (thread per run; the longest run of a batch is the duration of c1_mergeruns): a node's
// fields - its link included - are fetched together, one round trip, and a node that becomes the current one is carried over in
// registers instead of being read again (the SoA form `nxt[cur]`, then the fields of that node, cost two round trips per step).
struct MergeNode { int32_t start, end, nxt, prv, last, lo; double sd, mean; uint8_t rep; ClusterSums cs; };
SNF_HD MergeNode merge_node(const View& v, int32_t s) {
  MergeNode n;
  n.start = v.seed_start[s]; n.end = v.c_end[s]; n.nxt = v.nxt[s]; n.prv = v.prv[s]; n.last = v.c_last[s]; n.lo = v.seed_lo[s];
  n.sd = v.c_stdev[s]; n.mean = v.c_mean[s]; n.rep = v.c_repeat[s]; n.cs = v.c_ms[s];
  return n;
}
#if defined(SNF_C1_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
#define C1P_DECL unsigned long long c1t = wall_clock64(), c1a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define C1P(k) do { const unsigned long long n_ = wall_clock64(); c1a[k] += n_ - c1t; c1t = n_; } while (0)
#define C1P_FLUSH() do { if ((threadIdx.x & 63) == __builtin_ctzll(__ballot(1))) { for (int k_ = 0; k_ < 8; k_++) atomicAdd(&v.cnt->c1p[k_], c1a[k_]); atomicAdd(&v.cnt->c1p[8], 1ull); \
    unsigned long long tot_ = 0; for (int k_ = 0; k_ < 8; k_++) tot_ += c1a[k_]; atomicMax(&v.cnt->c1p[9], tot_); } } while (0)
#else
#define C1P_DECL
#define C1P(k) do { } while (0)
#define C1P_FLUSH() do { } while (0)
#endif
SNF_HD void merge_walk(const View& v, int32_t s0, int32_t s_end, bool group_first, int64_t run) {
  C1P_DECL
  int svtype = grp_svtype(v.seed_grp[s0]);
  int32_t cur = s0;
  int64_t idx = 0;
  MergeNode A = merge_node(v, s0);
  double b_sd = A.sd, b_am = fabs(A.mean);
  uint8_t b_rep = A.rep;
  C1P(0);   // group + first node
  for (;;) {
    const int32_t nx = A.nxt;
    if (nx < 0 || nx >= s_end) break;
    const MergeNode B = merge_node(v, nx);
    C1P(1);   // next node
    int64_t inner = (int64_t)B.start - A.end;
    int64_t outer = (int64_t)B.end - A.start;
    bool merge = merge_criterion(v, svtype, inner, outer, A.sd, B.sd, A.mean, B.mean, A.rep, B.rep);
    C1P(2);   // criterion
    if (merge) {
      int32_t nn = B.nxt;
      const int32_t last = B.last;
      v.c_last[cur] = last; A.last = last;
      v.c_end[cur] = B.end; A.end = B.end;
      A.rep = A.rep | B.rep; v.c_repeat[cur] = A.rep;
      v.nxt[cur] = nn; A.nxt = nn;
      if (nn >= 0) v.prv[nn] = cur;
      v.clflag[nx] = 0;
      double mean, sd;
      C1P(3);   // merge: stores
      // metrics of the merged cluster (cluster.py:300: compute_metrics over its leads).  Below 200 leads the reference reads EVERY
      // lead (step = len // min(len, 100) = 1), so the sums of the two parts add up - the part that joins re-based from its own
      // first position to this cluster's (exact integer arithmetic) - and no lead is read again; otherwise the leads are read
      const int32_t hi = B.cs.hi;
      const int64_t Lm = (int64_t)hi - A.lo;
      bool added = false;
      if (!v.merge_reread && Lm < 200 && A.cs.s2 != ~0ull && B.cs.s2 != ~0ull && A.cs.hi == B.lo) {
        const int64_t nb = (int64_t)hi - B.lo, dx = (int64_t)B.cs.x0 - A.cs.x0;
        const i128 S1m = (i128)A.cs.s1 + B.cs.s1 + (i128)nb * dx;
        const i128 S2m = (i128)A.cs.s2 + (i128)B.cs.s2 + (i128)2 * dx * B.cs.s1 + (i128)nb * dx * dx;
        const int64_t sum = A.cs.sum + B.cs.sum;
        const int64_t nn_ = Lm < 100 ? Lm : 100;
        mean = (double)sum / (double)nn_;
        sd = stdev_from_sums(Lm, S1m, (u128)S2m);
        const bool fits = S2m < ((i128)1 << 62) && S1m < ((i128)1 << 62) && S1m > -((i128)1 << 62);
        A.cs.sum = sum; A.cs.s1 = (int64_t)S1m; A.cs.s2 = fits ? (uint64_t)S2m : ~0ull;
        added = true;
      }
      if (!added) { compute_metrics(v, A.lo, hi, &mean, &sd); A.cs.s2 = ~0ull; }
      A.cs.hi = hi;
      v.c_ms[cur] = A.cs;
      C1P(4);   // merge: metrics
      v.c_mean[cur] = mean; v.c_stdev[cur] = sd; A.mean = mean; A.sd = sd;
      if (cur == s0) {
        if (sd > b_sd) b_sd = sd;
        if (fabs(mean) > b_am) b_am = fabs(mean);
        b_rep |= A.rep;
      }
      if (group_first) {
        if (idx == 0) {          // i = max(0,-2)+1 = 1: the merged cluster 0 is not re-checked
          int32_t n2 = A.nxt;
          if (n2 < 0 || n2 >= s_end) break;
          cur = n2; idx = 1; A = merge_node(v, cur);
        } else if (idx == 1) {   // i = max(0,-1)+1 = 1: stay
        } else { cur = A.prv; idx--; A = merge_node(v, cur); }
      } else {
        if (cur != s0) { cur = A.prv; A = merge_node(v, cur); }
      }
      C1P(5);   // merge: movement
    } else {
      cur = nx; idx++; A = B;
    }
  }
  C1P(6);
  if (run >= 0) {
    int32_t last = cur, nx = A.nxt;     // (A is the node of `cur` wherever the walk stops)
    while (!(nx < 0 || nx >= s_end)) { last = nx; nx = v.nxt[last]; }
    v.run_last_head[run] = last;
    v.run_b_stdev[run] = b_sd; v.run_b_absmean[run] = b_am; v.run_b_repeat[run] = b_rep;
  }
  C1P(7);   // tail
  C1P_FLUSH();
}

SNF_HD void c1_mergeruns_body(int64_t r, const View& v) {
  if (r >= v.cnt->n_runs) return;
  int32_t s0 = v.run_first[r], s_end = v.run_first[r + 1];
  merge_walk(v, s0, s_end, seed_first_of_group(v, s0), r);
}

// C2: a run boundary is valid iff the last cluster of the previous run can merge with NO state the first
// cluster of this run went through (conservative superset of the comparisons the serial scan makes).
SNF_HD void c2_validate_body(int64_t r, const View& v) {
  if (r >= v.cnt->n_runs) return;
  int32_t s0 = v.run_first[r];
  if (seed_first_of_group(v, s0)) return;
  int32_t a = v.run_last_head[r - 1];
  int g = v.seed_grp[s0];
  int64_t inner = (int64_t)v.seed_start[s0] - v.c_end[a];
  int64_t outer = (int64_t)v.seed_start[s0] + v.cfg.cluster_binsize - v.seed_start[a];  // smallest possible
  bool bad = merge_criterion(v, grp_svtype(g), inner, outer, v.c_stdev[a], v.run_b_stdev[r], v.c_mean[a],
                             v.run_b_absmean[r], v.c_repeat[a], v.run_b_repeat[r]);
  if (bad) atomic_or_i32(&v.grp_dirty[g], 1);
}

// C3: exact serial redo of a whole (task, svtype) sequence whose run cuts were not provably safe
SNF_HD void c3_serial_body(int64_t g, const View& v) {
  if (!v.grp_dirty[g]) return;
  int32_t lo = v.grp_seed_lo[g], hi = v.grp_seed_hi[g];
  if (lo < 0) return;
  for (int32_t s = lo; s < hi; s++) {
    v.c_mean[s] = v.s_mean0[s]; v.c_stdev[s] = v.s_stdev0[s]; v.c_repeat[s] = v.s_repeat0[s];
    v.c_last[s] = s; v.c_end[s] = v.seed_start[s] + v.cfg.cluster_binsize;
    { ClusterSums cs = v.c_ms[s]; cs.s2 = ~0ull; cs.hi = v.seed_hi[s]; v.c_ms[s] = cs; }     // (the serial redo reads the leads at every merge)
    v.prv[s] = s == lo ? -1 : s - 1; v.nxt[s] = s + 1 == hi ? -1 : s + 1;
    v.clflag[s] = 1;
  }
  merge_walk(v, lo, hi, true, -1);
  atomic_add_u64((unsigned long long*)&v.cnt->n_dirty_groups, 1ull);
}

// C4: merged cluster table (clscan = exclusive scan of clflag)
SNF_HD void c4_emit(int64_t s, const View& v);
SNF_HD void c4_clusters_body(int64_t s, const View& v) {
  if (s == 0) v.cnt->n_clusters = v.clscan[v.NS];
  c4_emit(s, v);
}
SNF_HD void c4_emit(int64_t s, const View& v) {
  if (s < v.cnt->n_seeds && v.clflag[s]) {
    const uint32_t c = v.clscan[s];
    v.cl_head[c] = (int32_t)s;
    ClusterHdr hd;
    hd.h = (int32_t)s; hd.lo = v.seed_lo[s]; hd.n = v.seed_hi[v.c_last[s]] - hd.lo; hd.grp = v.seed_grp[s]; hd.repeat = v.c_repeat[s];
    hd._pad[0] = hd._pad[1] = hd._pad[2] = 0;
    v.chdr[c] = hd;
  }
}

}  // namespace snf
