// snf_stage_window.h - gfx950: the "window front end" of a pass.  It replaces the occupancy prefilter, the device-wide radix
// sort of the lead keys and the seven binning launches behind it (a1_keys, a0k_*, rocPRIM onesweep = 16 launches, a2k ... a7k:
// 26 launches, ~0.55 ms of a 1.9-ms pass) by six launches that never sort the whole table:
//
//   w1_hist     per lead: (task, svtype, 100-bp bin) -> its WINDOW (2^W consecutive bins of one (task, svtype)); one count per window
//   w2a / w2b   exclusive scan of the window counts (bucket offsets) + the list of occupied windows
//   w3_scatter  per lead: into its window's bucket (any order inside a bucket)
//   w4_local    one wave per occupied window: the bucket's leads ordered by (bin, arrival) in LDS (rank sort: a window holds tens of
//               leads), bin heads, per-bin record_lead side effects (leadprov.py:400-418: the 10-per-bin sequence cap, the hap
//               counters) and seed eligibility (cluster.py:262); per window the number of seeds / `leads` / `leads_long`
//   w5a / w5b   exclusive scan of those three counts over the occupied windows
//   w6_emit     one wave per occupied window: seeds, L / LL / packed lead records at their global places
//
// Reference semantics are those of snf_stage_cluster.h (stage A): a bin = (task, svtype, int(ref_start / 100)), leads of a bin in
// arrival order, `ld.seq = None` from the 11th lead of a bin on, hap counters per bin, a seed where a bin holds at least
// dev_min_leads_cluster leads with a length.  Everything behind it (b1k_seedmetrics onwards) reads the same arrays as before;
// the bin id of a seed (seed_bin) is the seed id itself here - only eligible bins get a row in bin_hap.
// Used when a one-lead bin can never seed a cluster (dev_min_leads_cluster >= 2: the condition of the old prefilter) and no window
// holds more leads than the largest instance of w4_local / w6_emit takes; otherwise the sort path (snf_stage_cluster.h) runs.
#pragma once
#include "snf_fused.h"
#include "snf_wave_refine.h"   // wave_incl_max (DPP scan)

namespace snf {

#define SNF_WIN_MAXCAP 1024

// packed attributes of a lead inside its window: bin inside the window (W bits, W <= 12), is_long (INS with svlen None), hap
SNF_HD uint32_t win_pack(const View& v, uint32_t bin_low, bool is_long, uint32_t hap) { return bin_low | ((is_long ? 1u : 0u) << 12) | (hap << 13); }
SNF_HD uint32_t win_bin_low(uint32_t a) { return a & 0xfffu; }
SNF_HD bool win_is_long(uint32_t a) { return (a >> 12) & 1u; }
SNF_HD uint32_t win_hap(uint32_t a) { return (a >> 13) & 3u; }
// flags added by w4_local (same word): lead of an eligible bin with a length / without one, Lead.seq dropped, first lead of an eligible bin
#define SNF_WF_NORM (1u << 16)
#define SNF_WF_LONG (1u << 17)
#define SNF_WF_SEQNULL (1u << 18)
#define SNF_WF_SEED (1u << 19)

// window of a lead; false: the lead lies outside its contig (dropped, leadprov.py:464-468)
SNF_D bool win_of_lead(const View& v, int64_t i, uint32_t* w, uint32_t* attr) {
  const int t = v.lead_task[i];
  const int64_t rs = v.in_ref_start[i];
  if (rs < 0 || rs >= v.t_contig_len[t]) return false;
  const uint64_t bin = (uint64_t)(rs / v.cfg.cluster_binsize);
  const int svtype = v.in_svtype[i];
  const int64_t w0 = v.t_win_off[t], nwin = (v.t_win_off[t + 1] - w0) / SNF_NTYPES;
  *w = (uint32_t)(w0 + (int64_t)svtype * nwin + (int64_t)(bin >> v.win_bits));
  *attr = win_pack(v, (uint32_t)(bin & ((1u << v.win_bits) - 1u)), svtype == SNF_INS && v.in_svlen[i] == SNF_SVLEN_NONE, v.in_hap[i]);
  return true;
}

// the lanes of a wave that hit the same window share one atomic (leads that arrive together lie in a handful of windows; thousands of
// same-address atomics would queue in L2 at ~25 ns each).  Returns the lane's rank among the lanes of its window and their number;
// *leader: this lane issues the atomic.
SNF_D int win_group(bool valid, uint32_t w, int* count, bool* leader) {
  const int lane = threadIdx.x & 63;
  unsigned long long todo = __ballot(valid);
  int rank = 0, cnt = 0; bool lead = false;
  while (todo) {
    const int l0 = __ffsll((long long)todo) - 1;
    const uint32_t w0 = (uint32_t)__shfl((int)w, l0, 64);
    const unsigned long long same = __ballot(valid && w == w0);
    if (valid && w == w0) { cnt = __popcll(same); rank = __popcll(same & ((1ull << lane) - 1ull)); lead = lane == l0; }
    todo &= ~same;
  }
  *count = cnt; *leader = lead;
  return rank;
}

// W1: window histogram; the window and the packed attributes of every lead are kept for w3_scatter (val_in / val_out)
__global__ void __launch_bounds__(256) w1_hist(const View v, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t w = ~0u, attr = 0;
  const bool valid = i < n && win_of_lead(v, i, &w, &attr);
  if (i < n) { v.val_in[i] = valid ? w : ~0u; v.val_out[i] = attr; }
  if (i <= v.NS) v.rcflag[i] = 0;      // (reset for d1w_refine: rc_emit sets it; N >= NS)
  int cnt; bool leader;
  win_group(valid, w, &cnt, &leader);
  if (leader) atomicAdd(&v.wcnt[w], (uint32_t)cnt);
}

// W0 (upload only): occupied windows and the largest window of the batch
__global__ void __launch_bounds__(256) w0_stats(const View v, int64_t n) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const uint32_t c = p < n ? v.wcnt[p] : 0u;
  const unsigned long long occ = __ballot(c != 0);
  uint32_t m = c;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { const uint32_t y = (uint32_t)__shfl_xor((int)m, d, 64); if (y > m) m = y; }
  if ((threadIdx.x & 63) == 0 && occ) {
    atomicAdd((unsigned long long*)&v.cnt->n_occ, (unsigned long long)__popcll(occ));
    atomicMax((unsigned long long*)&v.cnt->max_win, (unsigned long long)m);
  }
}

// W2: bucket offsets = exclusive scan of the window counts; occupied windows listed in ascending order
SNF_FUSED_HEAD(w2a_sums)
  const uint32_t c = p < n ? v.wcnt[p] : 0u;
  unsigned long long val[2] = {(unsigned long long)c, c ? 1ull : 0ull};
  tile_publish<2>(v, TS_WIN, val, lds);
}
SNF_FUSED_HEAD(w2b_offsets)
  const uint32_t c = p < n ? v.wcnt[p] : 0u;
  unsigned long long val[2] = {(unsigned long long)c, c ? 1ull : 0ull}, off[2];
  tile_scan<2>(v, TS_WIN, val, off, lds);
  if (p < n) {
    v.wcnt[p] = 0; v.wfill[p] = 0;      // spent / not yet used in this pass: both counters are clean for w3_scatter and the next pass
    v.wbase[p] = (uint32_t)off[0];
    if (c) { v.wlist[3 * off[1]] = (uint32_t)p; v.wlist[3 * off[1] + 1] = c; v.wlist[3 * off[1] + 2] = (uint32_t)off[0]; }
    if (p == n - 1) { v.wbase[n] = (uint32_t)(off[0] + val[0]); v.cnt->n_valid = (int64_t)(off[0] + val[0]); v.cnt->n_occ = (int64_t)(off[1] + val[1]); }
  }
}

SNF_CHAIN_HEAD(w2c_offsets, TS_WIN)      // (the pair above in one launch: snf_fused.h chain_scan)
  const uint32_t c = p < n ? v.wcnt[p] : 0u;
  unsigned long long val[2] = {(unsigned long long)c, c ? 1ull : 0ull}, off[2], tot[2];
  chain_scan<2>(v, TS_WIN, tile, val, off, tot, lds);
  if (p < n) {
    v.wcnt[p] = 0; v.wfill[p] = 0;      // spent / not yet used in this pass: both counters are clean for w3_scatter and the next pass
    v.wbase[p] = (uint32_t)off[0];
    if (c) { v.wlist[3 * off[1]] = (uint32_t)p; v.wlist[3 * off[1] + 1] = c; v.wlist[3 * off[1] + 2] = (uint32_t)off[0]; }
    if (p == n - 1) { v.wbase[n] = (uint32_t)tot[0]; v.cnt->n_valid = (int64_t)tot[0]; v.cnt->n_occ = (int64_t)tot[1]; }
  }
}

// W3: every lead into its window's bucket: (attributes << 32 | input index), any order inside the bucket
__global__ void __launch_bounds__(256) w3_scatter(const View v, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const uint32_t w = i < n ? v.val_in[i] : ~0u;
  const bool valid = w != ~0u;
  int cnt; bool leader;
  const int rank = win_group(valid, w, &cnt, &leader);
  uint32_t first = 0;
  if (leader) first = atomicAdd(&v.wfill[w], (uint32_t)cnt);
  // the leader's lane holds the reservation: every lane of the group reads it from there
  const int lane = threadIdx.x & 63;
  unsigned long long todo = __ballot(valid);
  uint32_t mine = 0;
  while (todo) {
    const int l0 = __ffsll((long long)todo) - 1;
    const uint32_t w0 = (uint32_t)__shfl((int)w, l0, 64);
    const uint32_t f0 = (uint32_t)__shfl((int)first, l0, 64);
    const unsigned long long same = __ballot(valid && w == w0);
    if (valid && w == w0) mine = f0;
    todo &= ~same;
  }
  (void)lane;
  if (valid) v.key_in[(int64_t)v.wbase[w] + mine + rank] = ((uint64_t)v.val_out[i] << 32) | (uint64_t)(uint32_t)i;
}

// (task, svtype, first bin) of window w
SNF_D void win_decode(const View& v, uint32_t w, int* grp, int64_t* bin0) {
  int lo = 0, hi = v.T;            // last task with t_win_off[t] <= w
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (v.t_win_off[mid] <= (int64_t)w) lo = mid; else hi = mid; }
  const int64_t w0 = v.t_win_off[lo], nwin = (v.t_win_off[lo + 1] - w0) / SNF_NTYPES;
  const int64_t rem = (int64_t)w - w0;
  *grp = lo * 8 + (int)(rem / nwin);
  *bin0 = (rem % nwin) << v.win_bits;
}

// inclusive running maximum along the positions of a window, 64 positions per round (carry: the maximum of the rounds before)
SNF_D int wave_runmax(int x, int carry) {
  x = wave_incl_max(x);      // (six DPP steps; the __shfl_up form was six round trips through the LDS crossbar)
  return x > carry ? x : carry;
}

// W4: one wave per occupied window.  CAP: leads a window may hold (the host picks the instance from the largest window of the batch)
template <int CAP>
__global__ void __launch_bounds__(64) w4_local(const View v, int64_t n_unused) {
  constexpr int E = CAP / 64;
  __shared__ uint64_t keys[CAP];
  __shared__ uint16_t hpos[CAP];
  __shared__ uint32_t sA[CAP], sB[CAP];      // per bin (at its head position): leads | leads with a length << 16;  hap 1 | hap 2 << 16
  const int lane = threadIdx.x;
  const int64_t k = blockIdx.x;
  const int n = (int)v.wlist[3 * k + 1];                     // {window, leads, bucket offset}: one record per occupied window (w2)
  const int64_t base = v.wlist[3 * k + 2];
  uint64_t e[E];
#pragma unroll
  for (int j = 0; j < E; j++) {
    const int p = lane + 64 * j;
    e[j] = p < n ? v.key_in[base + p] : ~0ull;
    if (p < n) { keys[p] = e[j]; sA[p] = 0; sB[p] = 0; }
  }
  __syncthreads();
  // rank sort by (bin, arrival): keys are distinct (the input index is part of them), a window holds tens of leads; every lane
  // reads the same LDS word per step (a broadcast)
  int r[E];
#pragma unroll
  for (int j = 0; j < E; j++) r[j] = 0;
  // (the attribute bits above the bin do not disturb the order: the comparison masks them)
  const uint64_t mask = ((uint64_t)0xfffu << 32) | 0xffffffffull;
  // (a window holds ~23 leads on a 30x genome and the instance is sized by the largest one: only the rounds that hold leads compare)
  if (n <= 64) {
    const uint64_t e0 = e[0] & mask;
    for (int q = 0; q < n; q++) r[0] += (keys[q] & mask) < e0 ? 1 : 0;
  } else {
    for (int q = 0; q < n; q++) {
      const uint64_t kq = keys[q] & mask;
#pragma unroll
      for (int j = 0; j < E; j++) if (64 * j < n) r[j] += kq < (e[j] & mask) ? 1 : 0;
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < E; j++) if (lane + 64 * j < n) keys[r[j]] = e[j];
  __syncthreads();
  // bin heads and the head position of every lead
  int carry = -1;
#pragma unroll
  for (int j = 0; j < E; j++) {
    const int p = lane + 64 * j;
    if (64 * j < n) {
      int h = -1;
      if (p < n) {
        const uint32_t b = win_bin_low((uint32_t)(keys[p] >> 32));
        if (p == 0 || win_bin_low((uint32_t)(keys[p - 1] >> 32)) != b) h = p;
      }
      const int m = wave_runmax(h, carry);
      if (p < n) hpos[p] = (uint16_t)m;
      carry = __shfl(m, 63, 64);
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < E; j++) {
    const int p = lane + 64 * j;
    if (64 * j >= n) break;
    if (p < n) {
      const uint32_t a = (uint32_t)(keys[p] >> 32);
      const int h = hpos[p];
      atomicAdd(&sA[h], 1u + (win_is_long(a) ? 0u : (1u << 16)));
      const uint32_t hp = win_hap(a);
      if (hp) atomicAdd(&sB[h], hp == 1 ? 1u : (1u << 16));
    }
  }
  __syncthreads();
  int n_seed = 0, n_norm = 0, n_long = 0;
#pragma unroll
  for (int j = 0; j < E; j++) {
    const int p = lane + 64 * j;
    if (64 * j >= n) break;
    bool f_seed = false, f_norm = false, f_long = false;
    if (p < n) {
      const uint64_t key = keys[p];
      uint32_t a = (uint32_t)(key >> 32);
      const int h = hpos[p];
      const uint32_t A = sA[h];
      const int all = (int)(A & 0xffffu), with_len = (int)(A >> 16);
      const bool elig = with_len >= v.cfg.dev_min_leads_cluster;
      f_norm = elig && !win_is_long(a); f_long = elig && win_is_long(a); f_seed = elig && p == h;
      if (f_norm) a |= SNF_WF_NORM;
      if (f_long) a |= SNF_WF_LONG;
      if (p - h + 1 > v.cfg.consensus_max_reads_bin) a |= SNF_WF_SEQNULL;      // leadprov.py:406-408 (counts every lead of the bin)
      if (f_seed) {
        a |= SNF_WF_SEED;
        const uint32_t B = sB[h];
        v.whead[base + p] = (uint64_t)with_len | ((uint64_t)all << 16) | ((uint64_t)(B & 0xffffu) << 32) | ((uint64_t)(B >> 16) << 48);
      }
      v.key_out[base + p] = ((uint64_t)a << 32) | (key & 0xffffffffull);
    }
    n_seed += __popcll(__ballot(f_seed)); n_norm += __popcll(__ballot(f_norm)); n_long += __popcll(__ballot(f_long));
  }
  if (lane == 0) { v.ws_seeds[k] = (uint32_t)n_seed; v.ws_nf[k] = (uint32_t)n_norm; v.ws_nl[k] = (uint32_t)n_long; }
}

// W5: exclusive scans of the three per-window counts (in place) + the totals of the stage
SNF_FUSED_HEAD(w5a_sums)
  unsigned long long val[3] = {p < n ? (unsigned long long)v.ws_seeds[p] : 0ull, p < n ? (unsigned long long)v.ws_nf[p] : 0ull,
                               p < n ? (unsigned long long)v.ws_nl[p] : 0ull};
  tile_publish<3>(v, TS_WINC, val, lds);
}
SNF_FUSED_HEAD(w5b_offsets)
  unsigned long long val[3] = {p < n ? (unsigned long long)v.ws_seeds[p] : 0ull, p < n ? (unsigned long long)v.ws_nf[p] : 0ull,
                               p < n ? (unsigned long long)v.ws_nl[p] : 0ull}, off[3];
  tile_scan<3>(v, TS_WINC, val, off, lds);
  if (p < n) {
    v.ws_seeds[p] = (uint32_t)off[0]; v.ws_nf[p] = (uint32_t)off[1]; v.ws_nl[p] = (uint32_t)off[2];
    if (p == n - 1) {
      const int64_t ns = (int64_t)(off[0] + val[0]);
      v.cnt->n_seeds = ns; v.cnt->n_bins = ns; v.cnt->NF = (int64_t)(off[1] + val[1]); v.cnt->NLL = (int64_t)(off[2] + val[2]);
      v.eligscan[v.NS] = (uint32_t)ns;
    }
  }
}

SNF_CHAIN_HEAD(w5c_offsets, TS_WINC)     // (the pair above in one launch)
  unsigned long long val[3] = {p < n ? (unsigned long long)v.ws_seeds[p] : 0ull, p < n ? (unsigned long long)v.ws_nf[p] : 0ull,
                               p < n ? (unsigned long long)v.ws_nl[p] : 0ull}, off[3], tot[3];
  chain_scan<3>(v, TS_WINC, tile, val, off, tot, lds);
  if (p < n) {
    v.ws_seeds[p] = (uint32_t)off[0]; v.ws_nf[p] = (uint32_t)off[1]; v.ws_nl[p] = (uint32_t)off[2];
    if (p == n - 1) {
      const int64_t ns = (int64_t)tot[0];
      v.cnt->n_seeds = ns; v.cnt->n_bins = ns; v.cnt->NF = (int64_t)tot[1]; v.cnt->NLL = (int64_t)tot[2];
      v.eligscan[v.NS] = (uint32_t)ns;
    }
  }
}

// W6: one wave per occupied window: its seeds, `leads` (L, packed records) and `leads_long` (LL) at their global places
template <int CAP>
__global__ void __launch_bounds__(64) w6_emit(const View v, int64_t n_unused) {
  constexpr int E = CAP / 64;
  const int lane = threadIdx.x;
  const int64_t k = blockIdx.x;
  const uint32_t w = v.wlist[3 * k];
  const int n = (int)v.wlist[3 * k + 1];
  const int64_t base = v.wlist[3 * k + 2];
  const int64_t S0 = v.ws_seeds[k], F0 = v.ws_nf[k], L0 = v.ws_nl[k];
  int grp; int64_t bin0;
  win_decode(v, w, &grp, &bin0);
  int cs = 0, cf = 0, cl = 0;      // seeds / leads / long leads of the rounds before
#pragma unroll
  for (int j = 0; j < E; j++) {
    const int p = lane + 64 * j;
    if (64 * j >= n) break;
    uint64_t key = 0; uint32_t a = 0;
    if (p < n) { key = v.key_out[base + p]; a = (uint32_t)(key >> 32); }
    const bool f_norm = (a & SNF_WF_NORM) != 0, f_long = (a & SNF_WF_LONG) != 0, f_seed = (a & SNF_WF_SEED) != 0;
    const unsigned long long bn = __ballot(f_norm), bl = __ballot(f_long), bs = __ballot(f_seed);
    const unsigned long long below = (1ull << lane) - 1ull;
    const int64_t qf = F0 + cf + __popcll(bn & below), ql = L0 + cl + __popcll(bl & below), qs = S0 + cs + __popcll(bs & below);
    const uint32_t o = (uint32_t)key;
    if (f_norm) {
      v.L[qf] = o;
      LeadRec rec = v.in_rec[o];
      if (rec.seq_len >= 0 && (a & SNF_WF_SEQNULL)) { rec.seq_len = -1; rec.seq_off = 0; }   // 11th+ lead of a bin: Lead.seq = None
      v.Lrec[qf] = rec;
    }
    if (f_long) v.LL[ql] = o;
    if (f_seed) {
      const uint64_t hd = v.whead[base + p];
      const int with_len = (int)(hd & 0xffffu), all = (int)((hd >> 16) & 0xffffu), h1 = (int)((hd >> 32) & 0xffffu), h2 = (int)(hd >> 48);
      v.seed_bin[qs] = (int32_t)qs;
      v.seed_lo[qs] = (int32_t)qf; v.seed_hi[qs] = (int32_t)(qf + with_len);
      v.seedL_lo[qs] = (int32_t)ql; v.seedL_hi[qs] = (int32_t)(ql + all - with_len);
      v.seed_start[qs] = (int32_t)((bin0 + win_bin_low(a)) * v.cfg.cluster_binsize);
      v.seed_grp[qs] = grp;
      v.bin_hap[3 * qs + 0] = (uint16_t)(all - h1 - h2); v.bin_hap[3 * qs + 1] = (uint16_t)h1; v.bin_hap[3 * qs + 2] = (uint16_t)h2;
    }
    cs += __popcll(bs); cf += __popcll(bn); cl += __popcll(bl);
  }
}

}  // namespace snf
