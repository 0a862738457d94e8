// snf_stage_window.h - gfx950: the "window front end" of a pass.  It replaces the occupancy prefilter, the device-wide radix
// sort of the lead keys and the seven binning launches behind it (a1_keys, a0k_*, rocPRIM onesweep = 16 launches, a2k ... a7k:
// 26 launches, ~0.55 ms of a 1.9-ms pass) by six launches that never sort the whole table:
//
//   w1_hist     per lead: (task, svtype, 100-bp bin) -> its WINDOW (2^W consecutive bins of one (task, svtype)); one count per window
//   w2a / w2b   exclusive scan of the window counts (bucket offsets) + the list of occupied windows
//   w3_scatter  per lead: into its window's bucket (any order inside a bucket)
//   w4s_segment one wave per 64 positions of the bucket array = the handful of windows that start there (a large window: the whole
//               wave, in rounds): every window's leads ordered by (bin, arrival) in LDS (rank sort inside the window), bin heads,
//               per-bin record_lead side effects (leadprov.py:400-418: the 10-per-bin sequence cap, the hap counters) and seed
//               eligibility (cluster.py:262); per wave the number of seeds / `leads` / `leads_long`, per lead its rank among them
//   w5a / w5b   exclusive scan of those three counts over the waves
//   w6t_emit    one thread per bucket position: seeds, L / LL / packed lead records at their global places (offset of the owning
//               wave + rank)
// (Rounds 4's w4_local / w6_emit gave every occupied window a wave of its own: 122 k waves for windows of ~23 leads of which six
// are kept - 36 % and 9 % of the lanes had a lead.)
//
// Reference semantics are those of snf_stage_cluster.h (stage A): a bin = (task, svtype, int(ref_start / 100)), leads of a bin in
// arrival order, `ld.seq = None` from the 11th lead of a bin on, hap counters per bin, a seed where a bin holds at least
// dev_min_leads_cluster leads with a length.  Everything behind it (b1k_seedmetrics onwards) reads the same arrays as before;
// the bin id of a seed (seed_bin) is the seed id itself here - only eligible bins get a row in bin_hap.
// Used when a one-lead bin can never seed a cluster (dev_min_leads_cluster >= 2: the condition of the old prefilter) and no window
// holds more leads than the largest instance of w4s_segment takes; otherwise the sort path (snf_stage_cluster.h) runs.
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
  IT_SCOPE(0)
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
  const unsigned long long big = __ballot(c > 64u);
  if ((threadIdx.x & 63) == 0 && occ) {
    atomicAdd((unsigned long long*)&v.cnt->n_occ, (unsigned long long)__popcll(occ));
    atomicMax((unsigned long long*)&v.cnt->max_win, (unsigned long long)m);
    if (big) atomicAdd((unsigned long long*)&v.cnt->n_big64, (unsigned long long)__popcll(big));
  }
}

// one record per occupied window, ascending: {window, leads, bucket offset, task}; and per 64-position block of the bucket array the
// first window that starts in it or behind it (blk_k0: the windows of block i are blk_k0[i] .. blk_k0[i + 1] - 1 - what w4s_segment's
// wave i owns).  Window k answers for the blocks whose first position lies inside it or is its end: every block has one writer.
SNF_D void win_list(const View& v, int64_t p, uint32_t c, unsigned long long off, unsigned long long k) {
  if (p == 0) v.blk_k0[0] = 0;
  if (!c) return;
  int lo = 0, hi = v.T;            // last task with t_win_off[t] <= p
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (v.t_win_off[mid] <= p) lo = mid; else hi = mid; }
  ((uint4*)v.wlist)[k] = make_uint4((uint32_t)p, c, (uint32_t)off, (uint32_t)lo);
  for (unsigned long long bb = off / 64 + 1; bb <= (off + c) / 64; bb++) v.blk_k0[bb] = (uint32_t)(k + 1);
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
    win_list(v, p, c, off[0], off[1]);
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
    win_list(v, p, c, off[0], off[1]);
    if (p == n - 1) { v.wbase[n] = (uint32_t)tot[0]; v.cnt->n_valid = (int64_t)tot[0]; v.cnt->n_occ = (int64_t)tot[1]; }
  }
}

// W3: every lead into its window's bucket: (attributes << 32 | input index), any order inside the bucket
__global__ void __launch_bounds__(256) w3_scatter(const View v, int64_t n) {
  IT_SCOPE(1)
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

// inclusive running maximum along the positions of a window, 64 positions per round (carry: the maximum of the rounds before)
SNF_D int wave_runmax(int x, int carry) {
  x = wave_incl_max(x);      // (six DPP steps; the __shfl_up form was six round trips through the LDS crossbar)
  return x > carry ? x : carry;
}

// flags of a key_out word (high half) as w4s_segment leaves it for w6t_emit: lead of an eligible bin with a length / without one,
// Lead.seq dropped, first lead of an eligible bin; bits 4-15 the lead's rank among the wave's `leads` (or `leads_long`), bits 16-20
// how many 64-position blocks before the lead's own block the wave sits that owns it, bits 21-31 (seeds) the seed's rank in the wave
#define SNF_WS_NORM 1u
#define SNF_WS_LONG 2u
#define SNF_WS_SEQNULL 4u
#define SNF_WS_SEED 8u

// W4: one wave per SEGMENT of the bucket array.  Wave i owns the occupied windows whose bucket starts in positions [64 i, 64 i + 64)
// (blk_k0, written by w2): a handful of small windows - a window of a 30x genome holds ~23 leads, most hold fewer than eight - or
// one large one that reaches into the following blocks.  One lead per lane and round, whatever window it belongs to: every wave-wide
// step of the old wave-per-window kernel (ballots, running maxima, the rank sort's broadcast reads) works on the lead's own window.
// The sort key of a lead carries its window's index in the wave above (bin, input index): a rank loop that runs past the end of a
// window only meets larger keys, so it needs no bound per lane - one LDS read at an immediate offset, one compare, one add per step.
// CAP: leads the largest window may hold (the host picks the instance).
//   key = window in the wave << 47 | bin in the window << 35 | input index << 3 | hap << 1 | is_long
SNF_D uint64_t wkey_make(uint32_t widx, uint64_t word) {
  const uint32_t a = (uint32_t)(word >> 32);
  return ((uint64_t)widx << 47) | ((uint64_t)win_bin_low(a) << 35) | ((uint64_t)(uint32_t)word << 3) | ((uint64_t)win_hap(a) << 1) | (win_is_long(a) ? 1ull : 0ull);
}
SNF_D uint32_t wkey_group(uint64_t k) { return (uint32_t)(k >> 35); }     // (window, bin): equal for the leads of one bin
SNF_D uint32_t wkey_bin(uint64_t k) { return (uint32_t)(k >> 35) & 0xfffu; }
SNF_D uint32_t wkey_widx(uint64_t k) { return (uint32_t)(k >> 47); }
SNF_D uint32_t wkey_index(uint64_t k) { return (uint32_t)(k >> 3); }
SNF_D bool wkey_long(uint64_t k) { return (k & 1ull) != 0; }
SNF_D uint32_t wkey_hap(uint64_t k) { return (uint32_t)(k >> 1) & 3u; }

template <int CAP>
__global__ void __launch_bounds__(64) w4s_segment(const View v, int64_t n_unused) {
  IT_SCOPE(2)
  constexpr int CAPW = CAP + 64, E = CAPW / 64, PAD = 8;
  __shared__ uint64_t keys[CAPW + CAP + PAD];  // by position relative to the block's first (64 i); sorted in place; behind them "infinity"
  __shared__ uint16_t widx[CAPW];              // window marks, later the head position of every lead's bin
  uint16_t* const hpos = widx;
  __shared__ uint32_t sA[CAPW], sB[CAPW];      // per bin (at its head position): leads | leads with a length << 16;  hap 1 | hap 2 << 16
  __shared__ uint32_t wS[64], wG[64], wB[64];  // per window of the wave: first position (relative), group, first bin
  const int lane = threadIdx.x;
  int64_t i = blockIdx.x;
  if (v.w4_mode == 2) {      // the blocks the small instance left (their number is known on the device only: the grid is the host's upper bound)
    if (i >= (int64_t)v.cnt->n_w4big) return;
    i = v.w4_list[i];
  }
  const int64_t B0 = 64 * i;
  const int64_t n_valid = v.cnt->n_valid;
  // the first 64 positions are this block's own: their words are requested before the wave knows which of them it owns
  uint64_t w0 = 0;
  if (B0 + lane < n_valid) w0 = v.key_in[B0 + lane];
  uint32_t k0 = 0, k1 = 0;
  if (B0 < n_valid) { k0 = v.blk_k0[i]; k1 = B0 + 64 <= n_valid ? v.blk_k0[i + 1] : (uint32_t)v.cnt->n_occ; }
  const int nw = (int)(k1 - k0);                 // <= 64: every window holds a lead and starts inside the block
  if (nw <= 0) { if (lane == 0) { v.ws_seeds[i] = 0; v.ws_nf[i] = 0; v.ws_nl[i] = 0; } return; }
  uint4 wr = make_uint4(0, 0, 0, 0);             // {window, leads, bucket offset, task}
  if (lane < nw) wr = ((const uint4*)v.wlist)[k0 + lane];
  const int xlo = (int)((int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)wr.z) - B0);                 // owned positions: [xlo, xhi)
  const int xhi = (int)((int64_t)(uint32_t)__builtin_amdgcn_readlane((int)(wr.z + wr.y), nw - 1) - B0);   // <= 63 + CAP
  if (v.w4_mode == 1 && __shfl(wave_incl_max(lane < nw ? (int)wr.y : 0), 63, 64) > CAP) {
    // one of its windows holds more than CAP leads (the LDS rows and the pad behind the keys are sized for CAP): the large
    // instance's, through the list - at most as many blocks as the input has such windows
    if (lane == 0) v.w4_list[atomicAdd(&v.cnt->n_w4big, 1ull)] = (int32_t)i;
    return;
  }
  uint64_t wd[E];
  wd[0] = w0;
#pragma unroll
  for (int j = 1; j < E; j++) { const int x = lane + 64 * j; wd[j] = x < xhi ? v.key_in[B0 + x] : 0ull; }
  if (lane < nw) {
    const int t = (int)wr.w;
    const int64_t o0 = v.t_win_off[t], nwin = (v.t_win_off[t + 1] - o0) / SNF_NTYPES, rem = (int64_t)wr.x - o0;
    wS[lane] = (uint32_t)((int64_t)wr.z - B0); wG[lane] = (uint32_t)(t * 8 + (int)(rem / nwin)); wB[lane] = (uint32_t)((rem % nwin) << v.win_bits);
  }
#pragma unroll
  for (int j = 0; j < E; j++) { const int x = lane + 64 * j; if (x < xhi) { sA[x] = 0; sB[x] = 0; widx[x] = 0; } }
  // what a rank loop reads behind the last window: it runs as far as the wave's largest window is long, from any window's start
  const int nall = __shfl(wave_incl_max(lane < nw ? (int)wr.y : 0), 63, 64);
  for (int p = lane; p < nall + PAD; p += 64) keys[xhi + p] = ~0ull;
  __syncthreads();
  if (lane < nw) widx[wS[lane]] = (uint16_t)(lane + 1);
  __syncthreads();
  // the window of every position (a running maximum over the marks at the windows' first positions) -> its sort key
  uint64_t e[E];
  {
    int carry = 0;
#pragma unroll
    for (int j = 0; j < E; j++) {
      e[j] = 0;
      if (64 * j < xhi) {
        const int x = lane + 64 * j;
        const int m = wave_runmax(x < xhi ? (int)widx[x] : 0, carry);
        carry = __shfl(m, 63, 64);
        const bool own = x >= xlo && x < xhi;
        if (own) e[j] = wkey_make((uint32_t)(m - 1), wd[j]);
        if (x < xhi) keys[x] = e[j];
      }
    }
  }
  __syncthreads();
  // rank sort by (bin, arrival) inside every window
  int r[E], ws[E];
#pragma unroll
  for (int j = 0; j < E; j++) {
    r[j] = 0; ws[j] = 0;
    if (64 * j < xhi) {
      const int x = lane + 64 * j;
      const bool own = x >= xlo && x < xhi;
      const int wl = own ? (int)wkey_widx(e[j]) : 0;
      ws[j] = own ? (int)wS[wl] : 0;
      const int wend = own ? (wl + 1 < nw ? (int)wS[wl + 1] : xhi) : 0;
      const int nmax = __shfl(wave_incl_max(wend - ws[j]), 63, 64);
      const uint64_t* kp = keys + ws[j];
      const uint64_t ek = e[j];
      for (int q = 0; q < nmax; q += PAD) {
#pragma unroll
        for (int u = 0; u < PAD; u++) r[j] += kp[q + u] < ek ? 1 : 0;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < E; j++) { const int x = lane + 64 * j; if (x >= xlo && x < xhi) keys[ws[j] + r[j]] = e[j]; }
  __syncthreads();
  // bin heads and the head position of every lead
  uint64_t sk[E];
  {
    int carry = -1;
#pragma unroll
    for (int j = 0; j < E; j++) {
      const int x = lane + 64 * j;
      sk[j] = 0;
      if (64 * j < xhi) {
        int h = -1;
        if (x >= xlo && x < xhi) {
          sk[j] = keys[x];
          if (x == xlo || wkey_group(keys[x - 1]) != wkey_group(sk[j])) h = x;
        }
        const int m = wave_runmax(h, carry);
        if (x >= xlo && x < xhi) hpos[x] = (uint16_t)m;
        carry = __shfl(m, 63, 64);
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < E; j++) {
    const int x = lane + 64 * j;
    if (64 * j >= xhi) break;
    if (x >= xlo && x < xhi) {
      const int h = hpos[x];
      atomicAdd(&sA[h], 1u + (wkey_long(sk[j]) ? 0u : (1u << 16)));
      const uint32_t hp = wkey_hap(sk[j]);
      if (hp) atomicAdd(&sB[h], hp == 1 ? 1u : (1u << 16));
    }
  }
  __syncthreads();
  int cs = 0, cf = 0, cl = 0;      // seeds / leads / long leads of the rounds before
  const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
  for (int j = 0; j < E; j++) {
    const int x = lane + 64 * j;
    if (64 * j >= xhi) break;
    const bool own = x >= xlo && x < xhi;
    bool f_seed = false, f_norm = false, f_long = false, f_null = false;
    int h = 0, all = 0, with_len = 0;
    if (own) {
      h = hpos[x];
      const uint32_t A = sA[h];
      all = (int)(A & 0xffffu); with_len = (int)(A >> 16);
      const bool elig = with_len >= v.cfg.dev_min_leads_cluster;
      f_norm = elig && !wkey_long(sk[j]); f_long = elig && wkey_long(sk[j]); f_seed = elig && x == h;
      f_null = x - h + 1 > v.cfg.consensus_max_reads_bin;      // leadprov.py:406-408 (counts every lead of the bin)
    }
    const unsigned long long bn = __ballot(f_norm), bl = __ballot(f_long), bs = __ballot(f_seed);
    const uint32_t rn = (uint32_t)(cf + __popcll(bn & below)), rl = (uint32_t)(cl + __popcll(bl & below)), rs = (uint32_t)(cs + __popcll(bs & below));
    if (own) {
      const uint32_t hi = (f_norm ? SNF_WS_NORM : 0u) | (f_long ? SNF_WS_LONG : 0u) | (f_null ? SNF_WS_SEQNULL : 0u) | (f_seed ? SNF_WS_SEED : 0u) |
                          ((f_norm ? rn : rl) << 4) | ((uint32_t)j << 16) | (f_seed ? rs << 21 : 0u);
      v.key_out[B0 + x] = ((uint64_t)hi << 32) | (uint64_t)wkey_index(sk[j]);
      if (f_seed) {
        const uint32_t B = sB[h];
        const int wl = (int)wkey_widx(sk[j]);
        const int64_t start = ((int64_t)wB[wl] + (int64_t)wkey_bin(sk[j])) * v.cfg.cluster_binsize;
        v.whead[B0 + x] = (uint64_t)with_len | ((uint64_t)all << 16) | ((uint64_t)(B & 0xffffu) << 32) | ((uint64_t)(B >> 16) << 48);
        // seed start | group << 32 | the seed's offset in the OTHER list (`leads_long` for a seed lead with a length and vice versa) << 52
        v.whead2[B0 + x] = (uint64_t)(uint32_t)(int32_t)start | ((uint64_t)wG[wl] << 32) | ((uint64_t)(f_norm ? rl : rn) << 52);
      }
    }
    cs += __popcll(bs); cf += __popcll(bn); cl += __popcll(bl);
  }
  if (lane == 0) { v.ws_seeds[i] = (uint32_t)cs; v.ws_nf[i] = (uint32_t)cf; v.ws_nl[i] = (uint32_t)cl; }
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

// W6: one THREAD per bucket position: the seeds, `leads` (L, packed records) and `leads_long` (LL) at their global places = the
// offsets of the wave that owns the position (w5's scans) + the rank w4s_segment left in the word.  The pass's one gather (in_rec).
__global__ void __launch_bounds__(256) w6t_emit(const View v, int64_t n_unused) {
  IT_SCOPE(3)
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= v.cnt->n_valid) return;
  const uint64_t word = v.key_out[p];
  const uint32_t hi = (uint32_t)(word >> 32), o = (uint32_t)word;
  if (!(hi & (SNF_WS_NORM | SNF_WS_LONG))) return;      // (a seed lead is one of the two)
  const int64_t owner = p / 64 - (int64_t)((hi >> 16) & 31u);
  const uint32_t rank = (hi >> 4) & 0xfffu;
  int64_t qf = 0, ql = 0;
  if (hi & SNF_WS_NORM) {
    qf = (int64_t)v.ws_nf[owner] + rank;
    v.L[qf] = o;
    // the record as four 16-byte words (a LeadRec object patched in place ends up in LDS: "promote alloca")
    const uint4* src = (const uint4*)&v.in_rec[o];
    uint4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];
    static_assert(offsetof(LeadRec, seq_len) == 20 && offsetof(LeadRec, seq_off) == 24, "words of q1");
    if ((int32_t)q1.y >= 0 && (hi & SNF_WS_SEQNULL)) { q1.y = 0xffffffffu; q1.z = 0; q1.w = 0; }   // 11th+ lead of a bin: Lead.seq = None
    uint4* dst = (uint4*)&v.Lrec[qf];
    dst[0] = q0; dst[1] = q1; dst[2] = q2; dst[3] = q3;
  } else {
    ql = (int64_t)v.ws_nl[owner] + rank;
    v.LL[ql] = o;
  }
  if (hi & SNF_WS_SEED) {
    const uint64_t hd = v.whead[p], h2 = v.whead2[p];
    const uint32_t other = (uint32_t)(h2 >> 52);
    if (hi & SNF_WS_NORM) ql = (int64_t)v.ws_nl[owner] + other; else qf = (int64_t)v.ws_nf[owner] + other;
    const int64_t qs = (int64_t)v.ws_seeds[owner] + (hi >> 21);
    const int with_len = (int)(hd & 0xffffu), all = (int)((hd >> 16) & 0xffffu), h1 = (int)((hd >> 32) & 0xffffu), hh2 = (int)(hd >> 48);
    v.seed_bin[qs] = (int32_t)qs;
    v.seed_lo[qs] = (int32_t)qf; v.seed_hi[qs] = (int32_t)(qf + with_len);
    v.seedL_lo[qs] = (int32_t)ql; v.seedL_hi[qs] = (int32_t)(ql + all - with_len);
    v.seed_start[qs] = (int32_t)(uint32_t)h2;
    v.seed_grp[qs] = (int)((h2 >> 32) & 0xfffffu);
    v.bin_hap[3 * qs + 0] = (uint16_t)(all - h1 - hh2); v.bin_hap[3 * qs + 1] = (uint16_t)h1; v.bin_hap[3 * qs + 2] = (uint16_t)hh2;
  }
}

}  // namespace snf
