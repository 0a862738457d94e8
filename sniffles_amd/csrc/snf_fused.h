// snf_fused.h - gfx950: "sizes -> exclusive scan -> emit" chains as two launches (block sums, then block prefix + scan +
// emit) instead of a size kernel, one device-wide scan per value and an emit kernel.  The stages between the big
// kernels are launch- and latency-bound, so the number of dependent launches is what they cost.
#pragma once
#include "snf_stage_final.h"
#include "snf_stage_cluster.h"
#include "snf_stage_call.h"

namespace snf {

// exclusive scan of K 64-bit values per thread over a 256-thread block; tot = block totals.  lds: 4 * K words.
template <int K>
SNF_D void block_exscan256(const unsigned long long (&val)[K], unsigned long long (&excl)[K], unsigned long long (&tot)[K],
                           unsigned long long* lds) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  unsigned long long incl[K];
#pragma unroll
  for (int k = 0; k < K; k++) {
    unsigned long long x = val[k];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
    incl[k] = x;
    if (lane == 63) lds[wid * K + k] = x;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; k++) {
    unsigned long long base = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { const unsigned long long t = lds[w * K + k]; all += t; if (w < wid) base += t; }
    excl[k] = base + incl[k] - val[k];
    tot[k] = all;
  }
  __syncthreads();
}

// ---- two-level tile sums: 256-element tiles, 64 tiles per super tile.  A "sizes" kernel publishes its block total
// (plain store) and adds it to its super tile (atomics spread over N/16384 addresses, zeroed at the start of the pass);
// the "emit" kernel of the chain sums the preceding super tiles and the preceding tiles of its own super tile
// (<= N/16384 + 63 loads per value and block), scans the block and has every element's exclusive offset.
template <int K>
SNF_D void tile_publish(const View& v, int slot0, const unsigned long long (&val)[K], unsigned long long* lds) {
  unsigned long long excl[K], tot[K];
  block_exscan256<K>(val, excl, tot, lds);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < K; k++) {
      v.tile_sums[(int64_t)(slot0 + k) * v.tile_stride + blockIdx.x] = tot[k];
      if (tot[k]) atomicAdd(&v.tile_super[(int64_t)(slot0 + k) * v.super_stride + (blockIdx.x >> 6)], tot[k]);
    }
  }
}
template <int K>
SNF_D void tile_scan(const View& v, int slot0, const unsigned long long (&val)[K], unsigned long long (&off)[K], unsigned long long* lds) {
  unsigned long long part[K], dummy[K], prefix[K], excl[K], tot[K];
  const int64_t sup = blockIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; k++) {
    unsigned long long a = 0;
    for (int64_t s = threadIdx.x; s < sup; s += 256) a += v.tile_super[(int64_t)(slot0 + k) * v.super_stride + s];
    const int64_t t = (sup << 6) + threadIdx.x;
    if (threadIdx.x < 64 && t < (int64_t)blockIdx.x) a += v.tile_sums[(int64_t)(slot0 + k) * v.tile_stride + t];
    part[k] = a;
  }
  block_exscan256<K>(part, dummy, prefix, lds);
  block_exscan256<K>(val, excl, tot, lds);
#pragma unroll
  for (int k = 0; k < K; k++) off[k] = prefix[k] + excl[k];
}

// ---- the same chain in ONE launch: decoupled look-back.  A block takes a ticket (= its tile: tiles start in ticket order, so every
// predecessor of a tile is running or done), scans its 256 elements, publishes the tile aggregate, sums the aggregates of the tiles
// before it back to the nearest one whose inclusive prefix is known (64 predecessors per step, one per lane of wave 0), publishes its
// own inclusive prefix and emits.  Every published word carries the tag of its launch - the pass (View::chain_epoch, a counter in
// HBM that z0_init bumps) and the slot, which is used once per pass: nothing is zeroed between launches; the last ticket holder
// resets the ticket counter.  Values stay below 2^38.
#define SNF_CHAIN_TAG_SHIFT 38
SNF_D int64_t chain_ticket(const View& v, int slot0, unsigned long long* lds) {
  if (threadIdx.x == 0) {
    const uint32_t t = atomicAdd(&v.chain_ticket[slot0], 1u);
    if (t + 1 == gridDim.x) v.chain_ticket[slot0] = 0;      // every block holds its ticket: clean for the next launch on this slot
    lds[0] = t;
  }
  __syncthreads();
  const int64_t tile = (int64_t)lds[0];
  __syncthreads();
  return tile;
}
template <int K>
SNF_D void chain_scan(const View& v, int slot0, int64_t tile, const unsigned long long (&val)[K], unsigned long long (&off)[K],
                      unsigned long long (&total)[K], unsigned long long* lds) {
  unsigned long long excl[K], tot[K];
  block_exscan256<K>(val, excl, tot, lds);
  const unsigned long long launch = (((unsigned long long)(*v.chain_epoch) & 0x7ffffull) << 5) | (unsigned long long)slot0;
  const unsigned long long tagA = (launch * 4 + 1) << SNF_CHAIN_TAG_SHIFT, tagI = tagA + (1ull << SNF_CHAIN_TAG_SHIFT);
  const unsigned long long vmask = (1ull << SNF_CHAIN_TAG_SHIFT) - 1ull;
  const int lane = threadIdx.x & 63;
  if (threadIdx.x < 64) {
#pragma unroll
    for (int k = 0; k < K; k++) {
      unsigned long long* st = v.chain + ((int64_t)(slot0 + k) * v.chain_stride) * 2;
      if (lane == 0 && tile > 0) st_agent_u64(&st[2 * tile], tagA | tot[k]);
      unsigned long long ex = 0;
      int64_t j = tile - 1;
      bool done = tile == 0;
      while (!done) {
        const int64_t t = j - lane;
        unsigned long long w = 0; int state = 2;      // (a lane before tile 0 stands for the empty prefix)
        if (t >= 0) {
          for (;;) {
            const unsigned long long wi = ld_agent_u64(&st[2 * t + 1]);
            if ((wi & ~vmask) == tagI) { w = wi & vmask; state = 2; break; }
            const unsigned long long wa = ld_agent_u64(&st[2 * t]);
            if ((wa & ~vmask) == tagA) { w = wa & vmask; state = 1; break; }
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_s_sleep(1);
#endif
          }
        }
        const unsigned long long inc = __ballot(state == 2);
        const int first = __ffsll((long long)inc) - 1;      // nearest tile with a known inclusive prefix (or the start)
        unsigned long long part = (inc == 0 || lane <= first) ? w : 0ull;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
        ex += part;
        if (inc) done = true; else j -= 64;
      }
      if (lane == 0) { st_agent_u64(&st[2 * tile + 1], tagI | (ex + tot[k])); lds[k] = ex; }
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; k++) { off[k] = lds[k] + excl[k]; total[k] = lds[k] + tot[k]; }
  __syncthreads();
}
#define SNF_CHAIN_HEAD(name, slot)                                            \
  __global__ void __launch_bounds__(256) name(const View v, int64_t n) {      \
    __shared__ unsigned long long lds[24];                                    \
    const int64_t tile = chain_ticket(v, slot, lds);                          \
    const int64_t p = tile * 256 + threadIdx.x;

// slots of the candidate stage (the ALT chain of finalize reuses 0..4 with direct sums)
// (slot ids: enum TS_* in snf_view.h)

#define SNF_FUSED_HEAD(name)                                                  \
  __global__ void __launch_bounds__(256) name(const View v, int64_t n) {      \
    __shared__ unsigned long long lds[8];                                     \
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;

// start of a pass: every small reset in one launch (instead of ~17 fill kernels in front of the first real kernel)
__global__ void __launch_bounds__(256) z0_init(const View v, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t T = v.T, G = 8 * T + 8;
  if (i < (int64_t)(sizeof(Counts) / 8)) ((unsigned long long*)v.cnt)[i] = 0;
  if (i <= T) { v.t_cov_sum[i] = 0; v.t_status[i] = 0; }
  if (i <= T + 1) v.t_call_off[i] = 0;
  if (i < G) { v.grp_first_bin[i] = -1; v.grp_seed_lo[i] = -1; v.grp_seed_hi[i] = -1; v.grp_dirty[i] = 0; }
  if (i < TS_SLOTS * v.super_stride) v.tile_super[i] = 0;
  if (i == 0) *v.chain_epoch += 1u;                          // tags of this pass's chain launches (chain_scan)
  if (v.wave_path && i < 3 * 64 * 16) v.big_cnt[i] = 0;      // lists of the big-cluster kernels (x_big)
  if (v.wave_path && i < 3 * 64 * 16) v.d2cnt[i] = 0;       // lists of the grouped call kernels (snf_wave_call_g.h)
  if (i == 0 && v.NS > 0) {
    const int64_t N = v.NS;
    v.headflag[N] = 0; v.eligflag[N] = 0; v.fN[N] = 0; v.fL[N] = 0; v.runflag[N] = 0; v.clflag[N] = 0; v.rcflag[N] = 0; v.cdflag[N] = 0;
  }
}

// A0 (occupancy prefilter): keep flags + tile sums | scan + compaction of (key, input index) + clearing of the marks
SNF_FUSED_HEAD(a0k_keep)
  if (p < n) a0_keep_body(p, v);
  unsigned long long val[1] = {p < n ? (unsigned long long)v.pf_keep[p] : 0ull};
  tile_publish<1>(v, TS_KEEP, val, lds);
}
SNF_FUSED_HEAD(a0k_compact)
  unsigned long long val[1] = {p < n ? (unsigned long long)v.pf_keep[p] : 0ull}, off[1];
  tile_scan<1>(v, TS_KEEP, val, off, lds);
  if (p < n) {
    v.pf_scan[p] = (uint32_t)off[0];
    if (p == n - 1) v.cnt->n_kept = (int64_t)(off[0] + val[0]);
    a0_emit(p, v);
  }
}
// A2 + tile sums of the bin-head flags | scan + A3
SNF_FUSED_HEAD(a2k_heads)
  if (p < n) { a2_heads_body(p, v); v.rcflag[p] = 0; }  // rcflag: reset for d1w_refine / d1_refine (rc_emit sets it)
  unsigned long long val[1] = {p < n ? (unsigned long long)v.headflag[p] : 0ull};
  tile_publish<1>(v, TS_BINS, val, lds);
}
SNF_FUSED_HEAD(a3k_bins)
  unsigned long long val[1] = {p < n ? (unsigned long long)v.headflag[p] : 0ull}, off[1];
  tile_scan<1>(v, TS_BINS, val, off, lds);
  if (p < n) {
    v.headscan[p] = (uint32_t)off[0];
    if (p == n - 1) { const int64_t nb = (int64_t)(off[0] + val[0]); v.headscan[n] = (uint32_t)nb; v.cnt->n_bins = nb; v.bin_lo[nb] = (int32_t)v.cnt->n_valid; }
    a3_emit(p, v);
  }
}
// A4 + tile sums of the seed-eligibility flags | (A5, A6 in between) | scan + A7
SNF_FUSED_HEAD(a4k_binstats)
  if (p < n) a4_binstats_body(p, v);
  unsigned long long val[1] = {p < n ? (unsigned long long)v.eligflag[p] : 0ull};
  tile_publish<1>(v, TS_SEEDS, val, lds);
}
SNF_FUSED_HEAD(a7k_seeds)
  unsigned long long val[1] = {p < n ? (unsigned long long)v.eligflag[p] : 0ull}, off[1];
  tile_scan<1>(v, TS_SEEDS, val, off, lds);
  if (p < n) {
    v.eligscan[p] = (uint32_t)off[0];
    if (p == n - 1) { v.eligscan[n] = (uint32_t)(off[0] + val[0]); v.cnt->n_seeds = (int64_t)(off[0] + val[0]); }
    a7_seeds_body(p, v);
  }
}
// A5 + tile sums of the two membership flags | scans + A6
SNF_FUSED_HEAD(a5k_leadflags)
  if (p < n) a5_leadflags_body(p, v);
  unsigned long long val[2] = {p < n ? (unsigned long long)v.fN[p] : 0ull, p < n ? (unsigned long long)v.fL[p] : 0ull};
  tile_publish<2>(v, TS_LEADS, val, lds);
}
SNF_FUSED_HEAD(a6k_scatter)
  unsigned long long val[2] = {p < n ? (unsigned long long)v.fN[p] : 0ull, p < n ? (unsigned long long)v.fL[p] : 0ull}, off[2];
  tile_scan<2>(v, TS_LEADS, val, off, lds);
  if (p < n) {
    v.pN[p] = (uint32_t)off[0]; v.pL[p] = (uint32_t)off[1];
    if (p == n - 1) {
      v.pN[n] = (uint32_t)(off[0] + val[0]); v.pL[n] = (uint32_t)(off[1] + val[1]);
      v.cnt->NF = (int64_t)(off[0] + val[0]); v.cnt->NLL = (int64_t)(off[1] + val[1]);
    }
    a6_emit(p, v);
  }
}
// B1 + tile sums of the run-cut flags | scan + B2
SNF_FUSED_HEAD(b1k_seedmetrics)
  if (p < n) b1_seedmetrics_body(p, v);
  unsigned long long val[1] = {p < n ? (unsigned long long)v.runflag[p] : 0ull};
  tile_publish<1>(v, TS_RUNS, val, lds);
}
SNF_FUSED_HEAD(b2k_runs)
  unsigned long long val[1] = {p < n ? (unsigned long long)v.runflag[p] : 0ull}, off[1];
  tile_scan<1>(v, TS_RUNS, val, off, lds);
  if (p < n) {
    v.runscan[p] = (uint32_t)off[0];
    if (p == n - 1) { const int64_t nr = (int64_t)(off[0] + val[0]); v.runscan[n] = (uint32_t)nr; v.cnt->n_runs = nr; v.run_first[nr] = (int32_t)v.cnt->n_seeds; }
    b2_emit(p, v);
  }
}
// generic "count an existing flag array" + scan + emit chains: clusters (C4), refined clusters (D1b), calls (D3a)
#define SNF_FLAGSUM(name, flag, slot)                                          \
  SNF_FUSED_HEAD(name)                                                          \
    unsigned long long val[1] = {p < n ? (unsigned long long)v.flag[p] : 0ull}; \
    tile_publish<1>(v, slot, val, lds);                                         \
  }
SNF_FLAGSUM(c4a_count, clflag, TS_CLUSTERS)
SNF_FLAGSUM(d1a_count, rcflag, TS_REFINED)
// (flags behind n_rc are stale: the wave kernels only write the flags of existing refined clusters, and the thread kernel that
// used to reset the rest is not launched next to them)
SNF_FUSED_HEAD(d3a_count)
  unsigned long long val[1] = {(p < n && p < v.cnt->n_rc) ? (unsigned long long)v.cdflag[p] : 0ull};
  tile_publish<1>(v, TS_CALLS, val, lds);
}
SNF_FUSED_HEAD(c4k_clusters)
  unsigned long long val[1] = {p < n ? (unsigned long long)v.clflag[p] : 0ull}, off[1];
  tile_scan<1>(v, TS_CLUSTERS, val, off, lds);
  if (p < n) {
    v.clscan[p] = (uint32_t)off[0];
    if (p == n - 1) { v.clscan[n] = (uint32_t)(off[0] + val[0]); v.cnt->n_clusters = (int64_t)(off[0] + val[0]); }
    c4_emit(p, v);
  }
}
SNF_FUSED_HEAD(d1bk_rctable)
  unsigned long long val[1] = {p < n ? (unsigned long long)v.rcflag[p] : 0ull}, off[1];
  tile_scan<1>(v, TS_REFINED, val, off, lds);
  if (p < n) {
    v.rcscan[p] = (uint32_t)off[0];
    if (p == n - 1) { v.rcscan[n] = (uint32_t)(off[0] + val[0]); v.cnt->n_rc = (int64_t)(off[0] + val[0]); }
    d1b_emit(p, v);
  }
}
SNF_FUSED_HEAD(d3ck_compact)
  unsigned long long val[1] = {(p < n && p < v.cnt->n_rc) ? (unsigned long long)v.cdflag[p] : 0ull}, off[1];
  tile_scan<1>(v, TS_CALLS, val, off, lds);
  if (p < n) {
    v.cdscan[p] = (uint32_t)off[0];
    if (p == n - 1) { v.cdscan[n] = (uint32_t)(off[0] + val[0]); v.cnt->n_calls = (int64_t)(off[0] + val[0]); }
    d3_compact_emit(p, v);
  }
}

// D3c (sv ids, read-name counts) + tile sums | scan + D3d (supporting read names)
SNF_FUSED_HEAD(d3sk_svid)
  if (p < n) d3_svid_body(p, v);
  unsigned long long val[1] = {p < n ? (unsigned long long)v.rnf[p] : 0ull};
  tile_publish<1>(v, TS_RNAMES, val, lds);
}
SNF_FUSED_HEAD(d3rk_rnames)
  unsigned long long val[1] = {p < n ? (unsigned long long)v.rnf[p] : 0ull}, off[1];
  tile_scan<1>(v, TS_RNAMES, val, off, lds);
  if (p < n) {
    v.rnp[p] = (uint32_t)off[0];
    if (p == n - 1) { const int64_t tot = (int64_t)(off[0] + val[0]); v.rnp[n] = (uint32_t)tot; v.cnt->rn_total = tot; *v.res_rn_total = tot; }
    if (!v.rn_defer) d3_rnames_emit(p, v);
  }
}

// ---- single-launch forms of the pairs above (chain_scan)
// B1 + B2: seed metrics, run cuts, run table
SNF_CHAIN_HEAD(b12c_seedruns, TS_RUNS)
  if (p < n) b1_seedmetrics_body(p, v);
  unsigned long long val[1] = {p < n ? (unsigned long long)v.runflag[p] : 0ull}, off[1], tot[1];
  chain_scan<1>(v, TS_RUNS, tile, val, off, tot, lds);
  if (p < n) {
    v.runscan[p] = (uint32_t)off[0];
    if (p == n - 1) { const int64_t nr = (int64_t)tot[0]; v.runscan[n] = (uint32_t)nr; v.cnt->n_runs = nr; v.run_first[nr] = (int32_t)v.cnt->n_seeds; }
    b2_emit(p, v);
  }
}
// C4: merged cluster table
SNF_CHAIN_HEAD(c4c_clusters, TS_CLUSTERS)
  unsigned long long val[1] = {p < n ? (unsigned long long)v.clflag[p] : 0ull}, off[1], tot[1];
  chain_scan<1>(v, TS_CLUSTERS, tile, val, off, tot, lds);
  if (p < n) {
    v.clscan[p] = (uint32_t)off[0];
    if (p == n - 1) { v.clscan[n] = (uint32_t)tot[0]; v.cnt->n_clusters = (int64_t)tot[0]; }
    c4_emit(p, v);
  }
}
// D1b: refined cluster table
SNF_CHAIN_HEAD(d1bc_rctable, TS_REFINED)
  unsigned long long val[1] = {p < n ? (unsigned long long)v.rcflag[p] : 0ull}, off[1], tot[1];
  chain_scan<1>(v, TS_REFINED, tile, val, off, tot, lds);
  if (p < n) {
    v.rcscan[p] = (uint32_t)off[0];
    if (p == n - 1) { v.rcscan[n] = (uint32_t)tot[0]; v.cnt->n_rc = (int64_t)tot[0]; }
    d1b_emit(p, v);
  }
}
// D3: candidate compaction
SNF_CHAIN_HEAD(d3cc_compact, TS_CALLS)
  unsigned long long val[1] = {(p < n && p < v.cnt->n_rc) ? (unsigned long long)v.cdflag[p] : 0ull}, off[1], tot[1];
  chain_scan<1>(v, TS_CALLS, tile, val, off, tot, lds);
  if (p < n) {
    v.cdscan[p] = (uint32_t)off[0];
    if (p == n - 1) { v.cdscan[n] = (uint32_t)tot[0]; v.cnt->n_calls = (int64_t)tot[0]; }
    d3_compact_emit(p, v);
  }
}
// D3c + D3d: sv ids, read-name counts, supporting read names
SNF_CHAIN_HEAD(d3src_svid_rnames, TS_RNAMES)
  if (p < n) d3_svid_body(p, v);
  unsigned long long val[1] = {p < n ? (unsigned long long)v.rnf[p] : 0ull}, off[1], tot[1];
  chain_scan<1>(v, TS_RNAMES, tile, val, off, tot, lds);
  if (p < n) {
    v.rnp[p] = (uint32_t)off[0];
    if (p == n - 1) { v.rnp[n] = (uint32_t)tot[0]; v.cnt->rn_total = (int64_t)tot[0]; *v.res_rn_total = (int64_t)tot[0]; }
    if (!v.rn_defer) d3_rnames_emit(p, v);
  }
}

// ---- ALT sizing: E2 (sizes per call) + per-tile sums; E3 (offsets, totals, work descriptors)
#define SNF_ALT_K 5
SNF_D void alt_values(const View& v, int64_t i, int64_t nc, unsigned long long (&val)[SNF_ALT_K]) {
  if (i < nc) { val[0] = v.fN[i]; val[1] = v.fL[i]; val[2] = (unsigned long long)v.sz_tab[i]; val[3] = (unsigned long long)v.sz_aln[i]; val[4] = (unsigned long long)v.sz_rd[i]; }
  else { for (int k = 0; k < SNF_ALT_K; k++) val[k] = 0; }
}
__global__ void __launch_bounds__(256) e2a_sizes(const View v, int64_t n_unused) {
  __shared__ unsigned long long lds[4 * SNF_ALT_K];
  const int64_t nc = v.cnt->n_calls;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (int64_t k = i; k < 4 * 64 * 16; k += (int64_t)gridDim.x * 256) v.stripes[k] = 0;   // byte counters of the ALT kernels (this pass)
  e2_best_body(i, v);
  unsigned long long val[SNF_ALT_K], excl[SNF_ALT_K], tot[SNF_ALT_K];
  alt_values(v, i, nc, val);
  block_exscan256<SNF_ALT_K>(val, excl, tot, lds);
  if (threadIdx.x == 0) for (int k = 0; k < SNF_ALT_K; k++) v.tile_sums[(int64_t)k * v.tile_stride + blockIdx.x] = tot[k];
}
__global__ void __launch_bounds__(256) e3b_offsets(const View v, int64_t n_unused) {
  IT_SCOPE(8)
  __shared__ unsigned long long lds[4 * SNF_ALT_K];
  const int64_t nc = v.cnt->n_calls;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if ((int64_t)blockIdx.x * 256 >= nc) return;   // (the grid covers an upper bound of the calls; whole blocks leave)
  unsigned long long part[SNF_ALT_K], dummy[SNF_ALT_K], prefix[SNF_ALT_K];
#pragma unroll
  for (int k = 0; k < SNF_ALT_K; k++) {
    unsigned long long a = 0;
    for (int64_t j = threadIdx.x; j < (int64_t)blockIdx.x; j += 256) a += v.tile_sums[(int64_t)k * v.tile_stride + j];
    part[k] = a;
  }
  block_exscan256<SNF_ALT_K>(part, dummy, prefix, lds);   // prefix = sums of all preceding tiles
  unsigned long long val[SNF_ALT_K], excl[SNF_ALT_K], tot[SNF_ALT_K];
  alt_values(v, i, nc, val);
  block_exscan256<SNF_ALT_K>(val, excl, tot, lds);
  if (i < nc) {
    const unsigned long long o0 = prefix[0] + excl[0], o1 = prefix[1] + excl[1], o2 = prefix[2] + excl[2], o3 = prefix[3] + excl[3],
                             o4 = prefix[4] + excl[4];
    v.pN[i] = (uint32_t)o0; v.pL[i] = (uint32_t)o1; v.sc_tab[i] = (int64_t)o2; v.sc_aln[i] = (int64_t)o3; v.sc_rd[i] = (int64_t)o4;
    if (i == nc - 1) {  // totals
      v.pN[nc] = (uint32_t)(o0 + val[0]); v.pL[nc] = (uint32_t)(o1 + val[1]);
      v.sc_tab[nc] = (int64_t)(o2 + val[2]); v.sc_aln[nc] = (int64_t)(o3 + val[3]); v.sc_rd[nc] = (int64_t)(o4 + val[4]);
      v.cnt->alt_total = (int64_t)(o0 + val[0]); v.cnt->n_cons = (int64_t)(o1 + val[1]);
      v.cnt->tab_total = (int64_t)(o2 + val[2]); v.cnt->aln_total = (int64_t)(o3 + val[3]); v.cnt->n_cons_reads = (int64_t)(o4 + val[4]);
      alt_decide(v, (int64_t)(o0 + val[0]));
    }
    e3_emit(i, v);
  }
}

}  // namespace snf
