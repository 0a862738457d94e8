// snf_fused.h - gfx950: "sizes -> exclusive scan -> emit" chains as two launches (block sums, then block prefix + scan +
// emit) instead of a size kernel, one device-wide scan per value and an emit kernel.  The stages between the big
// kernels are launch- and latency-bound, so the number of dependent launches is what they cost.
#pragma once
#include "snf_stage_final.h"

#ifndef SNF_EMU
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
  e2_best_body(i, v);
  unsigned long long val[SNF_ALT_K], excl[SNF_ALT_K], tot[SNF_ALT_K];
  alt_values(v, i, nc, val);
  block_exscan256<SNF_ALT_K>(val, excl, tot, lds);
  if (threadIdx.x == 0) for (int k = 0; k < SNF_ALT_K; k++) v.tile_sums[(int64_t)k * v.tile_stride + blockIdx.x] = tot[k];
}
__global__ void __launch_bounds__(256) e3b_offsets(const View v, int64_t n_unused) {
  __shared__ unsigned long long lds[4 * SNF_ALT_K];
  const int64_t nc = v.cnt->n_calls;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
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
    }
    e3_emit(i, v);
  }
}

}  // namespace snf
#endif  // !SNF_EMU
