// snf_stage_out.h - the output stage: what a stage-1 fetch returns, compacted into ONE block on the device.
//
// Reference semantics: Task.finalize_candidates returns every candidate (parallel.py:129-201, the early exits are commented
// out); CallTask.execute then keeps `[s for s in svcalls if s.qc]` unless config.no_qc and sorts by pos (parallel.py:265-271),
// and only that list is pickled to the parent (parallel.py:757).  SNF_OUT_EXECUTE does the same two statements here, so that
// only what the parent will see crosses PCIe; tasks whose reference run raises (SNF_TASK_ERR_*) yield no records in
// either mode.
//
//   f1  keep flag + sizes per call                     (| tile sums)
//   f2  scan: compacted index, offsets inside the read-name / ALT sections; totals and block layout (OutHdr)
//   f3  per-task stable rank by pos (SNF_OUT_EXECUTE with config.sort)
//   f4  record (read-name offset rewritten) + read names -> block          (needs e1: qc, filter, genotype ...)
// The block is [records | read names]; it goes straight into pinned host memory when it fits the pinned buffer the batch
// holds (zero-copy stores: nothing trails the last kernel but one host wait), otherwise into HBM and the fetch copies it.
// The ALT bytes are a section of their own: all candidates' ALTs in candidate order, written by the ALT kernels themselves
// (pinned host memory or the HBM pool, `alt_base`), so a record keeps its `alt_off`; the few ALT bytes of calls the filter
// drops travel along (a copy behind the consensus kernels would cost the pass more than they do).
#pragma once
#include "snf_stage_final.h"
#include "snf_fused.h"

namespace snf {

SNF_HD bool out_keep(const View& v, const snf_call_t& c) {
  if (v.t_status[c.task_index] != SNF_TASK_OK) return false;
  if ((v.out_mode & SNF_OUT_EXECUTE) && !v.cfg.no_qc && !c.qc) return false;
  return true;
}
SNF_HD int64_t out_align(int64_t x) { return (x + 255) & ~(int64_t)255; }
SNF_HD uint8_t* out_base(const View& v) { return v.out_hdr->in_pinned ? v.out_pin : v.out_dev; }

// deferred supporting read names (View::rn_defer): written now, for the calls the output keeps (2: for all of them)
SNF_HD void d3l_rnames_late_body(int64_t i, const View& v) {
  if (i >= v.cnt->n_calls) return;
  if (v.rn_defer == 2 || out_keep(v, v.calls[i])) d3_rnames_emit(i, v);
}

// F1: keep flag (o_scan doubles as the flag array until f2 has scanned it), sizes
SNF_HD void f1_values(const View& v, int64_t i, int64_t nc, unsigned long long (&val)[2]) {
  val[0] = val[1] = 0;
  if (i >= nc) return;
  const snf_call_t& c = v.calls[i];
  if (!out_keep(v, c)) return;
  val[0] = 1; val[1] = (unsigned long long)c.rn_len;
}
// totals -> block layout; written once per pass by whoever holds the totals (last call's thread, or thread 0 when there are no calls)
SNF_HD void out_layout(const View& v, int64_t n_out, int64_t rn_out) {
  OutHdr h;
  h.n_out = n_out; h.rn_out = rn_out;
  h.off_rn = out_align(n_out * (int64_t)sizeof(snf_call_t));
  h.bytes = out_align(h.off_rn + rn_out * 4);
  h.in_pinned = (!(v.out_mode & SNF_OUT_DEVICE) && v.out_pin && h.bytes <= v.out_pin_cap) ? 1 : 0;
  h._pad = 0;
  *v.out_hdr = h;
}
// F2 per call, given its exclusive offsets
SNF_HD void f2_emit(const View& v, int64_t i, unsigned long long keep, unsigned long long q, unsigned long long rn_off) {
  v.o_scan[i] = (uint32_t)q;
  if (!keep) return;
  v.o_src[q] = (int32_t)i; v.o_dst[q] = (int32_t)q; v.o_key[q] = v.calls[i].pos;
  v.o_rn[q] = (int64_t)rn_off;
}
// F3: final record index of compacted call q = first slot of its task + stable rank by pos among the task's kept calls
SNF_HD void f3_rank_body(int64_t q, const View& v) {
  if (q >= v.out_hdr->n_out) return;
  const int32_t i = v.o_src[q];
  const int t = v.calls[i].task_index;
  const int64_t qlo = v.o_scan[v.t_call_off[t]], qhi = v.o_scan[v.t_call_off[t + 1]];
  const int32_t key = v.o_key[q];
  int64_t rank = 0;
  for (int64_t j = qlo; j < qhi; j++) { const int32_t kj = v.o_key[j]; rank += (kj < key || (kj == key && j < q)) ? 1 : 0; }
  v.o_dst[q] = (int32_t)(qlo + rank);
}
// F4 (thread form): record with rewritten offsets, read names
SNF_HD void f4_emit_body(int64_t q, const View& v) {
  if (q >= v.out_hdr->n_out) return;
  const OutHdr h = *v.out_hdr;
  uint8_t* base = out_base(v);
  const snf_call_t& src = v.calls[v.o_src[q]];
  snf_call_t c = src;
  c.rn_off = v.o_rn[q];
  ((snf_call_t*)base)[v.o_dst[q]] = c;
  uint32_t* rn = (uint32_t*)(base + h.off_rn) + v.o_rn[q];
  for (int32_t k = 0; k < src.rn_len; k++) rn[k] = v.rnames[src.rn_off + k];
}
// plain-scan path (emulation build, SNF_NO_FUSE): F1 writes the three value arrays, three device-wide scans, F2 reads them
SNF_HD void f1_flags_body(int64_t i, const View& v) {
  const int64_t nc = v.cnt->n_calls;
  unsigned long long val[2];
  f1_values(v, i, nc, val);
  v.o_scan[i] = (uint32_t)val[0]; v.sz_rd[i] = (int64_t)val[1];   // (sz_rd of the ALT sizing is free again)
}
SNF_HD void f2_scan_body(int64_t i, const View& v) {
  const int64_t nc = v.cnt->n_calls;
  if (i == 0) out_layout(v, (int64_t)v.pL[nc], v.sc_rd[nc]);
  if (i > nc) return;
  if (i == nc) { v.o_scan[nc] = v.pL[nc]; return; }
  unsigned long long val[2];
  f1_values(v, i, nc, val);
  f2_emit(v, i, val[0], v.pL[i], (unsigned long long)v.sc_rd[i]);
}

// ---- gfx950: F1 | F2 as a "sizes + tile sums" / "tile prefix + block scan + emit" pair like e2a_sizes / e3b_offsets (snf_fused.h):
// every block of F2 sums the preceding 256-call tiles directly (finalize may run more than once per candidate stage, so the
// atomically accumulated super-tile sums of the candidate stage's chains are not used here).  The grid covers the upper bound
// NS; blocks behind the last call publish zeros / return.
__global__ void __launch_bounds__(256) d3lk_rnames_late(const View v, int64_t n) {
  const int64_t nc = v.cnt->n_calls;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nc; i += (int64_t)gridDim.x * 256) d3l_rnames_late_body(i, v);
}
__global__ void __launch_bounds__(256) f1k_outflags(const View v, int64_t n) {
  __shared__ unsigned long long lds[4 * 2];
  const int64_t nc = v.cnt->n_calls;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  unsigned long long val[2], excl[2], tot[2];
  f1_values(v, i, nc, val);
  block_exscan256<2>(val, excl, tot, lds);
  if (threadIdx.x == 0) for (int k = 0; k < 2; k++) v.tile_sums[(int64_t)(TS_OUT + k) * v.tile_stride + blockIdx.x] = tot[k];
  if (i == 0 && nc == 0) { out_layout(v, 0, 0); v.o_scan[0] = 0; }
}
__global__ void __launch_bounds__(256) f2k_outscan(const View v, int64_t n) {
  __shared__ unsigned long long lds[4 * 2];
  const int64_t nc = v.cnt->n_calls;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if ((int64_t)blockIdx.x * 256 >= nc) return;      // (whole block: no barrier is skipped by part of it)
  unsigned long long part[2], dummy[2], prefix[2], val[2], excl[2], tot[2];
#pragma unroll
  for (int k = 0; k < 2; k++) {
    unsigned long long a = 0;
    for (int64_t j = threadIdx.x; j < (int64_t)blockIdx.x; j += 256) a += v.tile_sums[(int64_t)(TS_OUT + k) * v.tile_stride + j];
    part[k] = a;
  }
  block_exscan256<2>(part, dummy, prefix, lds);     // prefix = sums of all preceding tiles
  f1_values(v, i, nc, val);
  block_exscan256<2>(val, excl, tot, lds);
  if (i < nc) {
    const unsigned long long o0 = prefix[0] + excl[0], o1 = prefix[1] + excl[1];
    f2_emit(v, i, val[0], o0, o1);
    if (i == nc - 1) {
      v.o_scan[nc] = (uint32_t)(o0 + val[0]);
      out_layout(v, (int64_t)(o0 + val[0]), (int64_t)(o1 + val[1]));
    }
  }
}
// one wave per kept call: the lanes stride over the keys of the call's task (contiguous, cache-resident) and the ballots are counted
__global__ void __launch_bounds__(256) f3k_rank(const View v, int64_t n) {
  const int64_t n_out = v.out_hdr->n_out;
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * 256) >> 6;
  for (int64_t q = wave; q < n_out; q += nw) {
    const int t = v.calls[v.o_src[q]].task_index;
    const int64_t qlo = v.o_scan[v.t_call_off[t]], qhi = v.o_scan[v.t_call_off[t + 1]];
    const int32_t key = v.o_key[q];
    int rank = 0;
    for (int64_t j = qlo + lane; j < qhi; j += 64) { const int32_t kj = v.o_key[j]; rank += (kj < key || (kj == key && j < q)) ? 1 : 0; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) rank += __shfl_xor(rank, d, 64);
    if (lane == 0) v.o_dst[q] = (int32_t)(qlo + rank);
  }
}
typedef uint4 __attribute__((aligned(1))) out_u128_unaligned;
// one wave per kept call: the 240-byte record as fifteen 16-byte words (offsets patched in registers), read names lane-strided
__global__ void __launch_bounds__(256) f4w_emit(const View v, int64_t n) {
  IT_SCOPE(13)
  static_assert(sizeof(snf_call_t) == 240 && sizeof(snf_call_t) % 16 == 0, "record copied as 16-byte words");
  const OutHdr h = *v.out_hdr;
  uint8_t* base = h.in_pinned ? v.out_pin : v.out_dev;
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * 256) >> 6;
  for (int64_t q = wave; q < h.n_out; q += nw) {
    const int32_t i = v.o_src[q];
    const snf_call_t* src = v.calls + i;
    const int64_t rn_src = src->rn_off; const int32_t rn_len = src->rn_len;
    const int64_t rn_dst = v.o_rn[q];
    if (lane < (int)(sizeof(snf_call_t) / 16)) {
      union { uint4 w; uint8_t b[16]; } u;
      u.w = ((const uint4*)src)[lane];
      constexpr int orn = (int)offsetof(snf_call_t, rn_off);
      static_assert(orn % 8 == 0, "the 64-bit field lies inside one 16-byte word");
      if (lane == orn / 16) memcpy(u.b + orn % 16, &rn_dst, 8);
      ((uint4*)(base + (int64_t)v.o_dst[q] * (int64_t)sizeof(snf_call_t)))[lane] = u.w;
    }
    uint32_t* rn = (uint32_t*)(base + h.off_rn) + rn_dst;
    if (!v.rn_from_src) { for (int32_t k = lane; k < rn_len; k += 64) rn[k] = v.rnames[rn_src + k]; continue; }
    // names the candidate stage only sized (View::rn_defer): d3_rnames_emit, one wave per call, straight into the block
    const CallX& x = v.callx[i];
    const int32_t nq = x.rn_nq;
    const int32_t* a1 = v.w1 + x.flo;
    for (int32_t k = lane; k < nq; k += 64) rn[k] = (uint32_t)a1[k];
    if (src->svtype == SNF_INS && rn_len > nq) {
      const int32_t hd = v.cl_head[x.cluster];
      const int32_t llo = v.seedL_lo[hd], lhi = v.seedL_hi[v.c_last[hd]];
      int32_t w = nq;
      for (int32_t p0 = llo; p0 < lhi; p0 += 64) {      // in lead order: first appearance of a read that is not among the call's own
        const int32_t p = p0 + lane;
        bool add = false; int32_t qn = 0;
        if (p < lhi) {
          qn = (int32_t)v.in_qname[v.LL[p]];
          add = true;
          for (int32_t y = llo; y < p; y++) if ((int32_t)v.in_qname[v.LL[y]] == qn) { add = false; break; }
          if (add && contains_sorted_i32(a1, nq, qn)) add = false;
        }
        const unsigned long long mk = __ballot(add);
        if (add) rn[w + __builtin_popcountll(mk & ((1ull << lane) - 1ull))] = (uint32_t)qn;
        w += __builtin_popcountll(mk);
      }
    }
  }
}

// staged result: HBM -> pinned host memory, 16 bytes per lane (which = 0: the result block, 1: the ALT section); sizes read here
__global__ void __launch_bounds__(256) z2_stage_copy(const View v, int64_t which) {
  const int64_t n = which == 0 ? v.out_hdr->bytes : v.cnt->alt_total;
  const uint8_t* src = which == 0 ? v.out_dev : v.alt_pool;
  uint8_t* dst = which == 0 ? v.stage_out_pin : v.stage_alt_pin;
  if (!dst || n > (which == 0 ? v.stage_out_cap : v.stage_alt_cap)) return;      // (the fetch sees the same sizes and copies itself)
  const int64_t n16 = n >> 4, stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) ((uint4*)dst)[i] = ((const uint4*)src)[i];
  const int64_t tail = (n16 << 4) + (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (tail < n) dst[tail] = src[tail];
}

}  // namespace snf
