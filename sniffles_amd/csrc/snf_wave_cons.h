// snf_wave_cons.h - gfx950 workgroup-per-INS-call implementation of the ALT stage: k-mer anchored consensus
// (consensus.novel_from_reads, consensus.py:280-394) or the verbatim copy of the best read
// (postprocessing.py:65-66).  e4_anchor/e5_align/e6_vote (thread-per-item) only serve calls that do not fit the
// LDS budget of the LARGE class below.
//
// Per consensus call: 256 threads build the anchor hash table of the best read in LDS (sampled k-mers seen
// exactly once).  Then each of the 4 waves takes reads r = w, w+4, ...:
//   1. all sampled k-mers of the read are loaded up front (independent 8-byte loads), looked up in LDS and
//      compacted IN ORDER (ballot + popcount)
//   2. the monotone anchor chain (accept iff i > last accepted i) is a prefix-max filter
//   3. per segment between consecutive anchors (one lane each): clipped advance, identity on offsets 1..n and
//      column identity of the copied slice with 8-byte vector compares (the column cursor has the closed form
//      min(L, c0 + j - j0))
//   4. lane 0 groups consecutive copied segments into runs and applies the run filter
//      (matches/len > 0.5 and matches > 5)
//   5. LDS-vote instances (LCAP > 0): the aligned row is never materialised.  Every copied segment of a kept read (one
//      lane per segment) adds its bases to per-column vote counters in LDS (four 8-bit counters per column, one ds_add
//      each; any byte other than A/C/G/T goes to a short escape list) - dash columns cost nothing.  The best read, and
//      in the SMALL class the other read as well, is staged in LDS once (one coalesced pass) and every later phase reads
//      it from there.  Rows instance (LCAP == 0, calls beyond the LDS budget of the counters, or whose escape list
//      overflowed): the row is written column-parallel (binary search of the owning segment) to v.aln;
// finally the workgroup votes every column: from the counters, or over the rows it just wrote (still in L2).
// Size classes: SMALL (<= 120 sampled positions, <= 384 bp, <= 64 others: ~13 KB LDS), LARGE (<= 500 positions,
// <= 8192 bp, <= 254 others: 72 KB LDS) and ROWS (<= 500 positions, <= 512 others, < 65000 bp; global rows).
// Input sequences must not contain '-' (checked at snf_batch_add_task): the reference treats it as a gap.
#pragma once
#include "snf_stage_final.h"

namespace snf {

#define SNF_KEY_EMPTY (~0ull)

// instrumented build (-DSNF_CONS_PROFILE, tools/cons_profile.sh): s_memtime stamps between the phases, summed per wave and
// added to Counts::dbg at the end of the workgroup; [base + 0..6] phases, [base + 7] longest workgroup, [base + 8] workgroups
#ifdef SNF_CONS_PROFILE
#define SNF_PT_DECL unsigned long long pt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long pt_t = __builtin_amdgcn_s_memtime(); const unsigned long long pt_t0 = pt_t;
#define SNF_PT(k) do { const unsigned long long pt_n = __builtin_amdgcn_s_memtime(); pt_acc[k] += pt_n - pt_t; pt_t = pt_n; } while (0)
#define SNF_PT_FLUSH(base) do { if (lane == 0) { for (int pk = 0; pk < 7; pk++) atomicAdd(&v.cnt->dbg[(base) + pk], pt_acc[pk]); \
    if (wid == 0) { atomicMax(&v.cnt->dbg[(base) + 7], __builtin_amdgcn_s_memtime() - pt_t0); atomicAdd(&v.cnt->dbg[(base) + 8], 1ull); } } } while (0)
#elif defined(SNF_WG_TRACE)
#define SNF_PT_DECL unsigned long long pt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long pt_t = wall_clock64();
#define SNF_PT(k) do { const unsigned long long pt_n = wall_clock64(); pt_acc[k] += pt_n - pt_t; pt_t = pt_n; } while (0)
#define SNF_PT_FLUSH(base) do { } while (0)
#else
#define SNF_PT_DECL
#define SNF_PT(k) do { } while (0)
#define SNF_PT_FLUSH(base) do { } while (0)
#endif

template <int SLOTS, int MAXPOS, int MAXOTHERS, int NW, int LCAP, int SCAP, int ECAP>
struct ConsLdsT {
  unsigned long long key[SLOTS];
  uint32_t pc[SLOTS];              // (position << 16) | occurrence count
  uint8_t kept[MAXOTHERS];
  uint32_t cnt[LCAP ? LCAP : 1];   // LDS vote: per column four 8-bit counters of the other reads' bases (code 0 A, 1 C, 2 T, 3 G)
  uint32_t esc[ECAP ? ECAP : 1];   // votes with any other byte: column << 8 | byte
  uint32_t n_esc;
  alignas(16) uint8_t best[LCAP ? LCAP + 32 : 16];   // the best read, staged
  struct Wave {
    uint16_t ai[MAXPOS];           // candidates, then accepted anchors: position in best
    uint16_t aj[MAXPOS];           //                                    position in the read
    uint32_t seg_pref[MAXPOS];     // over the copied segments up to t: matches against best (low half) | columns written (high half)
    uint16_t seg_len[LCAP ? 1 : MAXPOS];   // rows instance: clipped advance (columns written)
    uint8_t seg_flag[LCAP ? 1 : MAXPOS];   //                0 dashes, 1 copy
    alignas(16) uint8_t s[SCAP ? SCAP : 16];          // the other read, staged (SMALL)
  } w[NW];
};

// 16 bytes per lane from an arbitrarily aligned global address into 16-byte aligned LDS
typedef uint4 __attribute__((aligned(1))) u128_unaligned;
SNF_D void stage16(uint8_t* dst_lds, const uint8_t* src, int nbytes, int tid, int nthreads) {
  for (int o = tid * 16; o < nbytes; o += nthreads * 16) *(uint4*)(dst_lds + o) = *(const u128_unaligned*)(src + o);
}

typedef uint64_t __attribute__((aligned(1))) u64_unaligned;
SNF_D uint64_t load_u64(const uint8_t* p) { return *(const u64_unaligned*)p; }  // pool has >= 16 B of slack
// 8 bytes at an arbitrary byte address.  In LDS an unaligned ds_read_b64 is served lane by lane (SQ_LDS_UNALIGNED_STALL was 80 %
// of the LDS pipe's busy cycles of these kernels - and the LDS pipe, shared by all waves of the CU, was their bottleneck):
// there the two aligned words around the address are read (one ds_read2_b64) and funnel-shifted.  The staged arrays have
// >= 16 B of slack behind what is ever addressed.
typedef uint64_t __attribute__((may_alias, aligned(8))) u64_alias;
template <bool IN_LDS> SNF_D uint64_t ld8(const uint8_t* p) {
  if constexpr (IN_LDS) {
    const uintptr_t a = (uintptr_t)p;
    const u64_alias* q = (const u64_alias*)(a & ~(uintptr_t)7);
    const uint64_t lo = q[0], hi = q[1];
    const int sh = (int)(a & 7) * 8;
    return sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
  } else return load_u64(p);
}
// number of equal bytes among the first n (<= 8) bytes of two little-endian words
SNF_D int eq_bytes(uint64_t a, uint64_t b, int n) {
  uint64_t x = a ^ b;
  if (n < 8) x |= ~0ull << (8 * n);                       // bytes past n count as different
  uint64_t t = (x & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full;
  t = ~(t | x | 0x7f7f7f7f7f7f7f7full);                   // 0x80 in every zero byte of x
  return __builtin_popcountll(t);
}
template <bool A_LDS, bool B_LDS> SNF_D int count_eq(const uint8_t* a, const uint8_t* b, int n) {
  int m = 0;
  for (int q = 0; q < n; q += 8) m += eq_bytes(ld8<A_LDS>(a + q), ld8<B_LDS>(b + q), n - q < 8 ? n - q : 8);
  return m;
}
// bytes [k, k + 8) of a 24-byte window held in three words (zero beyond the window), 0 <= k < 24
SNF_D uint64_t win24(uint64_t w0, uint64_t w1, uint64_t w2, int k) {
  const int wi = k >> 3, sh = (k & 7) * 8;
  const uint64_t lo = wi == 0 ? w0 : wi == 1 ? w1 : w2;
  const uint64_t hi = wi == 0 ? w1 : wi == 1 ? w2 : 0ull;
  return sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
}
// equal bytes of window[k0, k0 + n) and b[0, n)   (k0 + n <= 24)
template <bool B_LDS> SNF_D int count_eq_win(uint64_t w0, uint64_t w1, uint64_t w2, int k0, const uint8_t* b, int n) {
  int m = 0;
  for (int q = 0; q < n; q += 8) m += eq_bytes(win24(w0, w1, w2, k0 + q), ld8<B_LDS>(b + q), n - q < 8 ? n - q : 8);
  return m;
}
// injective key of the klen (<= 7) bytes in w; only has to agree between this kernel's table build and lookups
SNF_D unsigned long long kmer_key_le(unsigned long long w, int klen) { return w & ((1ull << (8 * klen)) - 1ull); }

// Wave scans through DPP (row shifts inside the rows of 16 lanes, row_bcast:15 / :31 for the row totals): six VALU
// instructions instead of six LDS-crossbar round trips (ds_bpermute) per scan.  All 64 lanes must be active.
SNF_D int wave_max_incl(int x, int lane) {
  (void)lane;
#define SNF_DPP_MAX(ctrl, rows) { const int y = __builtin_amdgcn_update_dpp(x, x, ctrl, rows, 0xf, false); x = y > x ? y : x; }
  SNF_DPP_MAX(0x111, 0xf) SNF_DPP_MAX(0x112, 0xf) SNF_DPP_MAX(0x114, 0xf) SNF_DPP_MAX(0x118, 0xf) SNF_DPP_MAX(0x142, 0xa) SNF_DPP_MAX(0x143, 0xc)
#undef SNF_DPP_MAX
  return x;
}
SNF_D uint32_t wave_sum_incl_u32(uint32_t x, int lane) {
  (void)lane;
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);   // row_shr:1
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);   // row_shr:2
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);   // row_shr:4
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);   // row_shr:8
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
  return x;
}
// the value of the lane before (lane 0: `first`)
SNF_D int wave_prev(int x, int first) {
  // wave_shr:1 is not available on gfx9+; row_shr:1 leaves lanes 0, 16, 32, 48 to be patched from lanes 15, 31, 47
  int y = __builtin_amdgcn_update_dpp(first, x, 0x111, 0xf, 0xf, false);
  const int l15 = __builtin_amdgcn_readlane(x, 15), l31 = __builtin_amdgcn_readlane(x, 31), l47 = __builtin_amdgcn_readlane(x, 47);
  const int lane = (int)(threadIdx.x & 63);
  if (lane == 16) y = l15;
  if (lane == 32) y = l31;
  if (lane == 48) y = l47;
  return y;
}

SNF_D int64_t rfl64(int64_t x) {  // wave-uniform 64-bit value -> SGPR pair
  return (int64_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(x >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)x));
}

#define SNF_ACTG 0x47544341u  /* byte z = the base with code z, code(c) = (c >> 1) & 3: A 0, C 1, T 2, G 3 */

// CLS: 1 SMALL, 2 LARGE, 4 ROWS (cons_class_of)
// NW: waves per workgroup (= per call).  4: the reads of a call are spread over four waves; 1: a call is one wave's work
// (no workgroup barriers, no waves idling while the wave with one read more finishes, four times as many calls in flight)
// LCAP > 0: LDS-vote instance for calls of at most LCAP columns; SCAP > 0: the other read is staged in LDS as well;
// ECAP: capacity of the escape list (votes with a byte other than A/C/G/T) - a call that needs more is handed to the
// ROWS instance through work list 7
template <int CLS, int SLOTS, int MAXPOS, int MAXOTHERS, int MINW, int NW = 4, int LCAP = 0, int SCAP = 0, int ECAP = 0>
__global__ void __launch_bounds__(64 * NW, MINW) e45w_consensus(const View v, int64_t n_unused) {
  IT_SCOPE(CLS == 1 ? 10 : CLS == 2 ? 11 : 14)
  typedef ConsLdsT<SLOTS, MAXPOS, MAXOTHERS, NW, LCAP, SCAP, ECAP> Lds;
  constexpr int NT = 64 * NW;
  constexpr bool LV = LCAP > 0;
  __shared__ Lds lds;
  constexpr int ROUNDS = MAXPOS / 64;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform -> SGPRs
  const int klen = v.cfg.consensus_kmer_len, maxshift = klen;
  // SMALL walks list 1; LARGE walks lists 2..5 (heaviest first) as one index space; ROWS walks list 7
  const int64_t n1 = (int64_t)v.cnt->n_cls[CLS == 1 ? 1 : CLS == 2 ? 2 : 7], n2 = CLS != 2 ? 0 : (int64_t)v.cnt->n_cls[3];
  const int64_t n3 = CLS != 2 ? 0 : (int64_t)v.cnt->n_cls[4], n4 = CLS != 2 ? 0 : (int64_t)v.cnt->n_cls[5];
  const int64_t n_items = n1 + n2 + n3 + n4;
  unsigned long long bytes_acc = 0;  // algorithmic bytes this block processed, one atomic at the end
  auto item_cid = [&](int64_t it) -> int32_t {
    if (CLS != 2 || it < n1) return v.cls_list[CLS == 1 ? 1 : CLS == 2 ? 2 : 7][it];
    if (it < n1 + n2) return v.cls_list[3][it - n1];
    if (it < n1 + n2 + n3) return v.cls_list[4][it - n1 - n2];
    return v.cls_list[5][it - n1 - n2 - n3];
  };
  // normally one call per workgroup (the hardware dispatcher balances the unequal calls best, see the launch site); when the
  // grid is capped the workgroup strides, and the descriptor of its next call is requested while the current one is processed
  int32_t cid_next = (int64_t)blockIdx.x < n_items ? item_cid(blockIdx.x) : 0;
  ConsDesc d_next = (int64_t)blockIdx.x < n_items ? v.cdesc[cid_next] : ConsDesc{};
  for (int64_t it = blockIdx.x; it < n_items; it += gridDim.x) {
    const int32_t cid = cid_next;
    const ConsDesc d = d_next;   // one record: no pointer chasing before the first useful load
    if (it + gridDim.x < n_items) { cid_next = item_cid(it + gridDim.x); d_next = v.cdesc[cid_next]; }
    // everything per call is wave-uniform: keep it in SGPRs (the compiler cannot prove it for values loaded from global
    // memory, and the kernel's occupancy is bound by VGPRs)
    SNF_PT_DECL
#ifdef SNF_WG_TRACE
    const unsigned long long wg_t0 = wall_clock64();
#endif
    const int L = __builtin_amdgcn_readfirstlane(d.L);   // < 65000 (cons_class): 32-bit column arithmetic throughout
    const int32_t n_others = __builtin_amdgcn_readfirstlane(d.n_others);
    const uint8_t* Bg = v.pool + rfl64(d.best_off);
    uint8_t* alt = alt_base(v) + rfl64(d.alt_off);
    const int skip = __builtin_amdgcn_readfirstlane(d.skip);
    const int64_t r0 = rfl64(d.read_off);
    // pool offset / length of every other read of the call, one entry per lane and slot: in flight while the best read is
    // staged and its table built (a read then takes them by a lane broadcast instead of a dependent scalar load)
    constexpr int NS = (MAXOTHERS + 63) / 64;
    int64_t my_off[NS]; int32_t my_len[NS];
#pragma unroll
    for (int sx = 0; sx < NS; sx++) {
      const int idx = sx * 64 + lane;
      my_off[sx] = idx < n_others ? v.crl_off[r0 + idx] : 0; my_len[sx] = idx < n_others ? v.crl_len[r0 + idx] : 0;
    }
    __syncthreads();
    // ---- anchor table of the best read (consensus.py:289-299): k-mers seen exactly once
    for (int s = tid; s < SLOTS; s += NT) { lds.key[s] = SNF_KEY_EMPTY; lds.pc[s] = 0; }
    if constexpr (LV) {
      stage16(lds.best, Bg, L + 8, tid, NT);                 // the pool has >= 16 B of slack behind every sequence
      for (int q = tid; q < L; q += NT) lds.cnt[q] = 0;
      if (tid == 0) lds.n_esc = 0;
    }
    __syncthreads();
    const uint8_t* B = LV ? (const uint8_t*)lds.best : Bg;
    const int npos = (int)cons_npos(L, klen, skip);
    for (int p = tid; p < npos; p += NT) {
      const int i = p * skip;
      const unsigned long long kk = kmer_key_le(ld8<LV>(B + i), klen);
      int64_t sl = kmer_slot(kk, SLOTS);
      for (;;) {
        const unsigned long long old = atomicCAS(&lds.key[sl], SNF_KEY_EMPTY, kk);
        if (old == SNF_KEY_EMPTY || old == kk) break;
        sl = (sl + 1) & (SLOTS - 1);
      }
      if ((atomicAdd(&lds.pc[sl], 1u) & 0xffffu) == 0) atomicOr(&lds.pc[sl], (uint32_t)i << 16);  // position of the 1st sighting
    }
    __syncthreads();
    // The anchor keys by sampled position.  The other reads never probe the table: an anchor needs |i - j| <= maxshift with i
    // and j on the same sampling grid, so position p of a read can only anchor at positions p - dmax .. p + dmax of the best
    // read (dmax = maxshift / skip, 0 once skip > maxshift).  kb[KPAD + p] = the k-mer key at sampled position p of the best
    // read if it is seen exactly once there, else a word no key equals (keys have at most 7 bytes) - also on KPAD >= dmax
    // positions either side: a probe is one aligned 8-byte LDS read and one compare, no bounds, no second look-up.  The table
    // is only needed up to here: kb takes its storage.
    constexpr int KPAD = 8, KIT = (MAXPOS + NT - 1) / NT;
    static_assert(SLOTS >= MAXPOS + 2 * KPAD, "the anchor keys by position reuse the table's storage");
    unsigned long long* const kb = lds.key;
    {
      unsigned long long kbv[KIT];
#pragma unroll
      for (int k = 0; k < KIT; k++) {
        const int p = k * NT + tid;
        kbv[k] = SNF_KEY_EMPTY;
        if (p < npos) {
          const unsigned long long kk = kmer_key_le(ld8<LV>(B + p * skip), klen);
          int sl = (int)kmer_slot(kk, SLOTS);
          while (lds.key[sl] != kk) sl = (sl + 1) & (SLOTS - 1);
          if ((lds.pc[sl] & 0xffffu) == 1u) kbv[k] = kk;
        }
      }
      __syncthreads();   // every look-up is done
#pragma unroll
      for (int k = 0; k < KIT; k++) { const int p = k * NT + tid; if (p < MAXPOS) kb[KPAD + p] = kbv[k]; }
      if (tid < KPAD) { kb[tid] = SNF_KEY_EMPTY; kb[KPAD + MAXPOS + tid] = SNF_KEY_EMPTY; }
    }
    __syncthreads();
    SNF_PT(0);   // descriptor, staging, table build
#ifdef SNF_WG_TRACE
    const unsigned long long wg_t1 = wall_clock64();
#endif
    typename Lds::Wave& W = lds.w[wid];
    uint8_t* rows = LV ? nullptr : v.aln + rfl64(d.aln_off);
    // Geometry of other read r: where it lives, how far the sampled positions go, how many bytes any phase touches.
    // candidates in read order: a sampled k-mer is an anchor iff it is in the table and |i - j| <= maxshift
    auto read_geom = [&](int32_t r, const uint8_t*& Sg, int& SL, int& jlim, int& P, int& ns) {
      int64_t s_off = 0; int32_t s_len = 0;
#pragma unroll
      for (int sx = 0; sx < NS; sx++) if (sx == (r >> 6)) { s_off = __shfl(my_off[sx], r & 63, 64); s_len = __shfl(my_len[sx], r & 63, 64); }
      Sg = v.pool + rfl64(s_off);
      SL = __builtin_amdgcn_readfirstlane(s_len);
      jlim = SL - klen;                                         // j < SL - klen
      if (L - klen + maxshift < jlim) jlim = L - klen + maxshift;   // an anchor needs i <= L-klen-1, |i-j| <= maxshift
      P = jlim <= 0 ? 0 : (jlim + skip - 1) / skip;             // <= npos + 2 < MAXPOS
      // every byte any phase reads lies below jlim + klen + 8 (k-mer words, segment compares, copied bases)
      ns = jlim + klen + 8; if (ns > SL + 8) ns = SL + 8; if (ns < 0) ns = 0;
    };
    // The first thing a read needs - its bytes (SMALL: staged in LDS) or its sampled k-mer words (read in HBM) - is requested
    // one read ahead: measured with s_memtime stamps, waiting for exactly these loads was 57 % (SMALL) and 32 % (LARGE)
    // of the waves' time when every read fetched them on demand.
    static_assert(SCAP <= 1024, "one 16-byte load per lane covers the staged read");
    uint4 pre_s = make_uint4(0, 0, 0, 0);
    unsigned long long pre_kw[SCAP > 0 ? 1 : ROUNDS];
    {
      const uint8_t* Sg0 = v.pool; int SL0 = 0, jl0 = 0, P0 = 0, ns0 = 0;
      if (wid < n_others) read_geom(wid, Sg0, SL0, jl0, P0, ns0);
      if constexpr (SCAP > 0) { if (lane * 16 < ns0) pre_s = *(const u128_unaligned*)(Sg0 + lane * 16); }
      else {
#pragma unroll
        for (int rd = 0; rd < ROUNDS; rd++) { const int p = rd * 64 + lane; pre_kw[rd] = (p < P0) ? load_u64(Sg0 + p * skip) : 0ull; }
      }
    }
    for (int32_t r = wid; r < n_others; r += NW) {
      const uint8_t* Sg; int SL, jlim, P, ns;
      read_geom(r, Sg, SL, jlim, P, ns);
      // ---- 1. the read's k-mer words (requested one read ago), then the request for the next read
      unsigned long long kw[ROUNDS];
      if constexpr (SCAP > 0) {
        if (lane * 16 < ns) *(uint4*)(W.s + lane * 16) = pre_s;
        __builtin_amdgcn_wave_barrier();
      } else {
#pragma unroll
        for (int rd = 0; rd < ROUNDS; rd++) kw[rd] = pre_kw[rd];
      }
      const uint8_t* S = SCAP > 0 ? (const uint8_t*)W.s : Sg;
      if (r + NW < n_others) {
        const uint8_t* Sgn; int SLn, jln, Pn, nsn;
        read_geom(r + NW, Sgn, SLn, jln, Pn, nsn);
        if constexpr (SCAP > 0) { if (lane * 16 < nsn) pre_s = *(const u128_unaligned*)(Sgn + lane * 16); }
        else {
#pragma unroll
          for (int rd = 0; rd < ROUNDS; rd++) { const int p = rd * 64 + lane; pre_kw[rd] = (p < Pn) ? load_u64(Sgn + p * skip) : 0ull; }
        }
      }
#ifdef SNF_WG_TRACE
      SNF_PT(7);
#endif
      if constexpr (SCAP > 0) {
#pragma unroll
        for (int rd = 0; rd < ROUNDS; rd++) {                          // all loads in flight before the first use
          const int p = rd * 64 + lane;
          kw[rd] = (p < P) ? ld8<true>(S + p * skip) : 0ull;
        }
      }
      int ncand = 0;
      const int dmax = maxshift / skip;
#pragma unroll
      for (int rd = 0; rd < ROUNDS; rd++) {
        if (rd * 64 >= P) break;
        const int p = rd * 64 + lane, j = p * skip;
        const unsigned long long kk = kmer_key_le(kw[rd], klen);
        int ci_ = -1;
        // (one candidate whenever skip > maxshift; at most one matches: anchors are unique)
#ifdef SNF_CONS_PROBE_LOOP
        if (false) {
#else
        if (dmax <= 2) {   // the usual cases: all reads in flight at once
#endif
          const unsigned long long* kq = kb + KPAD + p;
          const unsigned long long k0 = kq[-2], k1 = kq[-1], k2 = kq[0], k3 = kq[1], k4 = kq[2];
          if (k2 == kk) ci_ = j;
          if (dmax >= 1) { if (k1 == kk) ci_ = j - skip; if (k3 == kk) ci_ = j + skip; }
          if (dmax == 2) { if (k0 == kk) ci_ = j - 2 * skip; if (k4 == kk) ci_ = j + 2 * skip; }
        } else {
          for (int dd = -dmax; dd <= dmax; dd++)
            if (kb[KPAD + p + dd] == kk) ci_ = (p + dd) * skip;
        }
        if (p >= P) ci_ = -1;
        const unsigned long long mk = __ballot(ci_ >= 0);
        if (ci_ >= 0) { const int w = ncand + __builtin_popcountll(mk & ((1ull << lane) - 1ull)); W.ai[w] = (uint16_t)ci_; W.aj[w] = (uint16_t)j; }
        ncand += __builtin_popcountll(mk);
      }
      __builtin_amdgcn_wave_barrier();
      SNF_PT(1);   // k-mer words, probes, candidate compaction
      // ---- 2. monotone chain: accept iff i > every earlier candidate's i (== last accepted i)
      int na = 0, runmax = -1;
      for (int c0 = 0; c0 < ncand; c0 += 64) {
        const int cidx = c0 + lane;
        const int i = cidx < ncand ? (int)W.ai[cidx] : -1, j = cidx < ncand ? (int)W.aj[cidx] : 0;
        int pm = wave_max_incl(i, lane);
        int prev = wave_prev(pm, -1);
        if (lane == 0) prev = -1;
        if (runmax > prev) prev = runmax;
        const bool acc = cidx < ncand && i > prev;
        const unsigned long long mk = __ballot(acc);
        __builtin_amdgcn_wave_barrier();
        if (acc) { const int w = na + __builtin_popcountll(mk & ((1ull << lane) - 1ull)); W.ai[w] = (uint16_t)i; W.aj[w] = (uint16_t)j; }
        na += __builtin_popcountll(mk);
        const int tot = __builtin_amdgcn_readlane(pm, 63);
        if (tot > runmax) runmax = tot;
        __builtin_amdgcn_wave_barrier();
      }
      SNF_PT(2);   // monotone chain
      // ---- 3. segments between consecutive anchors
      // The comparisons (and later the votes) run one lane per SAMPLING STEP of the read, not per segment: a step is `skip`
      // bases wherever it lies, so every lane has the same amount of work whatever the distances between the anchors are
      // (one lane per segment left a wave waiting for its longest segment - a segment of 2 000 bases between two anchors
      // held a workgroup for 100 us).  Step p belongs to segment t when aj[t-1] <= p * skip < aj[t]; the partial counts of
      // a segment's steps meet in one LDS word per segment.
      const int i0 = __builtin_amdgcn_readfirstlane(na ? (int)W.ai[0] : 0), j0 = __builtin_amdgcn_readfirstlane(na ? (int)W.aj[0] : 0);
      const int c_first = na ? ((j0 > 0) ? i0 : 0) : 0;   // '-' * i only when j > 0 (consensus.py:316-318)
      const int j_end = __builtin_amdgcn_readfirstlane(na ? (int)W.aj[na - 1] : 0);
      int span = 0;
      // Read in HBM (SCAP == 0): the skip + 1 <= 24 bytes a step is compared and copied from are the step's k-mer word (already
      // here) and up to two more words, requested for all steps of the read at once and kept in registers through the vote
      constexpr int WR = SCAP == 0 ? ROUNDS : 1;
      // Read staged in LDS (SCAP > 0, the SMALL class: skip <= 7 by cons_class_of): the k-mer word alone holds the skip + 1 bytes of a step
      const bool use_win = SCAP == 0 ? skip <= 23 : true;
      uint64_t sw1[WR] = {}, sw2[WR] = {};
      if constexpr (SCAP == 0) {
#pragma unroll
        for (int rd = 0; rd < ROUNDS; rd++) {
          const int p = rd * 64 + lane;
          sw1[rd] = sw2[rd] = 0;
          if (use_win && p + 1 < P) {   // (a step inside a segment ends at or before the last sampled position)
            if (skip >= 8) sw1[rd] = load_u64(S + p * skip + 8);
            if (skip >= 16) sw2[rd] = load_u64(S + p * skip + 16);
          }
        }
      }
      // step -> segment: every segment marks its first step, a running maximum carries the mark over the segment's steps
#pragma unroll
      for (int rd = 0; rd < ROUNDS; rd++) { const int p = rd * 64 + lane; if (p < P) W.seg_pref[p] = 0u; }
      __builtin_amdgcn_wave_barrier();
      const float rskip = 1.0f / (float)skip;   // positions are multiples of skip below 65536: the rounded product is the exact quotient
#pragma unroll
      for (int it = 0; it < ROUNDS; it++) {
        const int t = 1 + it * 64 + lane;
        if (t < na) W.seg_pref[(int)((float)W.aj[t - 1] * rskip + 0.5f)] = (uint32_t)t;
      }
      __builtin_amdgcn_wave_barrier();
      int tseg[ROUNDS];
      {
        int carry = 0;
#pragma unroll
        for (int rd = 0; rd < ROUNDS; rd++) {
          tseg[rd] = 0;
          if (rd * 64 >= P) continue;
          const int p = rd * 64 + lane;
          int pm = wave_max_incl(p < P ? (int)W.seg_pref[p] : 0, lane);
          if (carry > pm) pm = carry;
          carry = __builtin_amdgcn_readlane(pm, 63);
          tseg[rd] = p * skip < j_end ? pm : 0;
        }
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < ROUNDS; it++) { const int t = 1 + it * 64 + lane; if (t < na) W.seg_pref[t] = 0u; }
      __builtin_amdgcn_wave_barrier();
#ifdef SNF_WG_TRACE
      SNF_PT(4);
#endif
      // geometry of segment t (same arithmetic for the step lanes and the segment lanes)
      auto seg_geom = [&](int t, int& li, int& lj, int& col, int& nfull, int& fwd_j) -> bool {
        li = W.ai[t - 1]; lj = W.aj[t - 1];
        const int i = W.ai[t], j = W.aj[t];
        col = c_first + (lj - j0); if (col > L) col = L;
        const int fwd_i = i - li; fwd_j = j - lj; nfull = fwd_j;
        if (col + fwd_j > L) fwd_j = L - col;
        return fwd_i == fwd_j && fwd_j > 0;      // copied (if it passes the identity test), else dashes
      };
#pragma unroll
      for (int rd = 0; rd < ROUNDS; rd++) {
        if (rd * 64 >= P) break;
        const int t = tseg[rd];
        if (t > 0) {
          int li, lj, col, nfull, fwd_j;
          if (seg_geom(t, li, lj, col, nfull, fwd_j)) {
            // identity on offsets off + 1 .. off + skip of the segment, column identity of the copied slice on off .. off + nc - 1
            const int x0 = (rd * 64 + lane) * skip, off = x0 - lj;
            const int nc = fwd_j - off < skip ? fwd_j - off : skip;
            int m, cm = 0;
            if (use_win) {
              m = count_eq_win<LV>(kw[rd], sw1[SCAP == 0 ? rd : 0], sw2[SCAP == 0 ? rd : 0], 1, B + li + off + 1, skip);
              if (nc > 0) cm = count_eq_win<LV>(kw[rd], sw1[SCAP == 0 ? rd : 0], sw2[SCAP == 0 ? rd : 0], 0, B + col + off, nc);
            } else {
              m = count_eq<(SCAP > 0), LV>(S + x0 + 1, B + li + off + 1, skip);
              if (nc > 0) cm = count_eq<(SCAP > 0), LV>(S + x0, B + col + off, nc);
            }
            atomicAdd(&W.seg_pref[t], (uint32_t)m | ((uint32_t)cm << 16));   // both sums stay below 65536 (a segment is shorter than the read)
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      uint32_t segn[ROUNDS];   // per segment of this lane: columns written | matches of the copied slice against best << 16; 0 = dashes
#pragma unroll
      for (int it = 0; it < ROUNDS; it++) {
        segn[it] = 0u;
        if (1 + it * 64 >= na) continue;
        const int t = 1 + it * 64 + lane;
        if (t < na) {
          int li, lj, col, nfull, fwd_j;
          if (seg_geom(t, li, lj, col, nfull, fwd_j)) {
            span += nfull;
            const uint32_t acc = W.seg_pref[t];
            if ((double)(acc & 0xffffu) / (double)nfull >= 0.5) segn[it] = (uint32_t)fwd_j | (acc & 0xffff0000u);
          }
        }
      }
      __builtin_amdgcn_wave_barrier();   // (the run filter reuses seg_pref)
      span = __builtin_amdgcn_readlane((int)wave_sum_incl_u32((uint32_t)span, lane), 63);
      SNF_PT(3);   // windows + segment compares
      // ---- 4. run filter over maximal groups of consecutive copied segments (consensus.py:343-360): a group stays iff
      // more than half of its columns agree with the best read and more than five do.  One lane per segment: prefix sums
      // of (matches, columns) over the copied segments go to LDS, the ballots of the copy flags give every lane the first
      // and last segment of its group, two LDS reads give the group's sums.
      unsigned long long fmask[ROUNDS];
      {
        uint32_t carry = 0;
#pragma unroll
        for (int it = 0; it < ROUNDS; it++) {
          fmask[it] = 0;
          if (1 + it * 64 >= na) continue;
          const int t = 1 + it * 64 + lane;
          const uint32_t n = segn[it] & 0xffffu, cmv = segn[it] >> 16;
          fmask[it] = __ballot(segn[it] != 0u);                      // lanes past na hold 0: they end a group
          const uint32_t pre = wave_sum_incl_u32(cmv | (n << 16), lane) + carry;   // both halves stay below 65536 (disjoint columns of one row)
          if (t < na) W.seg_pref[t] = pre;
          carry = (uint32_t)__builtin_amdgcn_readlane((int)pre, 63);
        }
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < ROUNDS; it++) {
        if (1 + it * 64 >= na) break;
        if (segn[it] != 0u) {
          int fs = -1, fe = -1;   // flat index (t - 1) of the first / last segment of this lane's group
#pragma unroll
          for (int k = ROUNDS - 1; k >= 0; k--) {
            if (k > it || fs >= 0) continue;
            unsigned long long zb = ~fmask[k];
            if (k == it) zb &= (1ull << lane) - 1ull;
            if (zb) fs = k * 64 + 64 - __builtin_clzll(zb);
          }
          if (fs < 0) fs = 0;
#pragma unroll
          for (int k = 0; k < ROUNDS; k++) {
            if (k < it || fe >= 0) continue;
            unsigned long long za = ~fmask[k];
            if (k == it) za &= ~((2ull << lane) - 1ull);
            if (za) fe = k * 64 + __builtin_ctzll(za) - 1;
          }
          if (fe < 0) fe = na - 2;   // (na < MAXPOS: a lane past the last segment always ends the group before this)
          const uint32_t hi = W.seg_pref[fe + 1], lo = fs ? W.seg_pref[fs] : 0u;   // inclusive prefix at t = fe + 1, at t = fs (flat fs - 1)
          const int ident = (int)((hi & 0xffffu) - (lo & 0xffffu)), len = (int)((hi >> 16) - (lo >> 16));
          if (!((double)ident / (double)len > 0.5 && ident > 5)) segn[it] = 0u;
        }
        if constexpr (!LV) {
          const int t = 1 + it * 64 + lane;
          if (t < na) { W.seg_len[t] = (uint16_t)(segn[it] & 0xffffu); W.seg_flag[t] = segn[it] != 0u ? 1 : 0; }
        }
      }
      __builtin_amdgcn_wave_barrier();
      SNF_PT(4);   // run filter
      // a read whose copied span is <= 20 % of the best read is dropped (consensus.py:361-363): it has no vote
      const bool keep_row = (double)span / (double)L > 0.2;
      if constexpr (LV) {
        // ---- 5. votes: one lane per sampling step again; the bases a kept segment copied from this step go to the counters of
        // their columns (a base at position x of the read sits in column c_first + (x - j0))
        if (keep_row) {
#pragma unroll
          for (int it = 0; it < ROUNDS; it++) { const int t = 1 + it * 64 + lane; if (t < na) W.seg_pref[t] = segn[it] & 0xffffu; }   // columns written, 0: dashes
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int rd = 0; rd < ROUNDS; rd++) {
            if (rd * 64 >= P) break;
            const int t = tseg[rd];
            const int n = t > 0 ? (int)W.seg_pref[t] : 0;
            if (n > 0) {
              const int x0 = (rd * 64 + lane) * skip, off = x0 - (int)W.aj[t - 1];
              const int nc = n - off < skip ? n - off : skip;
              const int col = c_first + (x0 - j0);
              for (int o8 = 0; o8 < nc; o8 += 8) {
                unsigned long long w8 = use_win ? win24(kw[rd], sw1[SCAP == 0 ? rd : 0], sw2[SCAP == 0 ? rd : 0], o8) : ld8<(SCAP > 0)>(S + x0 + o8);
                const int m = nc - o8 < 8 ? nc - o8 : 8;
                for (int o = 0; o < m; o++, w8 >>= 8) {
                  const uint32_t c = (uint32_t)(w8 & 0xffull), cd = (c >> 1) & 3u;
                  if (((SNF_ACTG >> (8 * cd)) & 0xffu) == c) atomicAdd(&lds.cnt[col + o8 + o], 1u << (8 * cd));
                  else { const uint32_t e = atomicAdd(&lds.n_esc, 1u); if (e < (uint32_t)ECAP) lds.esc[e] = ((uint32_t)(col + o8 + o) << 8) | c; }
                }
              }
            }
          }
        }
        if (lane == 0) lds.kept[r] = keep_row ? 1 : 0;
      } else {
        // ---- 5. write the row, column-parallel
        uint8_t* row = rows + (int64_t)r * L;
        int c_last = c_first;
        if (na) { c_last = c_first + (__builtin_amdgcn_readfirstlane((int)W.aj[na - 1]) - j0); if (c_last > L) c_last = L; }
        for (int q0 = 0; q0 < L && keep_row; q0 += 64) {
          const int q = q0 + lane;
          if (q < L) {
            uint8_t out = '-';
            if (na > 1 && q >= c_first && q < c_last) {
              int lo2 = 1, hi2 = na - 1;  // last segment t with seg_col[t] <= q
              // segment t starts at column c_first + (aj[t-1] - j0) (q < c_last <= L, so the clip at L never matters here)
              while (lo2 < hi2) { const int mid = (lo2 + hi2 + 1) >> 1; if (c_first + ((int)W.aj[mid - 1] - j0) <= q) lo2 = mid; else hi2 = mid - 1; }
              const int t = lo2;
              const int off = q - (c_first + ((int)W.aj[t - 1] - j0));
              if (W.seg_flag[t] && off < W.seg_len[t]) out = S[W.aj[t - 1] + off];
            }
            row[q] = out;
          }
        }
        if (lane == 0) { const uint8_t k = keep_row ? 1 : 0; v.aln_kept_w[r0 + r] = k; lds.kept[r] = k; }
      }
      __builtin_amdgcn_wave_barrier();
      SNF_PT(5);   // votes into the counters
    }
    // ---- column vote (consensus.py:365-380)
#ifdef SNF_WG_TRACE
    const unsigned long long wg_t2 = wall_clock64();
#endif
    __syncthreads();
#ifdef SNF_WG_TRACE
    const unsigned long long wg_t3 = wall_clock64();
#endif
    int nkept = 0;
    for (int32_t r = 0; r < n_others; r++) nkept += lds.kept[r];
    nkept = __builtin_amdgcn_readfirstlane(nkept);
    if constexpr (LV) {
      const int n_esc = __builtin_amdgcn_readfirstlane((int)lds.n_esc);
      if (n_esc > ECAP) {
        // more odd characters than the escape list holds: the ROWS instance redoes this call (work list 7)
        if (tid == 0) { const unsigned long long slot = atomicAdd(&v.cnt->n_cls[7], 1ull); v.cls_list[7][slot] = cid; }
        continue;
      }
      bytes_acc += (unsigned long long)((int64_t)n_others + 2) * (unsigned long long)L;
      for (int q = tid; q < L; q += NT) alt[q] = vote_column(lds.cnt[q], lds.esc, n_esc, q, lds.best[q], nkept);
      SNF_PT(6);   // barrier wait + column vote + ALT stores
      SNF_PT_FLUSH(CLS == 1 ? 0 : 16);
#ifdef SNF_WG_TRACE
      __syncthreads();
      const unsigned long long wg_t4 = wall_clock64();
      __syncthreads();
      if (tid == 0) { v.wgtrace[2 * (int64_t)cid] = wg_t0;
        v.wgtrace[2 * ((int64_t)cid + (1 << 19)) + 1] = (pt_acc[1] / 10) | ((pt_acc[4] / 10) << 12) | ((pt_acc[3] / 10) << 24) | ((pt_acc[7] / 10) << 36) | ((pt_acc[5] / 10) << 48);
        v.wgtrace[2 * ((int64_t)cid + (1 << 19))] = (wg_t1 - wg_t0) | ((wg_t2 - wg_t1) << 16) | ((wg_t3 - wg_t2) << 32) | ((wg_t4 - wg_t3) << 48);
        v.wgtrace[2 * ((int64_t)cid + (1 << 20))] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 15u) << 32);   // HW_ID | XCC_ID
        v.wgtrace[2 * (int64_t)cid + 1] = ((wall_clock64() - wg_t0) << 32) | ((unsigned long long)CLS << 28) | ((unsigned long long)(n_others & 0xff) << 16) | (unsigned long long)(L & 0xffff); }
#endif
      continue;
    }
    bytes_acc += (unsigned long long)((int64_t)n_others + 2) * (unsigned long long)L;
    const double maxal = (double)(1 + nkept);
    for (int q = tid; q < L; q += NT) {
      const uint8_t bq = B[q];
      uint8_t out = bq;
      {  // fast path: every character of the column is one of A C G T -> four packed 16-bit counters, one pass
        const uint32_t ACTG = 0x47544341u;  // code (c >> 1) & 3: A 0, C 1, T 2, G 3
        int cd = (bq >> 1) & 3;
        bool plain = ((ACTG >> (8 * cd)) & 0xffu) == bq;
        unsigned long long cnt4 = 1ull << (16 * cd);
        int nv = 0;
#pragma unroll 8
        for (int32_t r = 0; r < n_others; r++) {  // no early exit: keeps the row loads independent
          if (!lds.kept[r]) continue;
          const uint8_t c = rows[(int64_t)r * L + q];
          if (c == '-') continue;
          cd = (c >> 1) & 3;
          plain &= ((ACTG >> (8 * cd)) & 0xffu) == c;
          cnt4 += 1ull << (16 * cd); nv++;
        }
        if (plain) {
          if (!(nv < 2 || (double)nv / maxal < 0.25)) {
            int c0 = -1, c1 = -1, k0 = -1, k1 = -1, nd = 0;
#pragma unroll
            for (int z = 0; z < 4; z++) {
              const int cntc = (int)((cnt4 >> (16 * z)) & 0xffffull);
              if (!cntc) continue;
              const int c = (int)((ACTG >> (8 * z)) & 0xffu);
              nd++;
              if (cntc > c0 || (cntc == c0 && c > k0)) { c1 = c0; k1 = k0; c0 = cntc; k0 = c; }
              else if (cntc > c1 || (cntc == c1 && c > k1)) { c1 = cntc; k1 = c; }
            }
            if (nd > 1 && c0 - c1 >= 3) out = (uint8_t)k0;
          }
          alt[q] = out;
          continue;
        }
      }
      {  // generic path: some character of this column is not A/C/G/T (rare)
        int nvotes = 0;
        for (int32_t r = 0; r < n_others; r++) if (lds.kept[r] && rows[(int64_t)r * L + q] != '-') nvotes++;
        if (!(nvotes < 2 || (double)nvotes / maxal < 0.25)) {
          // util.most_common([best]+votes): (count, char) descending; replace iff top beats the runner-up by >= 3
          int c0 = -1, c1 = -1, k0 = -1, k1 = -1, nd = 0;
          for (int32_t r = -1; r < n_others; r++) {
            uint8_t c;
            if (r < 0) c = bq;
            else { if (!lds.kept[r]) continue; c = rows[(int64_t)r * L + q]; if (c == '-') continue; }
            bool seen = (r >= 0 && c == bq);
            for (int32_t r2 = 0; r2 < r && !seen; r2++) if (lds.kept[r2] && rows[(int64_t)r2 * L + q] == c) seen = true;
            if (seen) continue;
            int cntc = (c == bq) ? 1 : 0;
            for (int32_t r2 = 0; r2 < n_others; r2++) if (lds.kept[r2] && rows[(int64_t)r2 * L + q] == c) cntc++;
            nd++;
            if (cntc > c0 || (cntc == c0 && (int)c > k0)) { c1 = c0; k1 = k0; c0 = cntc; k0 = c; }
            else if (cntc > c1 || (cntc == c1 && (int)c > k1)) { c1 = cntc; k1 = c; }
          }
          if (nd > 1 && c0 - c1 >= 3) out = (uint8_t)k0;
        }
      }
      alt[q] = out;
    }
  }
  if (tid == 0 && bytes_acc) atomicAdd(&v.stripes[((CLS == 4 ? 0 : CLS) * 64 + (blockIdx.x & 63)) * 16], bytes_acc);  // striped: summed by z1_results
}

// verbatim ALT of calls with fewer than consensus_min_reads other reads (postprocessing.py:65-66): one wave per call
__global__ void __launch_bounds__(64) e4c_copy(const View v, int64_t n_unused) {
  const int64_t n_items = (int64_t)v.cnt->n_cls[0];
  const int32_t* list = v.cls_list[0];
  unsigned long long bytes_acc = 0;
  for (int64_t it = blockIdx.x; it < n_items; it += gridDim.x) {
    const ConsDesc d = v.cdesc[list[it]];
    const uint8_t* B = v.pool + d.best_off;
    uint8_t* alt = alt_base(v) + d.alt_off;
    const int32_t full = d.L & ~15;     // 16 bytes per lane and step (unaligned vector accesses; the pool has slack behind every sequence), byte tail
    for (int32_t o = (int32_t)threadIdx.x * 16; o < full; o += 64 * 16) *(u128_unaligned*)(alt + o) = *(const u128_unaligned*)(B + o);
    if ((int32_t)threadIdx.x < d.L - full) alt[full + threadIdx.x] = B[full + threadIdx.x];
    bytes_acc += 2ull * (unsigned long long)d.L;
  }
  if (threadIdx.x == 0 && bytes_acc) atomicAdd(&v.stripes[(3 * 64 + (blockIdx.x & 63)) * 16], bytes_acc);
}

}  // namespace snf
