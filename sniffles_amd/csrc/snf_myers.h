// snf_myers.h - bit-parallel Myers / Hyyro block primitives for global (NW) unit-cost edit distance
// (see snf_myers.hip for the batched kernels; also used on demand by the combine kernel, snf_combine.hip).
#pragma once
#include "snf_exact.h"

namespace snf {

// bit-planes of up to 64 pattern bytes: planes[k] bit i = bit k of p[i]; *valid bit i = (i < cnt)
SNF_HD void block_planes(const uint8_t* p, int cnt, uint64_t planes[8], uint64_t* valid) {
  for (int k = 0; k < 8; k++) planes[k] = 0;
  for (int i = 0; i < cnt; i++) {
    uint8_t c = p[i];
    for (int k = 0; k < 8; k++) planes[k] |= (uint64_t)((c >> k) & 1) << i;
  }
  *valid = cnt >= 64 ? ~0ull : ((1ull << cnt) - 1ull);
}
SNF_HD uint64_t eq_mask(const uint64_t planes[8], uint64_t valid, uint8_t c) {
  uint64_t e = valid;
  for (int k = 0; k < 8; k++) e &= ((c >> k) & 1) ? planes[k] : ~planes[k];
  return e;
}
// one block, one column (Hyyro 2003 / edlib calculateBlock); hin, hout in {-1, 0, +1}
SNF_HD int advance_block(uint64_t& Pv, uint64_t& Mv, uint64_t Eq, int hin) {
  uint64_t Xv = Eq | Mv;
  if (hin < 0) Eq |= 1ull;
  uint64_t Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
  uint64_t Ph = Mv | ~(Xh | Pv);
  uint64_t Mh = Pv & Xh;
  int hout = 0;
  if (Ph >> 63) hout = 1;
  if (Mh >> 63) hout = -1;
  Ph <<= 1; Mh <<= 1;
  if (hin < 0) Mh |= 1ull; else if (hin > 0) Ph |= 1ull;
  Pv = Mh | ~(Xv | Ph);
  Mv = Ph & Xv;
  return hout;
}
// D[m][n] from the score at the bottom row of the last (padded) block
SNF_HD int64_t unpad_score(int64_t score, uint64_t Pv, uint64_t Mv, int pad_rows) {
  for (int i = 0; i < pad_rows; i++) {
    int bit = 63 - i;
    score -= (int64_t)((Pv >> bit) & 1) - (int64_t)((Mv >> bit) & 1);
  }
  return score;
}

// serial edit distance of two strings (any lengths); `carry` needs max(la, lb) bytes of scratch
SNF_HD int64_t ed_serial(const uint8_t* A, int64_t la, const uint8_t* B, int64_t lb, int8_t* carry) {
  const uint8_t *P = A, *T = B; int64_t m = la, n = lb;
  if (la > lb) { P = B; m = lb; T = A; n = la; }
  if (m == 0) return n;
  int64_t nb = (m + 63) / 64;
  int64_t score = nb * 64;
  uint64_t Pv = ~0ull, Mv = 0;
  for (int64_t blk = 0; blk < nb; blk++) {
    uint64_t planes[8], valid;
    int cnt = (int)(m - blk * 64 < 64 ? m - blk * 64 : 64);
    block_planes(P + blk * 64, cnt, planes, &valid);
    Pv = ~0ull; Mv = 0;
    bool last = blk + 1 == nb;
    for (int64_t j = 0; j < n; j++) {
      int hin = blk == 0 ? 1 : carry[j];
      int hout = advance_block(Pv, Mv, eq_mask(planes, valid, T[j]), hin);
      if (last) score += hout; else carry[j] = (int8_t)hout;
    }
  }
  return unpad_score(score, Pv, Mv, (int)(nb * 64 - m));
}

// ---------------------------------------------------------------------------------------------------------------------
// Banded form (Ukkonen's cut-off on the Myers / Hyyro blocks, as edlib does for its `k` argument): the distance if it is
// <= k, otherwise -1.  With m <= n (pattern rows, text columns) a path of cost <= k stays on the diagonals
// -kk <= j - i <= (n - m) + kk, kk = (k - (n - m)) / 2, so column j only needs the blocks that hold rows
// [j - (n - m) - kk, j + kk].  Cells outside the band count as "one more per step" (a block entering the band starts with
// Pv = all ones below the block above it, a block whose upper neighbour has left takes hin = +1): every value computed is
// the cost of a real path, hence >= the true distance, and equal to it whenever that is <= k.  k < 0: no cut-off.
struct EdBand { int64_t m, n, dl, kk; };
SNF_HD bool ed_band(int64_t m, int64_t n, int64_t k, EdBand* bd) {   // false: the lengths alone exceed k
  bd->m = m; bd->n = n; bd->dl = n - m;
  if (k < 0) { bd->kk = m; return true; }
  if (bd->dl > k) return false;
  bd->kk = (k - bd->dl) / 2;
  return true;
}
// first / last column (0-based) in which block b (rows 64b .. 64b+63) is inside the band
SNF_HD int64_t ed_band_cs(const EdBand& bd, int64_t b) { int64_t c = 64 * b - bd.kk; return c < 0 ? 0 : c; }
SNF_HD int64_t ed_band_ce(const EdBand& bd, int64_t b) { int64_t c = 64 * b + 63 + bd.dl + bd.kk; return c > bd.n - 1 ? bd.n - 1 : c; }

// 64-bit scratch words ed_serial_k needs for a pattern (= the shorter string) of m bytes: bit-planes, valid mask, Pv, Mv
SNF_HD int64_t ed_serial_scratch_words(int64_t m) { return 11 * ((m + 63) / 64) + 11; }

// serial banded distance, column-major: the block states stay in `scratch`, hout -> hin is a register (no per-column
// carry array); this is the form the emulation build and the thread-per-pair kernel run
SNF_HD int64_t ed_serial_k(const uint8_t* A, int64_t la, const uint8_t* B, int64_t lb, int64_t k, uint64_t* scratch) {
  const uint8_t *P = A, *T = B; int64_t m = la, n = lb;
  if (la > lb) { P = B; m = lb; T = A; n = la; }
  EdBand bd;
  if (!ed_band(m, n, k, &bd)) return -1;
  if (m == 0) return n;   // (n <= k, or no cut-off)
  const int64_t nb = (m + 63) / 64;
  uint64_t *planes = scratch, *valid = scratch + 8 * nb, *Pv = scratch + 9 * nb, *Mv = scratch + 10 * nb;
  for (int64_t b = 0; b < nb; b++) {
    const int cnt = (int)(m - b * 64 < 64 ? m - b * 64 : 64);
    block_planes(P + b * 64, cnt, planes + 8 * b, &valid[b]);
  }
  int64_t lastb = -1, score = 0;     // score = D at the bottom row of block `lastb` in the current column
  for (int64_t j = 0; j < n; j++) {
    int64_t lo = j - bd.dl - bd.kk, hi = j + bd.kk;
    if (lo < 0) lo = 0;
    if (hi > m - 1) hi = m - 1;
    const int64_t fb = lo / 64, lbk = hi / 64;
    while (lastb < lbk) { lastb++; Pv[lastb] = ~0ull; Mv[lastb] = 0; score += 64; }   // enters the band: +1 per row below
    int hin = 1;                                                                          // row 0, or the block above has left
    for (int64_t b = fb; b <= lbk; b++) hin = advance_block(Pv[b], Mv[b], eq_mask(planes + 8 * b, valid[b], T[j]), hin);
    score += hin;
  }
  const int64_t d = unpad_score(score, Pv[nb - 1], Mv[nb - 1], (int)(nb * 64 - m));
  return (k >= 0 && d > k) ? -1 : d;
}

#if defined(__HIPCC__)
typedef uint64_t __attribute__((aligned(1))) ed_u64_unaligned;
// bit-planes of the 64 pattern bytes at p (cnt of them valid) from eight 8-byte loads: bit k of byte i of a word lands on
// bit i of the gathered byte ((x >> k) & 0x0101..01) * 0x0102040810204080 >> 56).  Reads up to 7 bytes past p + cnt.
__device__ inline void block_planes_words(const uint8_t* p, int cnt, uint64_t planes[8], uint64_t* valid) {
#pragma unroll
  for (int k = 0; k < 8; k++) planes[k] = 0;
  for (int w = 0; w * 8 < cnt; w++) {
    const uint64_t x = *(const ed_u64_unaligned*)(p + 8 * w);
#pragma unroll
    for (int k = 0; k < 8; k++) planes[k] |= ((((x >> k) & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56) << (8 * w);
  }
  *valid = cnt >= 64 ? ~0ull : ((1ull << cnt) - 1ull);
}

// Banded distance by one whole wave (all 64 lanes call it together; `k` and the strings are wave-uniform): lane = pattern
// block modulo 64 - block b lives in lane b & 63 while it is inside the band and hands the lane to block b + 64 afterwards
// - anti-diagonal schedule: block b works on column t - b at step t and takes hin from block b - 1 (lane - 1, rotating) by a
// shuffle, or +1 where that block has left the band.  Every lane walks consecutive text columns, so it reads the text eight
// bytes at a time (next word requested a word ahead); the pattern bit-planes and (Pv, Mv) stay in registers: no per-column
// carry bytes in HBM.  The text must be readable up to 15 bytes past its end (the pools carry 16 bytes of slack).
// Needs the band to be at most 63 blocks wide (callers fall back to ed_wave_pair otherwise).  Returns the distance if it
// is <= k (k < 0: no cut-off), else -1.
__device__ inline bool ed_wave_band_fits(int64_t la, int64_t lb, int64_t k) {
  int64_t m = la < lb ? la : lb, n = la < lb ? lb : la;
  EdBand bd;
  if (!ed_band(m, n, k, &bd)) return true;          // answered without any work
  return (64 + bd.dl + 2 * bd.kk) / 64 + 2 <= 63 || (m + 63) / 64 <= 63;
}
__device__ inline int64_t ed_wave_pair_k(const uint8_t* A, int64_t la, const uint8_t* B, int64_t lb, int64_t k) {
  const int lane = (int)(threadIdx.x & 63);
  const uint8_t *P = A, *T = B; int64_t m = la, n = lb;
  if (la > lb) { P = B; m = lb; T = A; n = la; }
  EdBand bd;
  if (!ed_band(m, n, k, &bd)) return -1;
  if (m == 0) return n;
  const int64_t nb = (m + 63) / 64;
  int64_t b = lane;                       // the block this lane holds (b, b + 64, ...)
  int64_t cs = 0, ce = -1, nxt_cs = 0;    // its column window; first column of block b + 1 (end of this block's "own" columns)
  uint64_t planes[8], valid = 0, Pv = ~0ull, Mv = 0;
#pragma unroll
  for (int q = 0; q < 8; q++) planes[q] = 0;
  uint64_t tw = 0, tw_next = 0; int64_t jb = 0;    // text bytes [jb, jb + 8) and [jb + 8, jb + 16) of this lane's column walk
  auto enter = [&](int64_t blk) {
    b = blk;
    if (b < nb) {
      const int cnt = (int)(m - b * 64 < 64 ? m - b * 64 : 64);
      block_planes_words(P + b * 64, cnt, planes, &valid);
      cs = ed_band_cs(bd, b); ce = ed_band_ce(bd, b);
      nxt_cs = b + 1 < nb ? ed_band_cs(bd, b + 1) : n;
      Pv = ~0ull; Mv = 0;
      jb = cs; tw = *(const ed_u64_unaligned*)(T + jb); tw_next = *(const ed_u64_unaligned*)(T + jb + 8);
    } else { cs = 0; ce = -1; nxt_cs = 0; }
  };
  enter(lane);
  int64_t acc = 0;       // sum over this lane's finished blocks b < nb - 1 of (64 + sum of hout over [cs_b, cs_{b+1}))
  int64_t rel = 0;       // sum of hout of the current block so far
  int64_t own = 0;       // ... over its own columns only (those before block b + 1 starts)
  int hout = 0;
  uint64_t fPv = ~0ull, fMv = 0; int64_t frel = 0; bool fin = false;
  const int64_t steps = n + nb - 1;      // block b works on column j at step j + b
  for (int64_t t = 0; t < steps; t++) {
    const int up = __shfl(hout, (lane + 63) & 63, 64);     // hout of block b - 1 at the previous step (= the same column)
    const int64_t j = t - b;
    if (b < nb && j >= cs && j <= ce) {
      // block b - 1 covers column j iff j <= ce_{b-1}; its window ends 64 columns before this one does
      const bool above = b > 0 && j <= ed_band_ce(bd, b - 1);
      const int hin = above ? up : 1;
      if (j - jb >= 8) { jb += 8; tw = tw_next; tw_next = *(const ed_u64_unaligned*)(T + jb + 8); }
      hout = advance_block(Pv, Mv, eq_mask(planes, valid, (uint8_t)(tw >> (8 * (j - jb)))), hin);
      rel += hout;
      if (j < nxt_cs) own += hout;
      if (j == ce) {                      // this block is through
        if (b == nb - 1) { fPv = Pv; fMv = Mv; frel = rel; fin = true; }
        else acc += 64 + own;
        rel = 0; own = 0;
        enter(b + 64);
      }
    }
  }
  // D at the bottom of the last (padded) block in the last column = 64 + sum_{b < nb-1} (64 + own_b) + rel_{nb-1}
  int64_t tot = acc + (fin ? frel : 0);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) tot += __shfl_xor(tot, d, 64);
  const int owner = (int)((nb - 1) & 63);
  const uint64_t oPv = __shfl(fPv, owner, 64), oMv = __shfl(fMv, owner, 64);
  const int64_t dist = unpad_score(64 + tot, oPv, oMv, (int)(nb * 64 - m));
  return (k >= 0 && dist > k) ? -1 : dist;
}

// ---- the same banded wavefront for patterns over {A, C, G, T} (what an insertion's ALT is; anything else - 'N', lower case, a
// symbolic "<DEL>" - takes ed_wave_pair_k above).  What bounds a window of kilobase insertions is the LATENCY of one column step of
// one wave (the pairs of a window are evaluated one after the other: an accepted candidate moves its group's means), so the step is
// cut down to what Myers' recurrence needs:
//   * Peq[code] (edlib's per-symbol match vectors, code = (c >> 1) & 3: A 0, C 1, T 2, G 3) built once per block instead of a match
//     mask rebuilt from eight bit-planes per column: three 64-bit selects instead of eight selects + eight ANDs; a text byte that is
//     not one of the four letters matches nothing;
//   * hout -> hin as two bits (plus, minus) ORed into the vectors - no compares - and handed to the next lane by a DPP wave rotate
//     (v_mov_b32_dpp wave_ror:1, a VALU move) instead of a ds_bpermute round trip through the LDS crossbar;
//   * 32-bit column bookkeeping (strings are shorter than 2^30 here; the caller checks).
SNF_D uint32_t ed_acgt_code(uint32_t c) { return (c >> 1) & 3u; }
SNF_D bool ed_is_acgt(uint32_t c) { return ((0x47544341u >> (8 * ed_acgt_code(c))) & 0xffu) == c; }
// all n bytes of p in {A, C, G, T}?  (whole wave; reads up to 7 bytes past p + n)
__device__ inline bool ed_wave_all_acgt(const uint8_t* p, int32_t n) {
  const int lane = (int)(threadIdx.x & 63);
  bool ok = true;
  for (int32_t o = lane * 8; o < n; o += 64 * 8) {
    uint64_t x = *(const ed_u64_unaligned*)(p + o);
    const int cnt = n - o < 8 ? n - o : 8;
    for (int q = 0; q < cnt; q++, x >>= 8) ok = ok && ed_is_acgt((uint32_t)(x & 0xffu));
  }
  return __ballot(!ok) == 0ull;
}
// Peq of the 64 pattern bytes at p (cnt valid): pe[code] bit i = (p[i] == letter(code)).  Reads up to 7 bytes past p + cnt.
__device__ inline void block_peq_words(const uint8_t* p, int cnt, uint64_t pe[4]) {
  pe[0] = pe[1] = pe[2] = pe[3] = 0;
  for (int w = 0; w * 8 < cnt; w++) {
    const uint64_t x = *(const ed_u64_unaligned*)(p + 8 * w);
#pragma unroll
    for (int z = 0; z < 4; z++) {
      const uint64_t y = x ^ (0x0101010101010101ull * (uint64_t)((0x47544341u >> (8 * z)) & 0xffu));   // zero byte <=> that letter
      uint64_t t = (y & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full;
      t = ~(t | y | 0x7f7f7f7f7f7f7f7full);                                                              // 0x80 in every zero byte
      pe[z] |= (((t >> 7) * 0x0102040810204080ull) >> 56) << (8 * w);
    }
  }
  const uint64_t valid = cnt >= 64 ? ~0ull : ((1ull << cnt) - 1ull);
  pe[0] &= valid; pe[1] &= valid; pe[2] &= valid; pe[3] &= valid;
}
// one block, one column; h = horizontal delta as bits: 1 plus, 2 minus
SNF_D uint32_t advance_block2(uint64_t& Pv, uint64_t& Mv, uint64_t Eq, uint32_t hin) {
  const uint64_t hm = (uint64_t)(hin >> 1), hp = (uint64_t)(hin & 1u);
  const uint64_t Xv = Eq | Mv;
  Eq |= hm;
  const uint64_t Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
  uint64_t Ph = Mv | ~(Xh | Pv);
  uint64_t Mh = Pv & Xh;
  const uint32_t hout = (uint32_t)(Ph >> 63) | ((uint32_t)(Mh >> 63) << 1);
  Ph = (Ph << 1) | hp; Mh = (Mh << 1) | hm;
  Pv = Mh | ~(Xv | Ph);
  Mv = Ph & Xv;
  return hout;
}
SNF_D int32_t ed_band_cs32(int32_t kk, int32_t b) { const int32_t c = 64 * b - kk; return c < 0 ? 0 : c; }
SNF_D int32_t ed_band_ce32(int32_t kk, int32_t dl, int32_t n, int32_t b) { const int32_t c = 64 * b + 63 + dl + kk; return c > n - 1 ? n - 1 : c; }
// P (m bytes, all of them A/C/G/T, m <= n < 2^30) against T; same contract as ed_wave_pair_k
__device__ inline int64_t ed_wave_pair_k_acgt(const uint8_t* P, int32_t m, const uint8_t* T, int32_t n, int64_t k) {
  const int lane = (int)(threadIdx.x & 63);
  EdBand bd;
  if (!ed_band((int64_t)m, (int64_t)n, k, &bd)) return -1;
  if (m == 0) return n;
  const int32_t nb = (m + 63) / 64, kk = (int32_t)bd.kk, dl = (int32_t)bd.dl;
  int32_t b = lane;
  int32_t cs = 0, ce = -1, nxt_cs = 0, ce_up = -1;   // ce_up: last column the block above is in the band for (-1: none above)
  uint64_t pe[4] = {0, 0, 0, 0}, Pv = ~0ull, Mv = 0;
  uint64_t tw = 0, tw_next = 0; int32_t jb = 0;
  auto enter = [&](int32_t blk) {
    b = blk;
    if (b < nb) {
      const int cnt = m - b * 64 < 64 ? m - b * 64 : 64;
      block_peq_words(P + b * 64, cnt, pe);
      cs = ed_band_cs32(kk, b); ce = ed_band_ce32(kk, dl, n, b);
      nxt_cs = b + 1 < nb ? ed_band_cs32(kk, b + 1) : n;
      ce_up = b > 0 ? ed_band_ce32(kk, dl, n, b - 1) : -1;
      Pv = ~0ull; Mv = 0;
      jb = cs; tw = *(const ed_u64_unaligned*)(T + jb); tw_next = *(const ed_u64_unaligned*)(T + jb + 8);
    } else { cs = 0; ce = -1; nxt_cs = 0; ce_up = -1; }
  };
  enter(lane);
  int32_t acc = 0, rel = 0, own = 0;     // as in ed_wave_pair_k (sums of +-1 over at most n columns: 32 bits)
  uint32_t hout = 0;
  uint64_t fPv = ~0ull, fMv = 0; int32_t frel = 0; bool fin = false;
  const int32_t steps = n + nb - 1;
  for (int32_t t = 0; t < steps; t++) {
    const uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hout, 0x13C, 0xf, 0xf, false);   // wave_ror:1: lane - 1, lane 0 <- 63
    const int32_t j = t - b;
    if (j >= cs && j <= ce) {               // (a lane without a block holds ce = -1 < cs = 0)
      const uint32_t hin = j <= ce_up ? up : 1u;
      if (j - jb >= 8) { jb += 8; tw = tw_next; tw_next = *(const ed_u64_unaligned*)(T + jb + 8); }
      const uint32_t c = (uint32_t)(tw >> (8 * (j - jb))) & 0xffu;
      const uint32_t cd = ed_acgt_code(c);
      const uint64_t e01 = (cd & 1u) ? pe[1] : pe[0], e23 = (cd & 1u) ? pe[3] : pe[2];
      uint64_t eq = (cd & 2u) ? e23 : e01;
      if (!ed_is_acgt(c)) eq = 0;
      hout = advance_block2(Pv, Mv, eq, hin);
      const int32_t dh = (int32_t)(hout & 1u) - (int32_t)(hout >> 1);
      rel += dh;
      if (j < nxt_cs) own += dh;
      if (j == ce) {
        if (b == nb - 1) { fPv = Pv; fMv = Mv; frel = rel; fin = true; }
        else acc += 64 + own;
        rel = 0; own = 0;
        enter(b + 64);
      }
    }
  }
  int64_t tot = (int64_t)acc + (fin ? frel : 0);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) tot += __shfl_xor(tot, d, 64);
  const int owner = (int)((nb - 1) & 63);
  const uint64_t oPv = __shfl(fPv, owner, 64), oMv = __shfl(fMv, owner, 64);
  const int64_t dist = unpad_score(64 + tot, oPv, oMv, (int)(nb * 64 - m));
  return (k >= 0 && dist > k) ? -1 : dist;
}
// ---- eight columns per step.  The wavefront above hands ONE column's carry to the next lane per step; here a lane works through a
// GROUP of eight consecutive columns (one 8-byte text word) of its block and hands the eight carries on as one 16-bit word: one DPP
// move, one window test, one text load and one score update per eight columns instead of per column, and the eight column steps of a
// group are straight-line code.  For that the band is widened to whole groups - a block enters at its first column rounded DOWN to a
// multiple of eight and is followed to the end of the group its last column lies in (clipped at the text's end: only that last group
// can be partial) - which keeps the staircase shape the banded recurrence needs (every value computed still is the cost of a real path,
// and the band still holds every path of cost <= k).  The skew between neighbouring blocks is one group, so a pair takes
// ceil(n / 8) + nb - 1 group steps: the fill of the pipeline costs eight times as many columns as above - ~750 of 6 200 for a 6-kb
// insertion - against a column step of ~45 instructions instead of ~75.  T_ACGT: the text too is over {A, C, G, T} (no per-column test).
template <bool T_ACGT>
__device__ inline int64_t ed_wave_pair_k_acgt8(const uint8_t* P, int32_t m, const uint8_t* T, int32_t n, int64_t k) {
  const int lane = (int)(threadIdx.x & 63);
  EdBand bd;
  if (!ed_band((int64_t)m, (int64_t)n, k, &bd)) return -1;
  if (m == 0) return n;
  const int32_t nb = (m + 63) / 64, kk = (int32_t)bd.kk, dl = (int32_t)bd.dl;
  int32_t b = lane;
  int32_t g_lo = 0, g_hi = -1, g_own_end = 0, g_up_hi = -1, n_last = 8;   // groups of the block's window; first group of block b + 1; last
  uint64_t pe[4] = {0, 0, 0, 0}, Pv = ~0ull, Mv = 0;                        // group the block above covers; columns of the last group
  uint64_t tw = 0, tw_next = 0;
  auto enter = [&](int32_t blk) {
    b = blk;
    if (b < nb) {
      const int cnt = m - b * 64 < 64 ? m - b * 64 : 64;
      block_peq_words(P + b * 64, cnt, pe);
      int32_t ce = ed_band_ce32(kk, dl, n, b) | 7;          // to the end of its group ...
      if (ce > n - 1) ce = n - 1;                            // ... or of the text: the only group that can be partial
      g_lo = ed_band_cs32(kk, b) >> 3; g_hi = ce >> 3; n_last = (ce & 7) + 1;
      g_own_end = b + 1 < nb ? (ed_band_cs32(kk, b + 1) >> 3) : 0x7fffffff;
      g_up_hi = b > 0 ? (ed_band_ce32(kk, dl, n, b - 1) >> 3) : -1;
      Pv = ~0ull; Mv = 0;
      tw = *(const ed_u64_unaligned*)(T + 8 * g_lo); tw_next = *(const ed_u64_unaligned*)(T + 8 * g_lo + 8);
    } else { g_lo = 0; g_hi = -1; g_own_end = 0; g_up_hi = -1; n_last = 8; }
  };
  enter(lane);
  int32_t acc = 0, rel = 0, own = 0;
  uint32_t hout = 0;                       // the group's carries: bit q = +1 after column q, bit 8 + q = -1
  uint64_t fPv = ~0ull, fMv = 0; int32_t frel = 0; bool fin = false;
  const int32_t steps = ((n + 7) >> 3) + nb - 1;
  auto column = [&](int q, uint32_t hin_p, uint32_t hin_m, uint32_t& out_p, uint32_t& out_m) {
    const uint32_t c = (uint32_t)(tw >> (8 * q)) & 0xffu;
    const uint64_t e01 = (c & 2u) ? pe[1] : pe[0], e23 = (c & 2u) ? pe[3] : pe[2];
    uint64_t eq = (c & 4u) ? e23 : e01;
    if (!T_ACGT) { if (!ed_is_acgt(c)) eq = 0; }
    const uint32_t h = advance_block2(Pv, Mv, eq, ((hin_p >> q) & 1u) | (((hin_m >> q) & 1u) << 1));
    out_p |= (h & 1u) << q; out_m |= (h >> 1) << q;
  };
  for (int32_t t = 0; t < steps; t++) {
    const uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hout, 0x13C, 0xf, 0xf, false);   // wave_ror:1: lane - 1, lane 0 <- 63
    const int32_t G = t - b;
    if (G >= g_lo && G <= g_hi) {          // (a lane without a block holds g_hi = -1 < g_lo = 0)
      const bool above = G <= g_up_hi;     // whole groups: the block above is in the band for all of this group's columns or for none
      const uint32_t hin_p = above ? (up & 0xffu) : 0xffu, hin_m = above ? (up >> 8) : 0u;
      uint32_t out_p = 0, out_m = 0;
      if (G < g_hi || n_last == 8) {
#pragma unroll
        for (int q = 0; q < 8; q++) column(q, hin_p, hin_m, out_p, out_m);
      } else {
        for (int q = 0; q < n_last; q++) column(q, hin_p, hin_m, out_p, out_m);
      }
      hout = out_p | (out_m << 8);
      const int32_t dsum = __builtin_popcount(out_p) - __builtin_popcount(out_m);
      rel += dsum;
      if (G < g_own_end) own += dsum;
      if (G == g_hi) {
        if (b == nb - 1) { fPv = Pv; fMv = Mv; frel = rel; fin = true; }
        else acc += 64 + own;
        rel = 0; own = 0;
        enter(b + 64);
      } else {
        tw = tw_next; tw_next = *(const ed_u64_unaligned*)(T + 8 * (G + 2));
      }
    }
  }
  int64_t tot = (int64_t)acc + (fin ? frel : 0);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) tot += __shfl_xor(tot, d, 64);
  const int owner = (int)((nb - 1) & 63);
  const uint64_t oPv = __shfl(fPv, owner, 64), oMv = __shfl(fMv, owner, 64);
  const int64_t dist = unpad_score(64 + tot, oPv, oMv, (int)(nb * 64 - m));
  return (k >= 0 && dist > k) ? -1 : dist;
}

// the banded wave form for any pair the band of which fits (ed_wave_band_fits): equal strings answer at once, patterns over
// {A, C, G, T} take the short column step, everything else the bit-plane form
__device__ inline int64_t ed_wave_pair_k_any(const uint8_t* A, int64_t la, const uint8_t* B, int64_t lb, int64_t k) {
  const uint8_t *P = A, *T = B; int64_t m = la, n = lb;
  if (la > lb) { P = B; m = lb; T = A; n = la; }
  if (n >= ((int64_t)1 << 30)) return ed_wave_pair_k(A, la, B, lb, k);
  if (m == n) {   // identical strings (every symbolic ALT of a type, a shared allele): distance 0 without a column step
    const int lane = (int)(threadIdx.x & 63);
    bool same = true;
    for (int64_t o = (int64_t)lane * 8; o < m && same; o += 64 * 8) {
      uint64_t x = *(const ed_u64_unaligned*)(P + o), y = *(const ed_u64_unaligned*)(T + o);
      if (m - o < 8) { const uint64_t keep = (1ull << (8 * (m - o))) - 1ull; x &= keep; y &= keep; }
      same = x == y;
    }
    if (__ballot(!same) == 0ull) return 0;
  }
  if (ed_wave_all_acgt(P, (int32_t)m)) {
#ifdef SNF_MYERS_ONE_COLUMN
    return ed_wave_pair_k_acgt(P, (int32_t)m, T, (int32_t)n, k);
#else
    // (the text may be read up to 23 bytes past its end here: the words of the group behind the last one; the pools carry that slack)
    return ed_wave_all_acgt(T, (int32_t)n) ? ed_wave_pair_k_acgt8<true>(P, (int32_t)m, T, (int32_t)n, k)
                                           : ed_wave_pair_k_acgt8<false>(P, (int32_t)m, T, (int32_t)n, k);
#endif
  }
  return ed_wave_pair_k(A, la, B, lb, k);
}

// the same distance computed by one whole wave (all 64 lanes must call it together): lane = 64-row block of the current
// 64-block pass, anti-diagonal schedule (lane l works on column t - l at step t and takes hin from lane l-1 by a
// shuffle); patterns of more than 64 blocks take several passes linked through `carry` (>= max(la, lb) bytes)
__device__ inline int64_t ed_wave_pair(const uint8_t* A, int64_t la, const uint8_t* B, int64_t lb, int8_t* carry) {
  const int lane = (int)(threadIdx.x & 63);
  const uint8_t *P = A, *T = B; int64_t m = la, n = lb;
  if (la > lb) { P = B; m = lb; T = A; n = la; }
  if (m == 0) return n;
  const int64_t nb = (m + 63) / 64;
  int64_t score = nb * 64;
  uint64_t fPv = ~0ull, fMv = 0;
  for (int64_t b0 = 0; b0 < nb; b0 += 64) {
    const int nl = (int)(nb - b0 < 64 ? nb - b0 : 64);   // lanes (blocks) active in this pass
    const int64_t blk = b0 + lane;
    uint64_t planes[8], valid = 0, Pv = ~0ull, Mv = 0;
    for (int k = 0; k < 8; k++) planes[k] = 0;
    if (lane < nl) {
      const int cnt = (int)(m - blk * 64 < 64 ? m - blk * 64 : 64);
      block_planes(P + blk * 64, cnt, planes, &valid);
    }
    const bool glast = lane == nl - 1 && b0 + nl == nb;    // owns the last block of the pattern
    int hout = 0;
    const int64_t steps = n + nl - 1;
    for (int64_t t = 0; t < steps; t++) {
      const int up = __shfl_up(hout, 1, 64);               // hout of the block above, previous step
      const int64_t j = t - lane;
      if (lane < nl && j >= 0 && j < n) {
        const int hin = lane == 0 ? (b0 == 0 ? 1 : (int)carry[j]) : up;
        hout = advance_block(Pv, Mv, eq_mask(planes, valid, T[j]), hin);
        if (glast) score += hout;
        else if (lane == nl - 1) carry[j] = (int8_t)hout;  // feeds block b0+64 in the next pass
      }
    }
    if (glast) { fPv = Pv; fMv = Mv; }
    __syncthreads();   // one-wave workgroups: orders the carry bytes between passes
  }
  const int owner = (int)((nb - 1) & 63);
  const int64_t sc = __shfl(score, owner, 64);
  const uint64_t oPv = __shfl(fPv, owner, 64), oMv = __shfl(fMv, owner, 64);
  return unpad_score(sc, oPv, oMv, (int)(nb * 64 - m));
}
#endif

}  // namespace snf
