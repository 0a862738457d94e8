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
