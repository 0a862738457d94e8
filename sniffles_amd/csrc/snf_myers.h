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

#if !defined(SNF_EMU) && defined(__HIPCC__)
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
