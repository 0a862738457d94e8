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

}  // namespace snf
