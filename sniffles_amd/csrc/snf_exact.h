// snf_exact.h - bit-exact numeric building blocks shared by the kernel bodies.
//
// The reference computes its statistics with CPython 3.10 `statistics` (exact rational variance,
// ONE rounding to double, then math.sqrt; util.py:25-27) and plain IEEE double arithmetic.  These
// helpers reproduce that on gfx950 without fused multiply-add (compile with -ffp-contract=off).
#pragma once
#include "snf_rt.h"

#include <cmath>

namespace snf {

SNF_HD int bitlen128(u128 x) {
  uint64_t hi = (uint64_t)(x >> 64), lo = (uint64_t)x;
  if (hi) return 128 - __builtin_clzll(hi);
  if (lo) return 64 - __builtin_clzll(lo);
  return 0;
}

// quotient and remainder of a 128-bit numerator by a 64-bit denominator, the quotient below 2^63 (ratio_to_double asks for 55-56 bits).
// A double-precision quotient is off by a few units at most; the remainder it leaves is formed exactly in 128-bit integers and a
// second estimate plus at most a few unit steps settle it - the result is exact whatever the floating-point division rounds to
// (host tier and GPU agree by construction).  The bitwise schoolbook loop this replaces ran ~3 000 dependent instructions per
// division; a lone wave doing four of them in a row was the duration of c1_mergeruns.
SNF_HD void udivmod128_64(u128 num, uint64_t den, u128* q, uint64_t* r) {
  const double dd = (double)den;
  const double dn = (double)(uint64_t)(num >> 64) * 18446744073709551616.0 + (double)(uint64_t)num;
  double e = dn / dd;
  if (e < 0.0) e = 0.0;
  if (e > 9223372036854775807.0) e = 9223372036854775807.0;
  uint64_t quo = (uint64_t)e;
  i128 rem = (i128)num - (i128)((u128)quo * den);       // |rem| is a small multiple of den: exact in 128 bits
  for (int it = 0; it < 2; it++) {                      // the estimate of what is left (each pass gains ~50 bits)
    const bool neg = rem < 0;
    const u128 mag = neg ? (u128)(-rem) : (u128)rem;
    if (!neg && mag < den) break;
    const double dm = (double)(uint64_t)(mag >> 64) * 18446744073709551616.0 + (double)(uint64_t)mag;
    double c = dm / dd;
    if (c > 9.0e18) c = 9.0e18;
    const uint64_t dq = (uint64_t)c;
    if (neg) { quo -= dq; rem += (i128)((u128)dq * den); } else { quo += dq; rem -= (i128)((u128)dq * den); }
  }
  while (rem < 0) { quo--; rem += den; }
  while (rem >= (i128)den) { quo++; rem -= den; }
  *q = quo;
  *r = (uint64_t)rem;
}

// correctly rounded (nearest-even) double of the exact rational num/den: float(Fraction(num, den))
SNF_HD double ratio_to_double(u128 num, uint64_t den) {
  if (num == 0) return 0.0;
  if ((num >> 53) == 0 && (den >> 53) == 0) return (double)(uint64_t)num / (double)den;  // both exact in fp64
  int bn = bitlen128(num), bd = 64 - __builtin_clzll(den);
  int s = 55 - (bn - bd);  // quotient gets 55 or 56 bits
  bool sticky = false;
  u128 N;
  if (s >= 0) {
    N = num << s;  // bn + s = 55 + bd <= 119 bits
  } else {
    N = num >> (-s);
    sticky = (num & ((((u128)1) << (-s)) - 1)) != 0;
  }
  u128 q;
  uint64_t r;
  udivmod128_64(N, den, &q, &r);
  uint64_t q64 = (uint64_t)q;
  if (r || sticky) q64 |= 1;
  return ldexp((double)q64, -s);
}

// statistics.stdev of n integers given S1 = sum(d), S2 = sum(d*d), d = x - x0 (any offset x0)
SNF_HD double stdev_from_sums(int64_t n, i128 S1, u128 S2) {
  if (n < 2) return 0.0;
  u128 num = (u128)n * S2 - (u128)(S1 * S1);
  uint64_t den = (uint64_t)n * (uint64_t)(n - 1);
  return sqrt(ratio_to_double(num, den));
}

// stdev of a[0..n) (already any order)
SNF_HD double stdev_i32(const int32_t* a, int64_t n) {
  if (n < 2) return 0.0;
  i128 S1 = 0;
  u128 S2 = 0;
  int64_t x0 = a[0];
  for (int64_t i = 0; i < n; i++) {
    int64_t d = (int64_t)a[i] - x0;
    S1 += d;
    S2 += (u128)((i128)d * d);
  }
  return stdev_from_sums(n, S1, S2);
}

// ---- small sorts on scratch memory ------------------------------------------------------------
// sort a[0..n) with a strict-weak `less`; callers make keys total (append the index) so the result
// is independent of the algorithm.  Insertion for short inputs, heapsort otherwise: O(n log n), in place.
template <class T, class Less>
SNF_HD void sort_inplace(T* a, int64_t n, Less less) {
  if (n < 2) return;
  if (n <= 20) {
    for (int64_t i = 1; i < n; i++) {
      T x = a[i];
      int64_t j = i - 1;
      while (j >= 0 && less(x, a[j])) {
        a[j + 1] = a[j];
        j--;
      }
      a[j + 1] = x;
    }
    return;
  }
  for (int64_t start = n / 2 - 1; start >= 0; start--) {  // heapify
    int64_t root = start;
    T x = a[root];
    for (;;) {
      int64_t c = 2 * root + 1;
      if (c >= n) break;
      if (c + 1 < n && less(a[c], a[c + 1])) c++;
      if (!less(x, a[c])) break;
      a[root] = a[c];
      root = c;
    }
    a[root] = x;
  }
  for (int64_t end = n - 1; end > 0; end--) {
    T x = a[end];
    a[end] = a[0];
    int64_t root = 0;
    for (;;) {
      int64_t c = 2 * root + 1;
      if (c >= end) break;
      if (c + 1 < end && less(a[c], a[c + 1])) c++;
      if (!less(x, a[c])) break;
      a[root] = a[c];
      root = c;
    }
    a[root] = x;
  }
}

struct LessI32 {
  SNF_HD bool operator()(int32_t a, int32_t b) const { return a < b; }
};
struct LessU32 {
  SNF_HD bool operator()(uint32_t a, uint32_t b) const { return a < b; }
};

// util.median_modes (= util.center, util.py:49-58,167) on a SORTED array: values whose count is within 2
// of the max count, upper median of those distinct values
SNF_HD int32_t center_sorted(const int32_t* s, int64_t n) {
  int64_t maxc = 0;
  for (int64_t i = 0; i < n;) {
    int64_t j = i;
    while (j < n && s[j] == s[i]) j++;
    if (j - i > maxc) maxc = j - i;
    i = j;
  }
  int64_t k = 0;
  for (int64_t i = 0; i < n;) {
    int64_t j = i;
    while (j < n && s[j] == s[i]) j++;
    if (maxc - (j - i) < 3) k++;
    i = j;
  }
  int64_t want = k / 2, seen = 0;
  for (int64_t i = 0; i < n;) {
    int64_t j = i;
    while (j < n && s[j] == s[i]) j++;
    if (maxc - (j - i) < 3) {
      if (seen == want) return s[i];
      seen++;
    }
    i = j;
  }
  return s[0];
}

#if defined(__HIPCC__)
// sort_inplace by a whole wave that executes the surrounding code UNIFORMLY (every lane runs the same statements on the same
// data: the x_big kernels of snf_wave_call.h): lane p ranks the elements p, p + 64, ... against all others (stable: equal
// elements keep their order) and scatters them through `tmp` (n elements).  O(n^2 / 64) comparisons instead of
// O(n log n) dependent ones on one lane - the sorts are what a serial cluster body spends its time in.
template <class T, class Less>
__device__ inline void wave_sort_inplace(T* a, int64_t n, Less less, T* tmp) {
  if (n < 2) return;
  const int lane = (int)(threadIdx.x & 63);
  __syncthreads();                                   // (one-wave workgroups) the stores that filled `a` are visible
  for (int64_t p = lane; p < n; p += 64) {
    const T x = a[p];
    int64_t r = 0;
    for (int64_t q = 0; q < n; q++) { const T y = a[q]; r += (less(y, x) || (!less(x, y) && q < p)) ? 1 : 0; }
    tmp[r] = x;
  }
  __syncthreads();
  for (int64_t p = lane; p < n; p += 64) a[p] = tmp[p];
  __syncthreads();
}
// `tmp`: scratch of n elements; `uniform`: the caller runs wave-uniformly (View::wave_uniform)
#define SNF_SORT(uniform, a, n, less, tmp) do { if (uniform) wave_sort_inplace((a), (n), (less), (tmp)); else sort_inplace((a), (n), (less)); } while (0)
#else
#define SNF_SORT(uniform, a, n, less, tmp) sort_inplace((a), (n), (less))
#endif

// util.stdev(util.trim(nums)) on a SORTED array (util.py:25-27,82-88)
SNF_HD double stdev_trim_sorted(const int32_t* s, int64_t n) {
  int64_t trim_n = (int64_t)((double)n / 100.0 * 25.0);
  if (trim_n > 0) return stdev_i32(s + trim_n, n - 2 * trim_n);
  return stdev_i32(s, n);
}

// first index in [lo,hi) with a[idx] >= x / > x
SNF_HD int64_t lower_bound_i32(const int32_t* a, int64_t lo, int64_t hi, int64_t x) {
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if ((int64_t)a[mid] < x) lo = mid + 1; else hi = mid;
  }
  return lo;
}
SNF_HD int64_t upper_bound_i32(const int32_t* a, int64_t lo, int64_t hi, int64_t x) {
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if ((int64_t)a[mid] <= x) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// the same bounds with a sampled top level: top[k] == a[k << SNF_TOP_SHIFT] for every k << SNF_TOP_SHIFT inside the array.
// The top level of a multi-million entry array stays cache resident, so a query touches 2-3 cold lines instead of ~20.
#define SNF_TOP_SHIFT 8
template <bool UPPER>
SNF_HD int64_t bound_top_i32(const int32_t* a, const int32_t* top, int64_t lo, int64_t hi, int64_t x) {
  int64_t kl = (lo + (1 << SNF_TOP_SHIFT) - 1) >> SNF_TOP_SHIFT, kh = (hi + (1 << SNF_TOP_SHIFT) - 1) >> SNF_TOP_SHIFT;
  const int64_t k0 = kl;
  while (kl < kh) {  // first sampled k in [kl, kh) whose value is past x
    int64_t mid = (kl + kh) >> 1;
    const int64_t val = top[mid];
    if (UPPER ? (val <= x) : (val < x)) kl = mid + 1; else kh = mid;
  }
  const int64_t kend = (hi + (1 << SNF_TOP_SHIFT) - 1) >> SNF_TOP_SHIFT;
  const int64_t nlo = kl > k0 ? ((kl - 1) << SNF_TOP_SHIFT) : lo;
  const int64_t nhi = kl < kend ? (kl << SNF_TOP_SHIFT) : hi;
  return UPPER ? upper_bound_i32(a, nlo, nhi, x) : lower_bound_i32(a, nlo, nhi, x);
}

// ---- the same upper bound as a 16-ary descent: three sampled levels over the GLOBAL positions of the array - top[k] == a[k << 8]
// (above), mid[j] == a[j << 4] - so that a query is a short binary search over every 16th entry of `top` (a few KB per task: cache
// resident) followed by THREE aligned 64-byte node reads (16 samples of level 8, of level 4, 16 entries of the array itself) instead
// of ~19 dependent 4-byte probes.  What bounds the coverage queries of a pass is the length of that dependent chain (a thread per
// query has nothing else to do), not bytes.  The arrays live in 256-byte granules, so a node read never leaves its allocation; the
// entries of a node outside [lo, hi) are masked by position.
SNF_HD void load16_i32(const int32_t* p, int32_t (&o)[16]) {
  typedef uint4 __attribute__((aligned(4))) uint4_dw;     // (a column of the input arena is only known to be 4-byte aligned here)
  const uint4_dw* q = (const uint4_dw*)p;
  const uint4 a = q[0], b = q[1], c = q[2], d = q[3];
  o[0] = (int32_t)a.x; o[1] = (int32_t)a.y; o[2] = (int32_t)a.z; o[3] = (int32_t)a.w;
  o[4] = (int32_t)b.x; o[5] = (int32_t)b.y; o[6] = (int32_t)b.z; o[7] = (int32_t)b.w;
  o[8] = (int32_t)c.x; o[9] = (int32_t)c.y; o[10] = (int32_t)c.z; o[11] = (int32_t)c.w;
  o[12] = (int32_t)d.x; o[13] = (int32_t)d.y; o[14] = (int32_t)d.z; o[15] = (int32_t)d.w;
}
// one level: lvl[p >> S] == a[p] for the positions p that are multiples of 2^S.  In: the answer lies in [bl, bh] and [bl, bh) sits
// inside one aligned block of 2^(S+4) positions.  Out: the same with 2^S.
template <int S>
SNF_HD void rank_level16(const int32_t* lvl, int64_t x, int64_t& bl, int64_t& bh) {
  if (bl >= bh) return;
  const int64_t B = (bl >> (S + 4)) << (S + 4);
  int32_t val[16];
  load16_i32(lvl + (B >> S), val);
  const int j0 = (int)((bl - B + ((int64_t)1 << S) - 1) >> S);      // first sampled position >= bl ...
  const int j1 = (int)((bh - B + ((int64_t)1 << S) - 1) >> S);      // ... and the first >= bh (at most 16)
  int c = 0;
#pragma unroll
  for (int j = 0; j < 16; j++) c += (j >= j0 && j < j1 && (int64_t)val[j] <= x) ? 1 : 0;
  const int jf = j0 + c;                                             // the first sample past x (the samples ascend)
  if (c > 0) bl = B + ((int64_t)(jf - 1) << S);
  if (jf < j1) bh = B + ((int64_t)jf << S);
}
SNF_HD int64_t rank_upper_16ary(const int32_t* a, const int32_t* mid, const int32_t* top, int64_t lo, int64_t hi, int64_t x) {
  if (lo >= hi) return lo;
  int64_t kl = (lo + 4095) >> 12, kh = (hi + 4095) >> 12;          // level 12 = every 16th entry of `top`
  const int64_t k0 = kl, kend = kh;
  while (kl < kh) {
    const int64_t m = (kl + kh) >> 1;
    if ((int64_t)top[m << 4] <= x) kl = m + 1; else kh = m;
  }
  int64_t bl = kl > k0 ? ((kl - 1) << 12) : lo;
  int64_t bh = kl < kend ? (kl << 12) : hi;
  rank_level16<8>(top, x, bl, bh);
  rank_level16<4>(mid, x, bl, bh);
  rank_level16<0>(a, x, bl, bh);
  return bh;
}

// upper bound of x in a[lo, hi) starting from a position `hint` in [lo, hi] where a nearby query ended: gallops away from
// the hint (1, 2, 4, ... entries) and bisects the bracket.  Same result as upper_bound_i32; a query a few hundred bp from
// the previous one stays inside the cache lines that one brought in.
SNF_HD int64_t upper_bound_hint_i32(const int32_t* a, int64_t lo, int64_t hi, int64_t x, int64_t hint) {
  int64_t l, h;
  if (hint < hi && (int64_t)a[hint] <= x) {        // answer is right of the hint
    int64_t step = 1; l = hint + 1; h = l;
    while (h < hi && (int64_t)a[h] <= x) { l = h + 1; h += step; step <<= 1; }
    if (h > hi) h = hi;
  } else {                                          // a[hint] > x (or hint == hi): answer is at or left of the hint
    int64_t step = 1; h = hint; l = h;
    while (l > lo && (int64_t)a[l - 1] > x) { h = l - 1; l -= step; step <<= 1; }
    if (l < lo) l = lo;
  }
  return upper_bound_i32(a, l, h, x);
}

// numpy's pairwise float64 summation (np.sum / np.nanmean, used by parallel.py:214), gather form:
// element i is get(i).  Iterative restatement of the recursion; DEPTH = frames needed: n <= 128 * 2^(DEPTH-1)
// (one frame for n <= 128: the wave kernels only see clusters of <= 64 leads and must not carry a 1.5 KB stack)
template <int DEPTH, class Get>
SNF_HD double np_pairwise_sum(Get get, int64_t n) {
  struct Fr { int64_t lo, n; int state; double left; };
  Fr st[DEPTH + 1];
  int sp = 0;
  st[0] = Fr{0, n, 0, 0.0};
  double ret = 0.0;
  while (sp >= 0) {
    Fr& f = st[sp];
    if (f.state == 0) {
      if (f.n < 8) {
        double res = 0.0;
        for (int64_t i = 0; i < f.n; i++) res += get(f.lo + i);
        ret = res; sp--;
      } else if (f.n <= 128) {
        double r[8];
        for (int j = 0; j < 8; j++) r[j] = get(f.lo + j);
        int64_t i;
        for (i = 8; i < f.n - (f.n % 8); i += 8)
          for (int j = 0; j < 8; j++) r[j] += get(f.lo + i + j);
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < f.n; i++) res += get(f.lo + i);
        ret = res; sp--;
      } else {
        int64_t n2 = f.n / 2;
        n2 -= n2 % 8;
        f.state = 1;
        st[sp + 1] = Fr{f.lo, n2, 0, 0.0};
        sp++;
      }
    } else if (f.state == 1) {
      f.left = ret;
      int64_t n2 = f.n / 2;
      n2 -= n2 % 8;
      f.state = 2;
      st[sp + 1] = Fr{f.lo + n2, f.n - n2, 0, 0.0};
      sp++;
    } else {
      ret = f.left + ret; sp--;
    }
  }
  return ret;
}

SNF_HD int64_t iabs64(int64_t x) { return x < 0 ? -x : x; }

}  // namespace snf
