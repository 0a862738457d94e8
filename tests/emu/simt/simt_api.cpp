// TEST-ONLY: counters of the fibre shim (hip/hip_runtime.h), one definition per shim-built library.
#include <hip/hip_runtime.h>

// out[0..8]: fibre switches, cross-lane operations (per wave), workgroup barriers, cross-lane operations released with a
// subset of the wave (divergent control flow), kernel launches, launches abandoned because lock step could not be set up;
// lock-step mode: page faults taken, intervals merged, bytes two lanes left with different values
extern "C" void snf_simt_counters(unsigned long long* out) {
  out[0] = simt::g_n_switch; out[1] = simt::g_n_wave_ops; out[2] = simt::g_n_block_syncs; out[3] = simt::g_n_divergent_ops;
  out[4] = simt::g_n_launches; out[5] = simt::g_n_unmodelled;
  out[6] = simt::g_n_lockstep_faults; out[7] = simt::g_n_lockstep_merges; out[8] = simt::g_n_lockstep_conflicts;
}
extern "C" unsigned long long snf_simt_unmodelled() { return simt::g_n_unmodelled; }

// Known answers for snf_exact.h::udivmod128_64 (the product's exact 128 / 64-bit division) against the host compiler's own
// __int128 division: numerators of the widths ratio_to_double produces (55 + bits(den), quotient of 55-56 bits), narrower ones,
// exact multiples of the denominator and their neighbours.  Returns the number of mismatches over `n` random cases.
#include "../../../sniffles_amd/csrc/snf_exact.h"
extern "C" long snf_simt_divcheck(long n, unsigned long long seed) {
  unsigned long long x = seed * 0x9E3779B97F4A7C15ull + 1;
  auto next = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
  long bad = 0;
  for (long it = 0; it < n; it++) {
    uint64_t den = next() >> (next() % 64); if (!den) den = 1;
    const int bd = 64 - __builtin_clzll(den);
    u128 num = ((u128)next() << 64) | next();
    int want = (it % 3 == 0) ? (int)(next() % (55 + bd)) + 1 : 55 + bd;
    if (want > 119) want = 119;
    num >>= (128 - want);
    if (it % 100 == 1) num = (u128)den * (next() >> 9);
    if (it % 100 == 2) num = (u128)den * (next() >> 9) + den - 1;
    if (it % 100 == 3) num = (u128)den * (next() >> 9) - 1;
    if (it % 100 == 4) num = 0;
    if ((num / den) >> 63) continue;
    u128 q; uint64_t r;
    snf::udivmod128_64(num, den, &q, &r);
    if (q != num / den || r != (uint64_t)(num % den)) bad++;
  }
  return bad;
}
