// TEST-ONLY stand-in for the two rocPRIM device primitives the library uses (see ../hip/hip_runtime.h).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <numeric>
#include <vector>

namespace rocprim {
template <class T> struct plus { T operator()(const T& a, const T& b) const { return a + b; } };

struct default_config {};
template <class A = default_config, class B = default_config, class C = default_config, size_t MergeSortLimit = 1024 * 1024> struct radix_sort_config {};

// stable, by the key bits [begin_bit, end_bit)   (Config: the algorithm selection of the real library; nothing to select here)
template <class Config = default_config, class K, class V>
inline hipError_t radix_sort_pairs(void* tmp, size_t& need, const K* kin, K* kout, const V* vin, V* vout, size_t n,
                                   unsigned begin_bit, unsigned end_bit, hipStream_t = nullptr, bool = false) {
  if (!tmp) { need = 16; return hipSuccess; }
  const unsigned nbits = end_bit - begin_bit;
  const K mask = nbits >= sizeof(K) * 8 ? (K)~(K)0 : (K)((((K)1) << nbits) - 1);
  std::vector<size_t> idx(n);
  std::iota(idx.begin(), idx.end(), (size_t)0);
  std::stable_sort(idx.begin(), idx.end(), [&](size_t x, size_t y) { return ((kin[x] >> begin_bit) & mask) < ((kin[y] >> begin_bit) & mask); });
  std::vector<K> ks(n); std::vector<V> vs(n);
  for (size_t i = 0; i < n; i++) { ks[i] = kin[idx[i]]; vs[i] = vin[idx[i]]; }
  for (size_t i = 0; i < n; i++) { kout[i] = ks[i]; vout[i] = vs[i]; }
  return hipSuccess;
}
template <class In, class Out, class T, class Op>
inline hipError_t exclusive_scan(void* tmp, size_t& need, In in, Out out, T init, size_t n, Op op, hipStream_t = nullptr, bool = false) {
  if (!tmp) { need = 16; return hipSuccess; }
  T acc = init;
  for (size_t i = 0; i < n; i++) { const T x = (T)in[i]; out[i] = acc; acc = op(acc, x); }
  return hipSuccess;
}
}  // namespace rocprim
