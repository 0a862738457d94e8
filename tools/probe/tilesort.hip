// micro-benchmark: block-local radix sort of 32-bit keys (rocprim::block_radix_sort), tiles of T keys (dev tool)
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/block/block_radix_sort.hpp>
#include <cstdio>
#include <vector>
#include <random>
template <int IPT>
__global__ void __launch_bounds__(256) tilesort(const uint32_t* in, uint32_t* out, int64_t n, int end_bit) {
  using BRS = rocprim::block_radix_sort<uint32_t, 256, IPT>;
  __shared__ typename BRS::storage_type st;
  const int64_t base = (int64_t)blockIdx.x * 256 * IPT;
  uint32_t k[IPT];
  for (int j = 0; j < IPT; j++) { int64_t p = base + (int64_t)threadIdx.x * IPT + j; k[j] = p < n ? in[p] : 0xffffffffu >> (32 - end_bit); }
  BRS().sort_to_striped(k, st, 0, end_bit);
  for (int j = 0; j < IPT; j++) { int64_t p = base + j * 256 + threadIdx.x; if (p < n) out[p] = k[j]; }
}
template <class F> float timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize(); float best = 1e9;
  for (int r = 0; r < 5; r++) { hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
  return best * 1e3f;
}
int main() {
  const int64_t n = 4632000;
  std::vector<uint32_t> h(n); std::mt19937 g(1); for (auto& x : h) x = g() & 0x3ffffff;
  uint32_t *in, *out; hipMalloc(&in, n * 4); hipMalloc(&out, n * 4); hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
  for (int bits : {20, 26, 30}) {
    printf("bits %d: T=4096 %7.1f us | T=8192 %7.1f us\n", bits,
           timeit([&] { hipLaunchKernelGGL(tilesort<16>, dim3((n + 4095) / 4096), dim3(256), 0, 0, in, out, n, bits); }),
           timeit([&] { hipLaunchKernelGGL(tilesort<32>, dim3((n + 8191) / 8192), dim3(256), 0, 0, in, out, n, bits); }));
  }
  return 0;
}
