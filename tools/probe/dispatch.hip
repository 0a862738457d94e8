// micro-benchmark: cost of dispatching many small workgroups on gfx950 (dev tool, not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct Big { void* p[180]; int x[64]; };  // ~1.7 KB of kernel arguments like the library's View
template <int LDS>
__global__ void __launch_bounds__(256) k_empty(Big b, int* out) {
  __shared__ int s[LDS / 4];
  if (threadIdx.x == 0) s[0] = blockIdx.x;
  __syncthreads();
  if (b.x[0] == 12345) out[blockIdx.x] = s[0];
}
template <int LDS>
__global__ void __launch_bounds__(256) k_small(int* out, int flag) {
  __shared__ int s[LDS / 4];
  if (threadIdx.x == 0) s[0] = blockIdx.x;
  __syncthreads();
  if (flag == 12345) out[blockIdx.x] = s[0];
}
__global__ void __launch_bounds__(64) k_wave(int* out, int flag) { if (flag == 12345) out[blockIdx.x] = 1; }
template <class F> float timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  float best = 1e9;
  for (int r = 0; r < 5; r++) { hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
  return best * 1e3f;
}
int main() {
  int* out; hipMalloc(&out, 1 << 24);
  Big big{}; 
  for (int nb : {1024, 10000, 40000}) {
    printf("blocks %6d: bigarg+lds9k %7.1f us | smallarg lds9k %7.1f us | smallarg lds37k %7.1f us | smallarg lds256B %7.1f us | 64-thread %7.1f us\n", nb,
           timeit([&] { hipLaunchKernelGGL(k_empty<9216>, dim3(nb), dim3(256), 0, 0, big, out); }),
           timeit([&] { hipLaunchKernelGGL(k_small<9216>, dim3(nb), dim3(256), 0, 0, out, 0); }),
           timeit([&] { hipLaunchKernelGGL(k_small<37376>, dim3(nb), dim3(256), 0, 0, out, 0); }),
           timeit([&] { hipLaunchKernelGGL(k_small<256>, dim3(nb), dim3(256), 0, 0, out, 0); }),
           timeit([&] { hipLaunchKernelGGL(k_wave, dim3(nb), dim3(64), 0, 0, out, 0); }));
  }
  return 0;
}
