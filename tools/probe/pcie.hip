// PCIe result-path probe (tools, not product): D2H copy bandwidth from HBM to pinned host memory by size, and the rate of
// kernel-side stores straight into pinned host memory (1 B / 16 B per lane) - the two ways the library returns results.
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/pcie.hip -o /tmp/pcie && /tmp/pcie
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void st1(uint8_t* dst, const uint8_t* src, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i]; }
__global__ void st16(uint4* dst, const uint4* src, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i]; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t MAXB = 64u << 20;
  uint8_t *d, *h; CK(hipMalloc(&d, MAXB)); CK(hipHostMalloc(&h, MAXB, hipHostMallocDefault)); CK(hipMemset(d, 1, MAXB));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  printf("{\"d2h_copy\": [");
  const size_t sizes[] = {64u << 10, 1u << 20, 4u << 20, 8u << 20, 16u << 20, 32u << 20, 64u << 20};
  for (int k = 0; k < 7; k++) {
    const size_t n = sizes[k];
    for (int w = 0; w < 3; w++) CK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    const int reps = 20;
    CK(hipEventRecord(a, s));
    for (int r = 0; r < reps; r++) CK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s));
    CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    const double t0 = now(); CK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); const double one = now() - t0;
    printf("%s{\"bytes\": %zu, \"GBps_back_to_back\": %.2f, \"one_copy_incl_sync_us\": %.1f}", k ? ", " : "", n, n * (double)reps / (ms * 1e-3) / 1e9, one * 1e6);
  }
  printf("], \"kernel_stores_to_pinned\": [");
  uint8_t* hd = nullptr; CK(hipHostGetDevicePointer((void**)&hd, h, 0));
  for (int mode = 0; mode < 2; mode++)
    for (int gk = 0; gk < 3; gk++) {
      const size_t n = 16u << 20; const int grid = gk == 0 ? 256 : gk == 1 ? 1024 : 4096;
      for (int w = 0; w < 2; w++) { if (mode) hipLaunchKernelGGL(st16, dim3(grid), dim3(256), 0, s, (uint4*)hd, (const uint4*)d, n / 16); else hipLaunchKernelGGL(st1, dim3(grid), dim3(256), 0, s, hd, d, n); }
      CK(hipStreamSynchronize(s));
      CK(hipEventRecord(a, s));
      for (int r = 0; r < 5; r++) { if (mode) hipLaunchKernelGGL(st16, dim3(grid), dim3(256), 0, s, (uint4*)hd, (const uint4*)d, n / 16); else hipLaunchKernelGGL(st1, dim3(grid), dim3(256), 0, s, hd, d, n); }
      CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
      float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
      printf("%s{\"bytes_per_lane\": %d, \"grid\": %d, \"GBps\": %.2f}", (mode || gk) ? ", " : "", mode ? 16 : 1, grid, n * 5.0 / (ms * 1e-3) / 1e9);
    }
  printf("]}\n");
  return 0;
}
