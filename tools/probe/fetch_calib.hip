// FETCH_SIZE / WRITE_SIZE calibration by access pattern (tools, not product).  MI355X_MICROARCH.md (HBM section): on gfx950
// FETCH_SIZE reports half of the bytes of a wide coalesced streaming read and "other access widths are uncalibrated: calibrate
// on a known byte count in your own access pattern".  The pass's largest readers are GATHERS of 64-byte records (w6_emit,
// d1w_refine, d2g / d2w_call: one LeadRec per lane at an unrelated address); this probe reads / writes a known byte count in four
// patterns over arrays far larger than the 256-MiB Infinity Cache:
//   k_stream16   16 B per lane, coalesced (the guide's calibration case)
//   k_stream4    4 B per lane, coalesced
//   k_gather64   one 64-byte record per lane at a random record index (4 x 16 B), every record read once
//   k_gather64h  the same, but only every other record of the array is read (the neighbour in the 128-B line is never wanted)
//   k_scatter64  one 64-byte record per lane written at a random record index
// build + run:  hipcc --offload-arch=gfx950 -O3 tools/probe/fetch_calib.hip -o /tmp/fetch_calib
//               rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out_f -o p --output-format csv -- /tmp/fetch_calib   (and WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void k_stream16(const uint4* a, size_t n, uint32_t* sink) {
  uint32_t acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = a[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_stream4(const uint32_t* a, size_t n, uint32_t* sink) {
  uint32_t acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += a[i];
  if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_gather64(const uint4* rec, const uint32_t* idx, size_t n, uint32_t* sink) {
  uint32_t acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint4* r = rec + (size_t)idx[i] * 4;
    const uint4 a = r[0], b = r[1], c = r[2], d = r[3];
    acc += a.x ^ b.y ^ c.z ^ d.w;
  }
  if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_scatter64(uint4* rec, const uint32_t* idx, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint4* r = rec + (size_t)idx[i] * 4;
    const uint4 v = make_uint4((uint32_t)i, 1, 2, 3);
    r[0] = v; r[1] = v; r[2] = v; r[3] = v;
  }
}
int main() {
  const size_t NREC = (size_t)24 << 20;          // 24 M records of 64 B = 1.5 GiB
  uint4* rec; uint32_t *idx, *idxh, *sink;
  CK(hipMalloc(&rec, NREC * 64)); CK(hipMalloc(&idx, NREC * 4)); CK(hipMalloc(&idxh, NREC / 2 * 4)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(rec, 1, NREC * 64));
  std::vector<uint32_t> h(NREC); std::iota(h.begin(), h.end(), 0u);
  std::mt19937_64 g(7); std::shuffle(h.begin(), h.end(), g);
  CK(hipMemcpy(idx, h.data(), NREC * 4, hipMemcpyHostToDevice));
  std::vector<uint32_t> hh(NREC / 2); for (size_t i = 0; i < NREC / 2; i++) hh[i] = (uint32_t)(2 * i); std::shuffle(hh.begin(), hh.end(), g);
  CK(hipMemcpy(idxh, hh.data(), NREC / 2 * 4, hipMemcpyHostToDevice));
  const dim3 grid(256 * 16), block(256);
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(k_stream16, grid, block, 0, 0, rec, NREC * 4, sink);
    hipLaunchKernelGGL(k_stream4, grid, block, 0, 0, (const uint32_t*)rec, NREC * 16, sink);
    hipLaunchKernelGGL(k_gather64, grid, block, 0, 0, rec, idx, NREC, sink);
    hipLaunchKernelGGL(k_gather64, grid, block, 0, 0, rec, idxh, NREC / 2, sink);       // "gather64h": second launch of the pair
    hipLaunchKernelGGL(k_scatter64, grid, block, 0, 0, rec, idx, NREC);
    CK(hipDeviceSynchronize());
  }
  printf("{\"records\": %zu, \"bytes\": {\"k_stream16\": %zu, \"k_stream4\": %zu, \"k_gather64 (1st of a pair)\": %zu, \"k_gather64 (2nd: every other record)\": %zu, \"k_scatter64 (written)\": %zu, \"k_scatter64 (index read)\": %zu}}\n",
         NREC, NREC * 64, NREC * 64, NREC * 64 + NREC * 4, NREC / 2 * 64 + NREC / 2 * 4, NREC * 64, NREC * 4);
  return 0;
}
