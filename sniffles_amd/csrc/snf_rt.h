// snf_rt.h - thin runtime layer over HIP (gfx950 only): error handling, atomics usable from kernel bodies, the
// kernel-definition macro.  There is one build of these sources: hipcc for the product.  The GPU-less test tier
// (tests/emu/simt) compiles the SAME sources with g++ against a stand-in for <hip/hip_runtime.h> - nothing in here knows
// about it; the `__HIP_DEVICE_COMPILE__` branches below are the usual host / device halves of __host__ __device__ code.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>
#define SNF_HD __host__ __device__ __forceinline__
#define SNF_D __device__ __forceinline__

typedef unsigned __int128 u128;
typedef __int128 i128;

namespace snf {

struct Error {
  std::string msg;
};
[[noreturn]] inline void fail(const std::string& m) { throw Error{m}; }

#define SNF_HIP(expr)                                                                         \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess)                                                                     \
      ::snf::fail(std::string(#expr) + ": " + hipGetErrorString(_e) + " (" + __FILE__ + ":" + \
                  std::to_string(__LINE__) + ")");                                            \
  } while (0)

// ---- atomics usable from kernel bodies -------------------------------------------------------
SNF_HD unsigned long long atomic_add_u64(unsigned long long* p, unsigned long long v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return atomicAdd(p, v);
#else
  unsigned long long o = *p;
  *p = o + v;
  return o;
#endif
}
SNF_HD uint32_t atomic_fetch_or_u32(uint32_t* p, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return atomicOr(p, v);
#else
  const uint32_t old = *p; *p = old | v; return old;
#endif
}
SNF_HD void atomic_or_i32(int* p, int v) {
#if defined(__HIP_DEVICE_COMPILE__)
  atomicOr(p, v);
#else
  *p |= v;
#endif
}

// ---- kernel definition / launch ---------------------------------------------------------------
// A kernel is a body `void name##_body(int64_t i, const View& v)`; SNF_KERNEL wraps it.
#define SNF_KERNEL(name, VIEW)                                                     \
  __global__ void __launch_bounds__(256) name(const VIEW v, int64_t n) {           \
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;                    \
    if (i < n) name##_body(i, v);                                                  \
  }

}  // namespace snf
