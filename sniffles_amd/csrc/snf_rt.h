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

#include <dlfcn.h>

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

// ---- roctx ranges (the reference has per-stage debug output; here the stages of a pass show up as named
// ranges in rocprofv3 --marker-trace / omnitrace).  libroctx64 is looked up at run time and only when SNF_ROCTX=1, so the
// library has no link-time dependency on the tracer and the default path pays one predictable branch per range.
struct Roctx {
  typedef int (*push_t)(const char*); typedef int (*pop_t)();
  push_t push = nullptr; pop_t pop = nullptr;
  static Roctx& get() {
    static Roctx r = [] {
      Roctx q;
      const char* e = getenv("SNF_ROCTX");
      if (e && atoi(e) != 0) {
        void* h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
        if (h) { q.push = (push_t)dlsym(h, "roctxRangePushA"); q.pop = (pop_t)dlsym(h, "roctxRangePop"); }
        if (!q.push || !q.pop) { q.push = nullptr; q.pop = nullptr; fprintf(stderr, "[sniffles_amd] SNF_ROCTX=1 but libroctx64.so could not be loaded\n"); }
      }
      return q;
    }();
    return r;
  }
};
struct TraceRange {
  bool on;
  explicit TraceRange(const char* name) : on(Roctx::get().push != nullptr) { if (on) Roctx::get().push(name); }
  ~TraceRange() { if (on) Roctx::get().pop(); }
  TraceRange(const TraceRange&) = delete; TraceRange& operator=(const TraceRange&) = delete;
};
#define SNF_TRACE(name) ::snf::TraceRange _snf_trace_##__LINE__(name)

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

// loads / stores that go to L2 (agent scope): the words blocks of one kernel publish to each other (snf_fused.h chain_scan)
SNF_HD unsigned long long ld_agent_u64(const unsigned long long* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  return *p;
#endif
}
SNF_HD void st_agent_u64(unsigned long long* p, unsigned long long x) {
#if defined(__HIP_DEVICE_COMPILE__)
  __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  *p = x;
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
