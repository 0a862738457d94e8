// snf_ctx.h - persistent per-device staging for the stand-alone entry points (edit distance, combine): one HBM arena, one
// pinned host mirror of the same layout and one stream, all grow-only and process-wide.  A call lays its arrays out in the
// arena (inputs first, scratch behind), fills the mirror, moves the input part with ONE copy, launches on the arena's
// stream and reads its results back from the mirror - no hipMalloc / hipFree and no per-array copies once the arena has
// reached its working size.
#pragma once
#include "snf_rt.h"

#include <mutex>

namespace snf {

#define SNF_MAX_DEVICES 16
struct DevArena {                // one per (entry point, device)
  std::mutex mu;                 // one call at a time per arena
  int device = -1;
  uint8_t* d = nullptr; uint8_t* h = nullptr; size_t cap = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;   // around the kernel(s) of the last call
  double last_kernel_ms = 0; long long last_stats[4] = {0, 0, 0, 0};
  bool ensure(int dev, size_t bytes) {
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || dev < 0 || dev >= nd) return false;   // no HIP device: the call fails
    if (hipSetDevice(dev) != hipSuccess) return false;
    if (!stream && hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) return false;
    if (!ev0 && (hipEventCreate(&ev0) != hipSuccess || hipEventCreate(&ev1) != hipSuccess)) return false;
    device = dev;
    if (bytes <= cap && d) return true;
    if (d) { (void)hipStreamSynchronize(stream); (void)hipFree(d); (void)hipHostFree(h); d = h = nullptr; }
    cap = bytes + bytes / 4 + (1u << 20);
    if (hipMalloc((void**)&d, cap) != hipSuccess) { d = nullptr; cap = 0; return false; }
    if (hipHostMalloc((void**)&h, cap, hipHostMallocDefault) != hipSuccess) { (void)hipFree(d); d = h = nullptr; cap = 0; return false; }
    return true;
  }
};

// byte offsets of a call's arrays inside the arena (256-byte aligned)
struct ArenaLayout {
  size_t at = 0;
  template <class T> size_t add(size_t n) { at = (at + 255) & ~(size_t)255; const size_t o = at; at += (n ? n : 1) * sizeof(T); return o; }
};

}  // namespace snf
