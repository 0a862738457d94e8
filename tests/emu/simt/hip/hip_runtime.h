// TEST-ONLY stand-in for <hip/hip_runtime.h>: runs the gfx950 WAVE kernels of sniffles_amd/csrc (the `#ifndef SNF_EMU`
// halves of the headers, which the serial emulation in tests/emu/emu.py never sees) on the host, in the GPU-less build
// container, with g++.  Never shipped, never on the product's include path (only tests/emu/simt.py adds this directory).
//
// Model: every thread of a workgroup is a fibre (own stack, cooperative switches, x86-64 only).  A lane runs until it
// reaches a cross-lane operation or a barrier, then the next lane of its wave (wave barrier) or of the workgroup
// (__syncthreads) runs.  Lanes of a wave are therefore NOT in lock step between synchronisation points - a kernel that
// relies on implicit lock step for LDS visibility (instead of a wave barrier) fails here, which is the stricter reading.
//   * __shfl / __shfl_xor / __shfl_up / __shfl_down / __ballot / readlane / readfirstlane exchange through a per-wave
//     buffer bracketed by two wave barriers; all 64 lanes must take part (a divergent call is reported as a deadlock);
//   * __builtin_amdgcn_update_dpp implements the controls the kernels use: row_shr:1..15 (0x111..0x11f), row_bcast:15
//     (0x142), row_bcast:31 (0x143), quad_perm (0x00..0xff), row_mirror (0x140), row_half_mirror (0x141), wave_shr / wave_ror /
//     wave_shl / wave_rol by one lane (0x138, 0x13C, 0x130, 0x134), with row_mask and
//     bound_ctrl = false semantics (disabled / sourceless lanes keep `old`);
//   * atomics are plain read-modify-writes (switches are cooperative, so they are atomic by construction);
//   * __shared__ is a function-local static: one workgroup at a time;
//   * LOCK STEP for the kernels that need it (x_big: all 64 lanes of a wave run one serial body on the same data, so a
//     read-modify-write of memory happens once on the GPU): between two synchronisation points every lane runs against
//     the memory image of the start of that interval - writes to device memory are caught page-wise (mprotect + SIGSEGV),
//     the kernel's __shared__ arrays (found in the library's symbol table) are compared against a snapshot - its changes
//     are taken aside and the image restored before the next lane runs; when all lanes have reached the synchronisation
//     point the changes of all lanes are applied (identical or disjoint in race-free code).
#pragma once
#if !defined(__x86_64__)
#error "the fibre switch of the SIMT test shim is written for x86-64"
#endif
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <utility>
#include <vector>
#include <dlfcn.h>
#include <elf.h>
#include <signal.h>
#include <sys/mman.h>
#include <unistd.h>
#include <execinfo.h>

#define __HIP_DEVICE_COMPILE__ 1
#define __HIPCC__ 1
#define __host__
#define __device__
#define __global__ static
#define __shared__ static
#define __forceinline__ inline
#define __launch_bounds__(...)

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }
static inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }
typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline const char* hipGetErrorString(hipError_t) { return "hip is not available in the SIMT shim"; }

inline dim3 threadIdx, blockIdx, blockDim, gridDim;   // one scheduler per library: shared by its translation units

extern "C" void snf_simt_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.local snf_simt_switch
.type snf_simt_switch,@function
snf_simt_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size snf_simt_switch,.-snf_simt_switch
)");

namespace simt {

constexpr int MAX_THREADS = 1024;
constexpr size_t STACK_BYTES = 256 * 1024;

enum : uint8_t { RUN = 0, AT_WAVE = 1, READING = 2, AT_BLOCK = 3 };
struct Fibre { void* sp; bool done; };
// per wave: the group of lanes whose cross-lane operation was released last, how many of them still read the exchange
// buffer, and how many unfinished lanes are not blocked at a wave operation or the workgroup barrier
struct WaveCtl { uint64_t grp; int readers; int busy; };

inline Fibre g_fib[MAX_THREADS];
inline uint8_t g_st[MAX_THREADS];
inline const void* g_site[MAX_THREADS];
inline uint64_t g_x[MAX_THREADS];
inline WaveCtl g_wave[MAX_THREADS / 64];
inline char* g_stacks = nullptr;
inline void* g_main_sp = nullptr;
inline int g_cur = 0, g_n = 0, g_done = 0, g_block_waiting = 0, g_block_gen = 0;
inline std::function<void()> g_body;
inline const char* g_kernel = "";
inline bool g_uniform = false, g_abandon = false;
inline unsigned long long g_progress = 0;
inline std::mutex g_launch_mutex;
inline unsigned long long g_n_switch = 0, g_n_wave_ops = 0, g_n_block_syncs = 0, g_n_divergent_ops = 0, g_n_launches = 0, g_n_unmodelled = 0;

[[noreturn]] inline void die(const char* what) {
  std::fprintf(stderr, "simt shim: %s (kernel %s, block %u, thread %d of %d); frames for addr2line -e <this library>:\n", what, g_kernel,
               blockIdx.x, g_cur, g_n);
  { const int b = g_cur & ~63; std::fprintf(stderr, "  wave group %016llx readers %d busy %d; lanes (state:site):", (unsigned long long)g_wave[b >> 6].grp, g_wave[b >> 6].readers, g_wave[b >> 6].busy);
    for (int l = 0; l < 64; l++) std::fprintf(stderr, " %d%s:%p", (int)g_st[b + l], g_fib[b + l].done ? "d" : "", g_site[b + l]); std::fprintf(stderr, "\n"); }
  void* frames[24];
  backtrace_symbols_fd(frames, backtrace(frames, 24), 2);
  std::abort();
}

inline void switch_to(int k) {
  const int old = g_cur;
  g_cur = k; threadIdx.x = (unsigned)k; g_n_switch++;
  snf_simt_switch(&g_fib[old].sp, g_fib[k].sp);
}
// next unfinished lane of the same wave / of the workgroup.  SNF_SIMT_ORDER=reverse | random[:seed] changes which lane runs
// next (default: ascending round robin): a result that depends on it is a race between lanes that no barrier orders
inline int g_order = 0;          // 0 ascending, 1 descending, 2 random
inline unsigned long long g_rng = 0x9E3779B97F4A7C15ull;
inline unsigned next_random() { g_rng ^= g_rng << 13; g_rng ^= g_rng >> 7; g_rng ^= g_rng << 17; return (unsigned)(g_rng >> 32); }
inline void read_order() {
  const char* e = std::getenv("SNF_SIMT_ORDER");
  g_order = !e ? 0 : !std::strncmp(e, "reverse", 7) ? 1 : !std::strncmp(e, "random", 6) ? 2 : 0;
  if (g_order == 2) { const char* c = std::strchr(e, ':'); g_rng = 0x9E3779B97F4A7C15ull ^ (c ? std::strtoull(c + 1, nullptr, 0) * 0xD1B54A32D192ED03ull : 0); if (!g_rng) g_rng = 1; }
}
inline void yield_wave() {
  const int w0 = g_cur & ~63, me = g_cur - w0;
  const int off = g_order == 2 ? (int)(next_random() & 63) : 0;
  for (int d = 1; d < 64; d++) {
    const int k = w0 + (g_order == 0 ? (me + d) & 63 : g_order == 1 ? (me - d) & 63 : (me + 1 + ((d - 1 + off) % 63)) & 63);
    if (!g_fib[k].done) { switch_to(k); return; }
  }
}
inline void yield_block() {
  const int off = g_order == 2 ? (int)(next_random() % (unsigned)(g_n > 1 ? g_n - 1 : 1)) : 0;
  for (int d = 1; d < g_n; d++) {
    const int k = g_order == 0 ? (g_cur + d) % g_n : g_order == 1 ? (g_cur - d + g_n) % g_n : (g_cur + 1 + ((d - 1 + off) % (g_n - 1))) % g_n;
    if (!g_fib[k].done) { switch_to(k); return; }
  }
}

// ------------------------------------------------------------------------------------------------ lock-step mode
constexpr size_t PAGE = 4096;
struct DevBlock { char* base; size_t bytes; char* user; bool guard; };   // the mapping without its guard page; what hipMalloc returned
struct PageImg { char* page; std::vector<uint8_t> data; };
struct LdsRegion { char* p; size_t n; std::vector<uint8_t> snap; };
struct LdsImg { int region; std::vector<uint8_t> data; };
inline std::vector<DevBlock> g_dev;            // hipMalloc'd blocks (page aligned, page granular)
inline std::mutex g_dev_mutex;
inline bool g_lockstep = false;                // a uniform kernel is running
inline std::vector<PageImg> u_pre, u_post;     // per interval: image of every touched page at its start; pages as the lanes left them
inline std::vector<char*> u_lane_dirty;        // pages the running lane has made writable
inline std::vector<LdsRegion> u_lds;
inline std::vector<LdsImg> u_lds_post;
inline struct sigaction u_old_segv;
inline pthread_t u_thread;
inline unsigned long long g_n_lockstep_faults = 0, g_n_lockstep_merges = 0, g_n_lockstep_conflicts = 0;

inline void uni_segv(int sig, siginfo_t* si, void* ctx) {
  char* a = (char*)si->si_addr;
  bool ours = false;
  if (g_lockstep && pthread_equal(pthread_self(), u_thread))
    for (const DevBlock& b : g_dev) if (a >= b.base && a < b.base + b.bytes) { ours = true; break; }
  if (!ours) {   // a real fault: hand it to whoever was there before
    sigaction(SIGSEGV, &u_old_segv, nullptr);
    return;
  }
  char* page = (char*)((uintptr_t)a & ~(uintptr_t)(PAGE - 1));
  bool seen = false;
  for (const PageImg& q : u_pre) if (q.page == page) { seen = true; break; }
  if (!seen) u_pre.push_back(PageImg{page, std::vector<uint8_t>(page, page + PAGE)});
  u_lane_dirty.push_back(page);
  mprotect(page, PAGE, PROT_READ | PROT_WRITE);
  g_n_lockstep_faults++;
}
// the running lane has reached a synchronisation point (or its end): take its changes aside, restore the interval's image
inline void uni_lane_end() {
  if (!g_lockstep) return;
  for (char* page : u_lane_dirty) {
    u_post.push_back(PageImg{page, std::vector<uint8_t>(page, page + PAGE)});
    for (const PageImg& q : u_pre) if (q.page == page) { std::memcpy(page, q.data.data(), PAGE); break; }
    mprotect(page, PAGE, PROT_READ);
  }
  u_lane_dirty.clear();
  for (size_t r = 0; r < u_lds.size(); r++) {
    LdsRegion& R = u_lds[r];
    if (std::memcmp(R.p, R.snap.data(), R.n) != 0) {
      u_lds_post.push_back(LdsImg{(int)r, std::vector<uint8_t>(R.p, R.p + R.n)});
      std::memcpy(R.p, R.snap.data(), R.n);
    }
  }
}
// every lane is at a synchronisation point: the interval's changes become the memory image of the next interval
inline void uni_merge() {
  if (!g_lockstep) return;
  if (!u_lane_dirty.empty()) die("lock step: a lane still holds writable pages at a merge");
  for (const PageImg& q : u_pre) mprotect(q.page, PAGE, PROT_READ | PROT_WRITE);
  for (const PageImg& post : u_post) {
    const PageImg* pre = nullptr;
    for (const PageImg& q : u_pre) if (q.page == post.page) { pre = &q; break; }
    for (size_t i = 0; i < PAGE; i++) if (post.data[i] != pre->data[i]) {
      if ((uint8_t)post.page[i] != pre->data[i] && (uint8_t)post.page[i] != post.data[i]) g_n_lockstep_conflicts++;   // two lanes, two values
      post.page[i] = (char)post.data[i];
    }
  }
  for (const PageImg& q : u_pre) mprotect(q.page, PAGE, PROT_READ);
  for (const LdsImg& post : u_lds_post) {
    LdsRegion& R = u_lds[post.region];
    for (size_t i = 0; i < R.n; i++) if (post.data[i] != R.snap[i]) {
      if ((uint8_t)R.p[i] != R.snap[i] && (uint8_t)R.p[i] != post.data[i]) g_n_lockstep_conflicts++;
      R.p[i] = (char)post.data[i];
    }
  }
  for (LdsRegion& R : u_lds) std::memcpy(R.snap.data(), R.p, R.n);
  u_pre.clear(); u_post.clear(); u_lds_post.clear();
  g_n_lockstep_merges++;
}
// the __shared__ arrays of kernel `name` ("x_big<0>"): local OBJECT symbols "_ZZ...<len><ident>ILi<arg>E..." of this library
inline bool uni_find_lds(const char* name, std::vector<LdsRegion>& out) {
  Dl_info di;
  if (!dladdr((void*)&g_cur, &di) || !di.dli_fname) return false;
  std::string ident(name), arg;
  const size_t lt = ident.find('<');
  if (lt != std::string::npos) { arg = ident.substr(lt + 1, ident.find('>') - lt - 1); ident = ident.substr(0, lt); }
  std::string pat = std::to_string(ident.size()) + ident + (arg.empty() ? "" : "ILi" + arg + "E");
  FILE* f = std::fopen(di.dli_fname, "rb");
  if (!f) return false;
  std::vector<char> img;
  std::fseek(f, 0, SEEK_END); const long sz = std::ftell(f); std::fseek(f, 0, SEEK_SET);
  img.resize((size_t)sz);
  const bool ok = std::fread(img.data(), 1, (size_t)sz, f) == (size_t)sz;
  std::fclose(f);
  if (!ok || sz < (long)sizeof(Elf64_Ehdr)) return false;
  const Elf64_Ehdr* eh = (const Elf64_Ehdr*)img.data();
  const Elf64_Shdr* sh = (const Elf64_Shdr*)(img.data() + eh->e_shoff);
  bool found_symtab = false;
  for (int k = 0; k < eh->e_shnum; k++) {
    if (sh[k].sh_type != SHT_SYMTAB) continue;
    found_symtab = true;
    const Elf64_Sym* sym = (const Elf64_Sym*)(img.data() + sh[k].sh_offset);
    const char* str = img.data() + sh[sh[k].sh_link].sh_offset;
    const size_t n = sh[k].sh_size / sizeof(Elf64_Sym);
    for (size_t i = 0; i < n; i++) {
      if (ELF64_ST_TYPE(sym[i].st_info) != STT_OBJECT || sym[i].st_size == 0) continue;
      const char* nm = str + sym[i].st_name;
      if (std::strncmp(nm, "_ZZ", 3) != 0 || !std::strstr(nm, pat.c_str())) continue;
      LdsRegion R; R.p = (char*)di.dli_fbase + sym[i].st_value; R.n = sym[i].st_size; R.snap.assign(R.p, R.p + R.n);
      out.push_back(R);
    }
  }
  return found_symtab;
}
inline bool uni_begin(const char* name) {
  u_lds.clear();
  if (!uni_find_lds(name, u_lds)) return false;
  u_thread = pthread_self();
  struct sigaction sa;
  std::memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = uni_segv; sa.sa_flags = SA_SIGINFO | SA_NODEFER;
  sigemptyset(&sa.sa_mask);
  sigaction(SIGSEGV, &sa, &u_old_segv);
  g_dev_mutex.lock();    // no allocation or release while pages are protected
  for (const DevBlock& b : g_dev) mprotect(b.base, b.bytes, PROT_READ);
  g_lockstep = true;
  return true;
}
inline void uni_end() {
  g_lockstep = false;
  for (const DevBlock& b : g_dev) mprotect(b.base, b.bytes, PROT_READ | PROT_WRITE);
  g_dev_mutex.unlock();
  sigaction(SIGSEGV, &u_old_segv, nullptr);
}

struct Idle {   // a waiting loop that sees no progress anywhere for several rounds over all fibres is a deadlock
  unsigned long long p0 = g_progress; long n = 0;
  void tick(const char* what) { if (g_progress != p0) { p0 = g_progress; n = 0; } else if (++n > 8 * MAX_THREADS) die(what); }
};

// Cross-lane operations.  On the GPU the lanes that take part are the active lanes at that instruction; here a lane that
// arrives publishes its operand and waits.  The operation is released when every unfinished lane of the wave is blocked (at
// a cross-lane operation or the workgroup barrier): normally all of them at the same call site - the convergent case, the
// group is the whole wave.  If the blocked lanes sit at different call sites (the operation is inside divergent control
// flow), the site that comes FIRST in the source goes first with the lanes that are there, the way structured code executes
// a branch before the code behind its join.
inline void wave_try_release(int w) {
  WaveCtl& W = g_wave[w];
  if (W.readers || W.busy) return;
  const int b = w * 64;
  const void* best = nullptr; int n_wait = 0;
  for (int l = 0; l < 64; l++) if (!g_fib[b + l].done && g_st[b + l] == AT_WAVE) { n_wait++; if (!best || g_site[b + l] < best) best = g_site[b + l]; }
  if (!best) return;   // everybody is at the workgroup barrier
  uint64_t grp = 0;
  for (int l = 0; l < 64; l++) if (!g_fib[b + l].done && g_st[b + l] == AT_WAVE && g_site[b + l] == best) { grp |= 1ull << l; g_st[b + l] = READING; }
  uni_merge();
  W.grp = grp; W.readers = __builtin_popcountll(grp); W.busy += W.readers;
  if (W.readers != n_wait) {
    g_n_divergent_ops++;
    if (std::getenv("SNF_SIMT_TRACE")) {
      std::fprintf(stderr, "simt: %s block %u wave %d: subset %016llx released at %p; waiting:", g_kernel, blockIdx.x, w, (unsigned long long)grp, best);
      for (int l = 0; l < 64; l++) if (!g_fib[b + l].done && g_st[b + l] == AT_WAVE) std::fprintf(stderr, " %d:%p", l, g_site[b + l]);
      std::fprintf(stderr, "\n");
    }
  }
  g_progress++; g_n_wave_ops++;
}
// publish `bits`, wait for the release; returns the mask of the lanes that take part.  wave_op_end() after the reads.
inline uint64_t wave_op_begin(const void* site, uint64_t bits) {
  const int t = g_cur, w = t >> 6;
  WaveCtl& W = g_wave[w];
  uni_lane_end();
  { Idle idle; while (W.readers > 0 && ((W.grp >> (t & 63)) & 1)) { yield_wave(); idle.tick("deadlock: a released group never finished reading"); } }
  g_x[t] = bits; g_site[t] = site; g_st[t] = AT_WAVE; W.busy--;
  Idle idle;
  for (;;) {
    wave_try_release(w);
    if (g_st[t] == READING) break;
    yield_wave();
    idle.tick("deadlock at a cross-lane operation");
  }
  return W.grp;
}
inline void wave_op_end() { g_st[g_cur] = RUN; g_wave[g_cur >> 6].readers--; }

inline void block_try_release() {
  if (g_block_waiting && g_block_waiting == g_n - g_done) {
    uni_merge();
    for (int t = 0; t < g_n; t++) if (!g_fib[t].done && g_st[t] == AT_BLOCK) { g_st[t] = RUN; g_wave[t >> 6].busy++; }   // running again, even if not yet scheduled
    g_block_waiting = 0; g_block_gen++; g_progress++; g_n_block_syncs++;
  }
}
inline void block_sync() {
  if (g_uniform && !g_lockstep) {
    // lock step could not be set up (no symbol table?): the launch is abandoned and counted, the test that sees the count
    // does not compare results
    g_n_unmodelled++; g_abandon = true;
    void* dummy; snf_simt_switch(&dummy, g_main_sp);
  }
  uni_lane_end();
  const int t = g_cur, gen = g_block_gen;
  g_st[t] = AT_BLOCK; g_wave[t >> 6].busy--; g_block_waiting++;
  block_try_release();
  Idle idle;
  while (g_block_gen == gen) { yield_block(); idle.tick("deadlock at __syncthreads"); }
}

inline void fibre_main() {
  g_body();
  uni_lane_end();
  g_fib[g_cur].done = true; g_done++; g_wave[g_cur >> 6].busy--; g_progress++;
  block_try_release();
  if (g_done == g_n) { void* dummy; snf_simt_switch(&dummy, g_main_sp); }
  yield_block();
  die("a finished fibre was resumed");
}

// one workgroup of `nthreads` (a multiple of 64) running body() per thread
inline void run_block(int nthreads) {
  if (nthreads <= 0 || nthreads > MAX_THREADS || (nthreads & 63)) die("workgroup size must be a multiple of 64, at most 1024");
  if (!g_stacks) { g_stacks = (char*)std::malloc((size_t)MAX_THREADS * STACK_BYTES); if (!g_stacks) die("no memory for the fibre stacks"); }
  g_n = nthreads; g_done = 0; g_block_waiting = 0; g_block_gen = 0;
  for (int w = 0; w < MAX_THREADS / 64; w++) g_wave[w] = WaveCtl{0, 0, 64};
  for (int t = 0; t < nthreads; t++) {
    uintptr_t top = ((uintptr_t)(g_stacks + (size_t)(t + 1) * STACK_BYTES)) & ~(uintptr_t)15;
    void** sp = (void**)(top - 64);                 // six callee-saved registers, the entry address, padding
    for (int k = 0; k < 6; k++) sp[k] = nullptr;
    sp[6] = (void*)&fibre_main; sp[7] = nullptr;
    g_fib[t].sp = (void*)sp; g_fib[t].done = false; g_st[t] = RUN;
  }
  g_cur = g_order == 0 ? 0 : g_order == 1 ? nthreads - 1 : (int)(next_random() % (unsigned)nthreads);
  threadIdx = dim3((unsigned)g_cur, 0, 0);
  snf_simt_switch(&g_main_sp, g_fib[g_cur].sp);
  if (!g_abandon) uni_merge();
  else { u_pre.clear(); u_post.clear(); u_lds_post.clear(); u_lane_dirty.clear(); }
}

template <class K, class... A>
inline void launch(const char* name, K kernel, unsigned grid, unsigned block, A... args) {
  std::lock_guard<std::mutex> one_launch_at_a_time(g_launch_mutex);   // host threads may drive several batches
  gridDim = dim3(grid); blockDim = dim3(block);
  read_order();
  g_kernel = name; g_uniform = std::strncmp(name, "x_big", 5) == 0; g_abandon = false; g_n_launches++;
  g_body = [&]() { kernel(args...); };
  const bool lockstep = g_uniform && block == 64 && !std::getenv("SNF_SIMT_NO_LOCKSTEP") && uni_begin(name);
  // workgroups run one after the other; with SNF_SIMT_ORDER also in descending / shuffled order (the hardware promises none):
  // a kernel that needs an earlier workgroup to be through, or whose result depends on who appended to a list first, shows here
  std::vector<unsigned> blocks(grid);
  for (unsigned b = 0; b < grid; b++) blocks[b] = g_order == 1 ? grid - 1 - b : b;
  if (g_order == 2) for (unsigned b = grid; b > 1; b--) { const unsigned j = next_random() % b; std::swap(blocks[b - 1], blocks[j]); }
  for (unsigned k = 0; k < grid && !g_abandon; k++) { blockIdx = dim3(blocks[k]); run_block((int)block); }
  if (lockstep) uni_end();
  g_body = nullptr; g_kernel = "";
}

template <class T> inline uint64_t to_bits(T v) { static_assert(sizeof(T) <= 8, "cross-lane values are at most 64 bits"); uint64_t b = 0; std::memcpy(&b, &v, sizeof(T)); return b; }
template <class T> inline T from_bits(uint64_t b) { T v; std::memcpy(&v, &b, sizeof(T)); return v; }
inline int lane() { return g_cur & 63; }
inline unsigned alignbyte(unsigned hi, unsigned lo, unsigned sh) { return (unsigned)(((((unsigned long long)hi) << 32) | lo) >> (8 * (sh & 3))); }

// value of lane `src` (0..63) of this wave
template <class T> inline T read_lane_at(const void* site, T v, int src) {
  const uint64_t grp = wave_op_begin(site, to_bits(v));
  src &= 63;
  if (!((grp >> src) & 1)) die("a cross-lane read names a lane that does not take part in the operation");
  const T r = from_bits<T>(g_x[(g_cur & ~63) + src]);
  wave_op_end();
  return r;
}
inline unsigned long long ballot_at(const void* site, bool p) {
  const uint64_t grp = wave_op_begin(site, p ? 1 : 0);
  unsigned long long m = 0;
  const int w0 = g_cur & ~63;
  for (int k = 0; k < 64; k++) if ((grp >> k) & 1) m |= (unsigned long long)(g_x[w0 + k] & 1) << k;
  wave_op_end();
  return m;
}
inline void wave_sync_at(const void* site) { wave_op_begin(site, 0); wave_op_end(); }
inline int dpp_at(const void* site, int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  if (bank_mask != 0xf || bound_ctrl) die("update_dpp: only bank_mask 0xf, bound_ctrl false are modelled");
  const int l = lane(), row = l >> 4;
  int from = -1;   // source lane, -1: none
  if (ctrl >= 0x111 && ctrl <= 0x11f) { const int n = ctrl - 0x110; if ((l & 15) >= n) from = l - n; }
  else if (ctrl == 0x142) { if (row >= 1) from = row * 16 - 1; }
  else if (ctrl == 0x143) { if (row >= 2) from = 31; }
  else if (ctrl >= 0 && ctrl <= 0xff) from = (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);      // quad_perm
  else if (ctrl == 0x140) from = (l & ~15) | (15 - (l & 15));                                // row_mirror
  else if (ctrl == 0x141) from = (l & ~7) | (7 - (l & 7));                                   // row_half_mirror
  // whole-wave shifts / rotates by one lane (GFX9 only; checked on the MI355X by tools/probe/dpp_wave.hip)
  else if (ctrl == 0x138) { if (l >= 1) from = l - 1; }                                      // wave_shr:1
  else if (ctrl == 0x13C) from = (l + 63) & 63;                                              // wave_ror:1
  else if (ctrl == 0x130) { if (l <= 62) from = l + 1; }                                     // wave_shl:1
  else if (ctrl == 0x134) from = (l + 1) & 63;                                               // wave_rol:1
  else die("update_dpp: control not modelled");
  const uint64_t grp = wave_op_begin(site, to_bits(src));
  int r = old;
  if (((row_mask >> row) & 1) && from >= 0 && ((grp >> from) & 1)) r = from_bits<int>(g_x[(g_cur & ~63) + from]);
  wave_op_end();
  return r;
}

}  // namespace simt

// A call site is its position in the preprocessed source (__COUNTER__): robust against the compiler duplicating or merging
// code, and ordered the way the source is.
#define SIMT_SITE ((const void*)(uintptr_t)(__COUNTER__ + 1))
#define __syncthreads() ::simt::block_sync()
#define __builtin_amdgcn_s_barrier() ::simt::block_sync()
#define __builtin_amdgcn_wave_barrier() ::simt::wave_sync_at(SIMT_SITE)
#define __builtin_amdgcn_readfirstlane(x) ::simt::read_first_lane_at(SIMT_SITE, (x))
#define __builtin_amdgcn_readlane(x, k) ::simt::read_lane_at(SIMT_SITE, (x), (k))
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) ::simt::dpp_at(SIMT_SITE, (old), (src), (ctrl), (rm), (bm), (bc))
#define __builtin_amdgcn_s_memtime() 0ull
#define __builtin_amdgcn_fence(...) ((void)0)      /* one thread runs at a time: program order is the memory order */
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, x, order, scope) ((void)(*(p) = (x)))
#define __builtin_amdgcn_s_sleep(n) ((void)0)
#define __builtin_amdgcn_alignbyte(hi, lo, sh) ::simt::alignbyte((hi), (lo), (sh))   /* v_alignbyte_b32 */
#define __shfl(...) ::simt::shfl_at(SIMT_SITE, __VA_ARGS__)
#define __shfl_xor(...) ::simt::shfl_xor_at(SIMT_SITE, __VA_ARGS__)
#define __shfl_up(...) ::simt::shfl_up_at(SIMT_SITE, __VA_ARGS__)
#define __shfl_down(...) ::simt::shfl_down_at(SIMT_SITE, __VA_ARGS__)
#define __ballot(p) ::simt::ballot_at(SIMT_SITE, (p) != 0)

namespace simt {
// the first lane that takes part
template <class T> inline T read_first_lane_at(const void* site, T v) {
  const uint64_t grp = wave_op_begin(site, to_bits(v));
  const T r = from_bits<T>(g_x[(g_cur & ~63) + __builtin_ctzll(grp)]);
  wave_op_end();
  return r;
}
template <class T> inline T shfl_at(const void* site, T v, int src, int width = 64) { (void)width; return read_lane_at(site, v, src); }
template <class T> inline T shfl_xor_at(const void* site, T v, int m, int width = 64) { (void)width; return read_lane_at(site, v, lane() ^ m); }
template <class T> inline T shfl_up_at(const void* site, T v, unsigned d, int width = 64) { (void)width; const int l = lane(); return read_lane_at(site, v, l >= (int)d ? l - (int)d : l); }
template <class T> inline T shfl_down_at(const void* site, T v, unsigned d, int width = 64) { (void)width; const int l = lane(); return read_lane_at(site, v, l + (int)d < 64 ? l + (int)d : l); }
}  // namespace simt

inline unsigned __lane_id() { return (unsigned)::simt::lane(); }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
inline int __ffsll(long long x) { return __builtin_ffsll(x); }
inline int __ffs(int x) { return __builtin_ffs(x); }
inline long long __double_as_longlong(double d) { return ::simt::from_bits<long long>(::simt::to_bits(d)); }
inline int __double2hiint(double d) { return (int)(::simt::to_bits(d) >> 32); }
inline int __double2loint(double d) { return (int)(::simt::to_bits(d) & 0xffffffffull); }
inline double __hiloint2double(int hi, int lo) { return ::simt::from_bits<double>(((uint64_t)(unsigned)hi << 32) | (unsigned)lo); }
inline double __longlong_as_double(long long x) { return ::simt::from_bits<double>(::simt::to_bits(x)); }

template <class T, class U> inline T atomicAdd(T* p, U v) { const T o = *p; *p = (T)(o + (T)v); return o; }
template <class T, class U> inline T atomicSub(T* p, U v) { const T o = *p; *p = (T)(o - (T)v); return o; }
template <class T, class U> inline T atomicOr(T* p, U v) { const T o = *p; *p = (T)(o | (T)v); return o; }
template <class T, class U> inline T atomicAnd(T* p, U v) { const T o = *p; *p = (T)(o & (T)v); return o; }
template <class T, class U> inline T atomicMax(T* p, U v) { const T o = *p; if ((T)v > o) *p = (T)v; return o; }
template <class T, class U> inline T atomicMin(T* p, U v) { const T o = *p; if ((T)v < o) *p = (T)v; return o; }
template <class T, class U> inline T atomicExch(T* p, U v) { const T o = *p; *p = (T)v; return o; }
template <class T, class U, class V> inline T atomicCAS(T* p, U cmp, V val) { const T o = *p; if (o == (T)cmp) *p = (T)val; return o; }

// ------------------------------------------------------------------------------------------------------------------
// Host side of the runtime: device memory is host memory, streams and events are tokens, every operation completes
// before the call returns (one legal serialisation of the stream program).  hipMalloc fills the block with 0xA5 so that a
// kernel relying on fresh device memory being zero fails here as it may on the GPU.
// ------------------------------------------------------------------------------------------------------------------
#include <chrono>
#include <type_traits>

typedef struct simt_event { double t_ms; }* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
#define hipStreamNonBlocking 1u
#define hipEventDisableTiming 2u
#define hipHostMallocDefault 0u
struct hipDeviceProp_t { char name[256]; int multiProcessorCount; size_t totalGlobalMem; int warpSize; char gcnArchName[256]; };

namespace simt { static inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); } }

namespace simt {
// Device blocks end at an inaccessible page (the block is pushed against it, 16-byte granular): a kernel that reads or writes
// behind a buffer faults here instead of picking up whatever the GPU's allocator left there.  SNF_SIMT_NO_GUARD=1 turns it off.
inline void* dev_alloc(size_t n) {
  const size_t data = ((n ? n : 1) + 15) & ~(size_t)15;
  const size_t bytes = (data + PAGE - 1) & ~(PAGE - 1);
  const bool guard = !std::getenv("SNF_SIMT_NO_GUARD");
  char* q = (char*)mmap(nullptr, bytes + (guard ? PAGE : 0), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (q == (char*)MAP_FAILED) return nullptr;
  const char* fill = std::getenv("SNF_SIMT_FILL");          // what fresh device memory holds (default 0xA5): results must not depend on it
  std::memset(q, fill ? (int)std::strtol(fill, nullptr, 0) : 0xA5, bytes);
  if (guard) mprotect(q + bytes, PAGE, PROT_NONE);
  char* user = guard ? q + bytes - data : q;
  std::lock_guard<std::mutex> g(g_dev_mutex);
  g_dev.push_back(DevBlock{q, bytes, user, guard});
  return user;
}
inline bool dev_free(void* p) {
  if (!p) return true;
  std::lock_guard<std::mutex> g(g_dev_mutex);
  for (size_t k = 0; k < g_dev.size(); k++) if (g_dev[k].user == (char*)p) {
    munmap(g_dev[k].base, g_dev[k].bytes + (g_dev[k].guard ? PAGE : 0));
    g_dev[k] = g_dev.back(); g_dev.pop_back();
    return true;
  }
  return false;
}
}  // namespace simt
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) { void* q = simt::dev_alloc(n); if (!q) return 2; *p = (T*)q; return hipSuccess; }
static inline hipError_t hipFree(void* p) { return simt::dev_free(p) ? hipSuccess : 1; }
#define hipHostRegisterDefault 0u
static inline hipError_t hipHostRegister(void*, size_t, unsigned) { return hipSuccess; }
static inline hipError_t hipHostUnregister(void*) { return hipSuccess; }
static inline hipError_t hipHostGetDevicePointer(void** dp, void* hp, unsigned) { *dp = hp; return hipSuccess; }
template <class T> static inline hipError_t hipHostMalloc(T** p, size_t n, unsigned flags = 0) { (void)flags; void* q = std::malloc(n ? n : 1); if (!q) return 2; *p = (T*)q; return hipSuccess; }
static inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::lock_guard<std::mutex> g(simt::g_launch_mutex); if (n) std::memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { std::lock_guard<std::mutex> g(simt::g_launch_mutex); if (n) std::memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int c, size_t n) { std::lock_guard<std::mutex> g(simt::g_launch_mutex); if (n) std::memset(d, c, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int c, size_t n, hipStream_t = nullptr) { std::lock_guard<std::mutex> g(simt::g_launch_mutex); if (n) std::memset(d, c, n); return hipSuccess; }
// SNF_SIMT_DEVICES=n: the stand-in reports n devices (all of them this host; what differs per device in the library is its own
// bookkeeping - arenas, caches, the pacing state of the passes - which is what a test with two device indices exercises)
static inline int simt_device_count() { const char* e = getenv("SNF_SIMT_DEVICES"); const int n = e ? atoi(e) : 1; return n < 1 ? 1 : n; }
static inline int& simt_current_device() { static thread_local int d = 0; return d; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = simt_device_count(); return hipSuccess; }
static inline hipError_t hipSetDevice(int d) { if (d < 0 || d >= simt_device_count()) return 101; simt_current_device() = d; return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = simt_current_device(); return hipSuccess; }
static inline hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t = nullptr) {
  std::lock_guard<std::mutex> g(simt::g_launch_mutex);
  for (size_t r = 0; r < height; r++) std::memmove((char*)d + r * dpitch, (const char*)s + r * spitch, width);
  return hipSuccess;
}
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
  std::memset(p, 0, sizeof *p); std::snprintf(p->name, sizeof p->name, "SIMT test shim"); std::snprintf(p->gcnArchName, sizeof p->gcnArchName, "host");
  p->multiProcessorCount = 4; p->warpSize = 64; p->totalGlobalMem = (size_t)8 << 30; return hipSuccess;
}
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = std::malloc(1); return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = std::malloc(1); return hipSuccess; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 0; *greatest = -1; return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = std::malloc(1); return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { std::free(s); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)std::calloc(1, sizeof(simt_event)); return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipEventDestroy(hipEvent_t e) { std::free(e); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t_ms = simt::now_ms(); return hipSuccess; }
/* graphs: not modelled - a capture cannot begin, so the library's passes stay eager on this tier (snf_lib.hip run_pass) */
typedef struct simtGraph* hipGraph_t; typedef struct simtGraphExec* hipGraphExec_t; typedef struct simtGraphNode* hipGraphNode_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1, hipStreamCaptureModeRelaxed = 2 };
static inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return 801; }
static inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { if (g) *g = nullptr; return 801; }
static inline hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t, hipGraphNode_t*, char*, size_t) { if (e) *e = nullptr; return 801; }
static inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return 801; }
static inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
static inline hipError_t hipGraphGetNodes(hipGraph_t, hipGraphNode_t*, size_t* n) { if (n) *n = 0; return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t_ms - a->t_ms); return hipSuccess; }
template <class K> static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, K, int, size_t) { *n = 2; return hipSuccess; }

template <class K, class... A>
static inline void simt_launch(const char* name, K kernel, dim3 grid, dim3 block, size_t, hipStream_t, A... args) {
  if (grid.y != 1 || grid.z != 1 || block.y != 1 || block.z != 1) simt::die("only one-dimensional launches are modelled");
  simt::launch(name, kernel, grid.x, block.x, args...);
}
#define hipLaunchKernelGGL(kern, ...) simt_launch(#kern, kern, __VA_ARGS__)
