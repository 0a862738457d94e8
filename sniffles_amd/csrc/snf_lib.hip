// snf_lib.hip - C-ABI implementation (include/sniffles_amd.h): batch state in HBM, launch sequence,
// per-kernel HIP-event timing.  Built for gfx950 with
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared
// (-ffp-contract=off: the reference's double arithmetic has no fused multiply-add).
#include "snf_stage_final.h"
#include "snf_wave_refine.h"
#include "snf_wave_cons.h"
#include "snf_fused.h"
#include "snf_stage_window.h"
#include "snf_stage_out.h"
#include "snf_wave_call.h"
#include "snf_wave_call_g.h"
#include "snf_wave_refine_g.h"
#include "snf_ctx.h"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <fcntl.h>
#include <sys/file.h>
#include <unistd.h>

#include <atomic>
#include <map>
#include <tuple>
#include <cmath>
#include <ctime>
#include <memory>
#include <mutex>
#include <thread>

using namespace snf;

// the three instances of the workgroup consensus kernel (snf_wave_cons.h): <class, table slots, sampled positions, others,
// waves/SIMD it is compiled for, waves per call, vote columns in LDS, staged-read bytes, escape entries>
#define K_CONS_SMALL(MINW) e45w_consensus<1, 256, 128, 64, MINW, 4, SNF_CONS_SMALL_L, 448, 96>
#define K_CONS_SMALL_1W e45w_consensus<1, 256, 128, 64, 5, 1, SNF_CONS_SMALL_L, 448, 96>
#define K_CONS_LARGE e45w_consensus<2, 1024, 512, 256, 2, 4, SNF_CONS_LARGE_L, 0, 512>
#define K_CONS_LARGE_8W e45w_consensus<2, 1024, 512, 256, 2, 8, SNF_CONS_LARGE_L, 0, 512>
#define K_CONS_LARGE_16W e45w_consensus<2, 1024, 512, 256, 4, 16, SNF_CONS_LARGE_L, 0, 512>
#define K_CONS_ROWS e45w_consensus<4, 1024, 512, 512, 3>

// ---------------------------------------------------------------------------------------------- kernels
SNF_KERNEL(a1_keys, View)
SNF_KERNEL(a0_keep, View)
SNF_KERNEL(a0_compact, View)
SNF_KERNEL(a2_heads, View)
SNF_KERNEL(a3_bins, View)
SNF_KERNEL(a4_binstats, View)
SNF_KERNEL(a5_leadflags, View)
SNF_KERNEL(a6_scatter, View)
SNF_KERNEL(a7_seeds, View)
SNF_KERNEL(b1_seedmetrics, View)
SNF_KERNEL(b2_runs, View)
SNF_KERNEL(c1_mergeruns, View)
SNF_KERNEL(c2_validate, View)
SNF_KERNEL(c3_serial, View)
SNF_KERNEL(c4_clusters, View)
SNF_KERNEL(d1_refine, View)
SNF_KERNEL(d1b_rctable, View)
SNF_KERNEL(d2_call, View)
SNF_KERNEL(d3_compact, View)
SNF_KERNEL(d3_taskoff, View)
SNF_KERNEL(d3_svid, View)
SNF_KERNEL(d3_rnames, View)
SNF_KERNEL(d4_coverage, View)
SNF_KERNEL(d5_covsum, View)
SNF_KERNEL(d5_covavg, View)
SNF_KERNEL(e1_finalize, View)
SNF_KERNEL(e2_best, View)
SNF_KERNEL(e3_conslist, View)
SNF_KERNEL(e4_anchor, View)
SNF_KERNEL(e5_align, View)
SNF_KERNEL(e6_vote, View)
SNF_KERNEL(z1_results, View)
SNF_KERNEL(d3l_rnames_late, View)
SNF_KERNEL(f1_flags, View)
SNF_KERNEL(f2_scan, View)
SNF_KERNEL(f3_rank, View)
SNF_KERNEL(f4_emit, View)
SNF_KERNEL(s1_blockcov, BlockCov)
SNF_KERNEL(s2_covends, CovCalls)
SNF_KERNEL(s2_covcalls, CovCalls)
SNF_KERNEL(d5x_covexact, CovExact)
SNF_KERNEL(r3_maxdepth, MaxDepth)

// read preparation (coverage rank structures + REF haplotype prefix counts)
struct ReadPrep {
  const int32_t* r_end; const uint8_t* r_hp; const int32_t* r_task; const int32_t* r_start;
  uint64_t* rk_in; uint32_t* rv_in; const uint64_t* rk_out; const uint32_t* rv_out;
  int32_t* re_sorted; int32_t* re_top; uint64_t *fs2, *fe2; int64_t R;  // fs2/fe2: (hp == 1) << 32 | (hp == 2) in start / end order
  int32_t *rs_mid, *re_mid;  // every 16th start / sorted end (View::rs_mid)
  const uint64_t* t_base;  // [T+1] prefix of (max read end + 1): key = t_base[task] + end orders by (task, end)
  int key32;               // the key space fits 32 bits: rk_in / rk_out hold uint32_t keys
};
namespace snf {
SNF_HD void r1_endkeys_body(int64_t r, const ReadPrep& p) {
  const uint64_t k = p.t_base[p.r_task[r]] + (uint64_t)(uint32_t)p.r_end[r];
  if (p.key32) ((uint32_t*)p.rk_in)[r] = (uint32_t)k; else p.rk_in[r] = k;
  p.rv_in[r] = p.r_hp[r];
  p.fs2[r] = p.r_hp[r] == 1 ? (1ull << 32) : (p.r_hp[r] == 2 ? 1ull : 0ull);
  if ((r & 15) == 0) p.rs_mid[r >> 4] = p.r_start[r];
  if (r == 0) p.fs2[p.R] = 0;
}
SNF_HD void r2_unpack_body(int64_t r, const ReadPrep& p) {
  // the sort keeps every task's reads in that task's slots, so r_task[r] is also the task of sorted position r
  const uint64_t k = p.key32 ? (uint64_t)((const uint32_t*)p.rk_out)[r] : p.rk_out[r];
  p.re_sorted[r] = (int32_t)(k - p.t_base[p.r_task[r]]);
  if ((r & ((1 << SNF_TOP_SHIFT) - 1)) == 0) p.re_top[r >> SNF_TOP_SHIFT] = p.re_sorted[r];
  if ((r & 15) == 0) p.re_mid[r >> 4] = p.re_sorted[r];
  p.fe2[r] = p.rv_out[r] == 1u ? (1ull << 32) : (p.rv_out[r] == 2u ? 1ull : 0ull);
  if (r == 0) p.fe2[p.R] = 0;
}
}  // namespace snf
SNF_KERNEL(r1_endkeys, ReadPrep)
SNF_KERNEL(r2_unpack, ReadPrep)

// ---------------------------------------------------------------------------------------------- batch
namespace {

thread_local std::string g_err;

struct DevBuf { void* p; size_t bytes; bool slab = false; };

// Slabs of destroyed batches are kept for the next batch of the process (a pipeline runs one task after the other, each with
// a slab of up to a few GB: hipMalloc / hipFree of that size costs up to hundreds of milliseconds on some boxes, far more
// than the upload itself).  At most SNF_SLAB_CACHE_MAX slabs are kept; the smallest one that is large enough is reused.
#define SNF_SLAB_CACHE_MAX 6   /* per device */
struct SlabCache {
  struct E { void* p; size_t bytes; int device; };
  std::mutex mu; std::vector<E> free_list;
  // every cached slab of `device` (< 0: of all devices) goes back to the driver; returns the bytes released
  size_t trim(int device) {
    std::vector<E> gone;
    {
      std::lock_guard<std::mutex> g(mu);
      for (size_t i = 0; i < free_list.size();)
        if (device < 0 || free_list[i].device == device) { gone.push_back(free_list[i]); free_list.erase(free_list.begin() + (long)i); }
        else i++;
    }
    size_t bytes = 0;
    for (auto& e : gone) {
      int cur = 0; (void)hipGetDevice(&cur);
      (void)hipSetDevice(e.device); (void)hipFree(e.p); (void)hipSetDevice(cur);
      bytes += e.bytes;
    }
    return bytes;
  }
  void* take(int device, size_t bytes, size_t* got) {
    std::lock_guard<std::mutex> g(mu);
    int best = -1;
    for (size_t i = 0; i < free_list.size(); i++)
      if (free_list[i].device == device && free_list[i].bytes >= bytes && (best < 0 || free_list[i].bytes < free_list[(size_t)best].bytes)) best = (int)i;
    if (best < 0) return nullptr;
    void* p = free_list[(size_t)best].p; *got = free_list[(size_t)best].bytes;
    free_list.erase(free_list.begin() + best);
    return p;
  }
  bool give(int device, void* p, size_t bytes) {   // false: the cache is full, the caller frees the slab
    std::lock_guard<std::mutex> g(mu);
    size_t mine = 0;
    for (auto& e : free_list) mine += e.device == device;
    if (mine >= SNF_SLAB_CACHE_MAX) return false;
    free_list.push_back({p, bytes, device});
    return true;
  }
};
SlabCache g_slabs;

struct Timing { const char* name; float ms; int64_t bytes; int launches; };   // (timing_acc: ms = sum, launches = passes)

// ---- pinned, grow-only host buffers for results (pageable D2H is 3-4x slower) ----
// Pinning memory costs more than filling it, and a pipeline creates one batch per task: released buffers go to a small
// process-wide cache and the next batch takes the smallest one that is large enough.
struct PinnedCache {
  struct E { void* p; size_t cap; };
  std::mutex mu; std::vector<E> free_list;
  size_t trim() {
    std::vector<E> gone;
    { std::lock_guard<std::mutex> g(mu); gone.swap(free_list); }
    size_t bytes = 0;
    for (auto& e : gone) {
      (void)hipHostFree(e.p);
      bytes += e.cap;
    }
    return bytes;
  }
  void* take(size_t bytes, size_t* cap) {
    std::lock_guard<std::mutex> g(mu);
    int best = -1;
    for (size_t i = 0; i < free_list.size(); i++)
      if (free_list[i].cap >= bytes && (best < 0 || free_list[i].cap < free_list[(size_t)best].cap)) best = (int)i;
    if (best < 0) return nullptr;
    void* p = free_list[(size_t)best].p; *cap = free_list[(size_t)best].cap;
    free_list.erase(free_list.begin() + best);
    return p;
  }
  bool give(void* p, size_t cap) {
    std::lock_guard<std::mutex> g(mu);
    if (free_list.size() >= 16) return false;
    free_list.push_back({p, cap});
    return true;
  }
};
PinnedCache g_pinned;

struct HostBuf {
  void* p = nullptr; size_t cap = 0;
  bool external = false;      // caller's memory (snf_batch_set_result_memory): registered with the device here, never freed here
  // `registered`: the caller's ranges this batch has page-locked so far (switching between them - one segment per pass in
  // flight - must not pin again; they are unpinned when the batch goes)
  void adopt(void* mem, size_t bytes, std::vector<std::pair<void*, size_t>>& registered) {
    release();
    if (!mem || !bytes) return;
    bool known = false;
    for (auto it = registered.begin(); it != registered.end();) {
      if (it->first == mem && it->second >= bytes) { known = true; ++it; }
      else if (it->first == mem) { (void)hipHostUnregister(mem); it = registered.erase(it); }     // the same base with a larger size: pinned anew below
      else ++it;
    }
    if (!known) {
      SNF_HIP(hipHostRegister(mem, bytes, hipHostRegisterDefault));
      void* dp = nullptr;
      if (hipHostGetDevicePointer(&dp, mem, 0) != hipSuccess || dp != mem) {     // (the kernels and the host use one address)
        (void)hipGetLastError(); (void)hipHostUnregister(mem);
        fail("snf_batch_set_result_memory: the device maps this memory at another address");
      }
      registered.push_back({mem, bytes});
    }
    p = mem; cap = bytes; external = true;
  }
  void* ensure(size_t bytes) {
    // SNF_STAGE_ARENA_MB: the arena is never smaller than this.  A process that serves batches of unknown size (sniffles_amd/server.py)
    // reserves for the largest it expects once, at its first upload, instead of growing when that batch arrives: pinning ~0.5 GB takes
    // ~0.15 s, during which every other thread of the process that faults a page waits as well
    static const size_t floor_b = getenv("SNF_STAGE_ARENA_MB") ? (size_t)atoll(getenv("SNF_STAGE_ARENA_MB")) << 20 : 0;
    if (bytes < floor_b) bytes = floor_b;
    if (bytes <= cap && p) return p;
    if (external) fail("the result does not fit the memory given to snf_batch_set_result_memory (" + std::to_string(bytes) + " bytes needed, " + std::to_string(cap) + " given)");
    release();
    size_t got = 0;
    void* c = g_pinned.take(bytes, &got);
    if (c && got <= 4 * bytes + ((size_t)64 << 20)) { p = c; cap = got; return p; }
    if (c) g_pinned.give(c, got);
    cap = bytes + bytes / 4 + 4096;
    if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) {   // idle pinned buffers of finished batches go first, then once more
      (void)hipGetLastError();
      p = nullptr;
      g_pinned.trim();
      SNF_HIP(hipHostMalloc(&p, cap, hipHostMallocDefault));
    }
    return p;
  }
  void release() {
    if (!p) return;
    if (external) external = false;          // (unpinned with the batch: snf_batch_impl::ext_ranges)
    else if (!g_pinned.give(p, cap)) {
      (void)hipHostFree(p);
    }
    p = nullptr; cap = 0;
  }
};

struct snf_batch_impl {
  snf_config_t cfg;
  int device = 0;
  hipStream_t stream = nullptr;   // main stream (also the one fetch/sync wait on)
  hipStream_t stream2 = nullptr;  // side stream: read preparation, finalize scalar kernels
  hipStream_t stream3 = nullptr;  // third stream: the SMALL consensus class and the verbatim copies next to the LARGE class
  bool stream3_high = false;      // ... created with the device's highest stream priority
  hipStream_t stream4 = nullptr;  // fourth stream: sv ids + supporting read names and their D2H copy (off the coverage / QC chain)
  hipStream_t cur = nullptr;      // stream the LAUNCH / prim_* helpers enqueue on
  int cur_slot = 0;
  bool fused = false;             // flag -> scan -> emit chains as fused kernel pairs (snf_fused.h); off: rocPRIM scans
  bool timing = true;             // HIP events around the heavy kernels (snf_batch_set_timing)
  bool time_all = false;          // SNF_TIME_ALL=1: HIP events around every launch, not only the heavy kernels
  int time_every = 8;             // the event brackets are recorded on every n-th pass of the handle (1: every pass; snf_batch_timing_every):
  bool time_now = true;           //   two records per launch keep the streams from running launches back to back - measured 80 us of a
  uint64_t pass_count = 0;        //   1.29-ms step with two batches in flight when every pass carried them
  bool timeline = false;          // SNF_TIMELINE=1: print (offset, duration) of every bracketed op of the step to stderr
  bool res_current = false;       // z1_results has run after the last kernel that changes what it publishes
  int64_t* h_rn_total = nullptr;  // pinned (hb_res): see View::res_rn_total
  int sched_readprep = 1;         // SNF_READPREP: 0 first, 1 enqueued behind d1w (may start at once), 2 after d3_taskoff, 3 starts with d1w
  void (*k_d2w)(const View, int64_t) = nullptr; void (*k_e1w)(const View, int64_t) = nullptr;  // occupancy variants
  int slots_d1w = 8192, slots_d2w = 8192, slots_e1w = 8192, slots_big = 8192;
  int slots_cons_s = 1 << 22, slots_cons_l = 1 << 22, slots_cons_s1 = 65536;   // grid caps of the SMALL / LARGE consensus kernels
  bool d1_groups = true;          // merge_inner / resplit: small merged clusters eight per wave (snf_wave_refine_g.h); SNF_NO_D1_GROUPS=1: a wave per cluster
  bool d2_groups = true;          // call_from: small refined clusters several per wave (snf_wave_call_g.h); SNF_NO_D2_GROUPS=1: a wave per cluster
  int cons_nw = 1;                // waves per SMALL consensus call: 1 = one wave per call (default: single-wave workgroups leave room for the
                                  // LARGE class next to them - LARGE in place 0.75 -> 0.5 ms, the pass 2.5 % shorter), SNF_CONS_NW=4: four
  int cons_large_nw = 4;          // SNF_CONS_LARGE_NW: waves per LARGE consensus call (4, 8, 16)
  int occ_s = 5;                  // SNF_OCC_S: waves/SIMD the SMALL consensus kernel is compiled for (5, 6, 8)
  int read_key_bits = 64;         // significant bits of the read-end sort key
  std::vector<int32_t> h_rend_max; // per task: largest read end (filled by the upload's validation pass)
  bool uploaded = false;
  bool finalized = false;         // run_finalize has run on the current candidates
  int finalize_runs = 0;          // finalize calls since the last call_candidates
  int rn_state = 0;               // supporting read names of the current candidates: 0 all written, 1 deferred (sizes only), 2 written for the kept calls
  int64_t pf_words = 0;           // prefilter bitmap size (uint32 words)
  // a whole pass (call_candidates + finalize) as ONE HIP graph per result configuration (snf_batch_pass): captured on the second
  // pass of a configuration - every launch size is known from the first -, replayed afterwards
  struct PassGraph { hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; int passes = 0; int rn_state = 0; int rn_defer = 0; size_t cap_out = 0, cap_alt = 0; };
  std::map<std::tuple<void*, void*, int>, PassGraph> graphs;   // key: (result block, ALT block, output mode)
  int graph_mode = 0;             // set at upload: 2 = replay (small batches, or SNF_GRAPH=1), 0 = eager, 1 = replay only while no other batch of
                                  // this process has a pass in flight
  bool in_flight = false;         // counted in its device's DevicePacing::passes_in_flight
  int slot_fd = -1;               // the GPU slot this pass holds (SNF_GPU_SLOTS)
  double pass_start_ms = 0.0;     // when this handle's pass in flight began (pacing of overlapping passes)
  bool capturing = false;         // run_pass is capturing this pass into a graph
  int chain_slot = -1;            // SNF_CHAIN_GATE: the device's event this pass records at the end of its chain
  // Result staged through HBM (run_finalize): with another pass in flight on the device the kernels of a pass store the result block
  // and the ALT section into HBM; they are taken to the pinned buffers by two copies at the fetch (default) or by two small copy
  // kernels behind their producers (SNF_STAGE_COPY=kernel: z2_stage_copy reads the sizes on the device)
  bool staged = false; bool staged_kernel = false, stage_block_copied = false, stage_alt_copied = false;
  bool graph_failed = false;      // a capture / instantiate error: eager from then on (reported once with SNF_PROF)
  bool w4_split = false;          // SNF_W4_SPLIT=1 sets it (read when the batch is opened)
  int64_t h_n_big64 = 0;          // windows of more than 64 leads (counted at upload): upper bound of the blocks the large instance of w4s_segment gets
  int64_t h_n_occ = 0; int win_cap = 0;   // window front end: occupied windows (a property of the input, counted at upload), instance of w4 / w6
  bool reads_ready = false;       // the read index (sorted ends, hap prefix counts) of the uploaded tasks exists
  bool cov_avg_ready = false;     // a call_candidates pass has formed coverage.mean() per task
  bool readprep_each_pass = false; // SNF_READPREP_EACH_PASS=1: rebuild it in every call_candidates (round-1 behaviour)
  int run_gap = 1000;
  // host side of the inputs: the caller's arrays are BORROWED from snf_batch_add_task until snf_batch_upload returns
  // (validated, staged into pinned memory and copied by the upload); only the scalars are kept afterwards
  std::vector<snf_task_input_t> tasks;
  std::vector<uint8_t> task_on_device;   // 1: the task's array pointers are HBM of this device (snf_batch_add_task_device)
  std::vector<int64_t> h_lead_off{0}, h_read_off{0}, h_tr_off{0}, h_pool_off{0};
  std::vector<int32_t> h_trs, h_tre, h_trp, h_has_tr;
  std::vector<int32_t> h_nms, h_nme; std::vector<int64_t> h_nm_off{0};   // reference 'N' mask intervals per task
  std::vector<int32_t> h_cov_exact;                                        // per task: coverage means by the exact walk
  int32_t* d_max_depth = nullptr;
  // device
  std::vector<DevBuf> bufs;
  uint8_t* slab = nullptr; size_t slab_cap = 0, slab_used = 0, slab_next = (size_t)64 << 20;  // bump allocator (dalloc)
  View v{};
  ReadPrep rp{};
  Counts* h_cnt = nullptr;        // counters as last read back (lives in the pinned result block hb_res)
  void* sort_tmp[2] = {nullptr, nullptr}; size_t sort_tmp_bytes[2] = {0, 0};  // rocPRIM temp storage per stream
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_fork3 = nullptr, ev_join3 = nullptr, ev_base = nullptr,
             ev_counts = nullptr, ev_rn = nullptr, ev_join4 = nullptr, ev_e3 = nullptr;  // host waits: counters published (main), read-name total published (side)
  // growable finalize scratch
  int64_t tab_cap = 0, aln_cap = 0, cr_cap = 0, alt_cap = 0;
  // results (host)
  HostBuf hb_calls, hb_rn, hb_res;   // stage-0 fetch: candidate records, read names; the pinned result block
  bool pass_idle = false;            // a fetch has waited for the pass and neither stage has been enqueued since (set_result_memory need not wait again)
  HostBuf hb_out, hb_alt;            // stage-1 fetch: the output block (snf_stage_out.h), the ALT section
  std::vector<std::pair<void*, size_t>> ext_ranges;   // caller memory page-locked for them (snf_batch_set_result_memory)
  // class sizes of this handle's previous finalize (same input -> same sizes; first pass: one host wait for them)
  bool have_hist = false; int64_t hist_calls = 0, hist_small = 0, hist_large = 0, hist_copy = 0;
  bool have_big_hist = false; int64_t hist_big[3] = {0, 0, 0};   // items of the previous whole pass for x_big<0 / 1 / 2> (0: that launch is skipped)
  bool skipped_big[3] = {false, false, false};
  int out_mode = 0;                  // enum snf_output
  std::vector<int32_t> r_status; std::vector<int64_t> r_off; std::vector<double> r_cov;
  // snf_batch_fetch_clusters result (host)
  std::vector<int32_t> cl_task, cl_svtype, cl_start, cl_end, cl_seed, cl_seed_index, cl_nlong, cl_lead, cl_lead_svlen; std::vector<uint8_t> cl_repeat;
  std::vector<int64_t> cl_lead_off;
  // timing
  std::vector<Timing> timings;
  std::vector<Timing> timing_acc;   // sums over the passes since snf_batch_timing_mean_reset
  struct Ev { hipEvent_t a, b; const char* name; int64_t bytes; };
  std::vector<Ev> evs; size_t ev_used = 0;
};

// ---- device memory ----
// dalloc_own: one hipMalloc per buffer (growable scratch that is freed and re-allocated: dfree_one).
// dalloc: carved out of large slabs - a batch has ~140 arrays that live as long as the batch, and a hipMalloc costs
// far more than the bump of a pointer (the upload of a task is on the wall clock of the drop-in path).
template <class T>
T* dalloc_own(snf_batch_impl* b, size_t n) {
  size_t bytes = (n ? n : 1) * sizeof(T);
  void* p = nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess) {
    // out of device memory while idle slabs of finished batches sit in this process's own cache: hand them back and try
    // once more (several ranks / threads on one GPU, torch or RCCL next door, a large contig after many small ones)
    (void)hipGetLastError();
    p = nullptr;
    g_slabs.trim(b->device);
    SNF_HIP(hipMalloc(&p, bytes));
  }
  b->bufs.push_back({p, bytes});
  return (T*)p;
}
template <class T>
T* dalloc(snf_batch_impl* b, size_t n) {
  const size_t bytes = (((n ? n : 1) * sizeof(T)) + 255) & ~(size_t)255;
  if (b->slab_used + bytes > b->slab_cap) {
    size_t cap = b->slab_next; if (cap < bytes) cap = bytes;
    size_t got = 0;
    void* cached = getenv("SNF_NO_SLAB_CACHE") ? nullptr : g_slabs.take(b->device, cap, &got);
    if (cached && got <= 2 * cap + ((size_t)256 << 20)) {   // (a much larger slab is left for a batch that needs it)
      b->bufs.push_back({cached, got, true});
      b->slab = (uint8_t*)cached; cap = got;
    } else {
      if (cached) g_slabs.give(b->device, cached, got);
      b->slab = dalloc_own<uint8_t>(b, cap);
      b->bufs.back().slab = true;
    }
    b->slab_cap = cap; b->slab_used = 0;
    b->slab_next = (size_t)256 << 20;   // a batch that outgrows its first slab continues in small ones
  }
  T* p = (T*)(b->slab + b->slab_used);
  b->slab_used += bytes;
  return p;
}
void dfree_all(snf_batch_impl* b) {
  for (auto& d : b->bufs) {
    if (d.slab && !getenv("SNF_NO_SLAB_CACHE") && g_slabs.give(b->device, d.p, d.bytes)) continue;   // kept for the next batch
    (void)hipFree(d.p);
  }
  b->bufs.clear();
}
void dfree_one(snf_batch_impl* b, void* p) {
  if (!p) return;
  for (size_t i = 0; i < b->bufs.size(); i++)
    if (b->bufs[i].p == p) {
      (void)hipFree(p);
      b->bufs.erase(b->bufs.begin() + i);
      return;
    }
}
void h2d(snf_batch_impl* b, void* dst, const void* src, size_t bytes) {
  if (!bytes) return;
  SNF_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, b->cur));
}
void d2h(snf_batch_impl* b, void* dst, const void* src, size_t bytes) {
  if (!bytes) return;
  SNF_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, b->cur));
}
void d2d(snf_batch_impl* b, void* dst, const void* src, size_t bytes) {
  if (!bytes) return;
  SNF_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, b->cur));
}
void dzero(snf_batch_impl* b, void* p, size_t bytes, int val = 0) {
  if (!bytes) return;
  SNF_HIP(hipMemsetAsync(p, val, bytes, b->cur));
}
void dsync(snf_batch_impl* b) {
  SNF_HIP(hipStreamSynchronize(b->cur));
}
// enqueue on the side stream for the lifetime of this object; fork()/join() order it against the main stream
void fork_mark(snf_batch_impl* b) {  // point on the main stream the side stream may start after
  SNF_HIP(hipEventRecord(b->ev_fork, b->stream));
}
struct SideStream {
  snf_batch_impl* b;
  SideStream(snf_batch_impl* b_) : b(b_) {
    SNF_HIP(hipStreamWaitEvent(b->stream2, b->ev_fork, 0));
    b->cur = b->stream2; b->cur_slot = 1;
  }
  ~SideStream() {
    (void)hipEventRecord(b->ev_join, b->stream2);
    b->cur = b->stream; b->cur_slot = 0;
  }
};
void join_side(snf_batch_impl* b) {  // main stream waits for everything enqueued on the side stream so far
  SNF_HIP(hipStreamWaitEvent(b->stream, b->ev_join, 0));
}
template <class T>
T* upload_vec(snf_batch_impl* b, const std::vector<T>& h, size_t extra = 0) {
  T* d = dalloc<T>(b, h.size() + extra);
  h2d(b, d, h.data(), h.size() * sizeof(T));
  return d;
}

// ---- timing ----
struct Scope {
  snf_batch_impl* b;
  size_t idx = (size_t)-1;
  // always: bracketed on every pass (the kernel bench.py states the roofline on); the others only on the handle's sampled passes
  Scope(snf_batch_impl* b_, const char* name, int64_t bytes, bool always = false) : b(b_) {
    if (!b->timing || !(always || b->time_now)) return;
    if (b->ev_used == b->evs.size()) {
      snf_batch_impl::Ev e{};
      SNF_HIP(hipEventCreate(&e.a)); SNF_HIP(hipEventCreate(&e.b));
      b->evs.push_back(e);
    }
    idx = b->ev_used++;
    b->evs[idx].name = name; b->evs[idx].bytes = bytes;
    SNF_HIP(hipEventRecord(b->evs[idx].a, b->cur));
  }
  ~Scope() {
    if (idx != (size_t)-1) (void)hipEventRecord(b->evs[idx].b, b->cur);
  }
};

void d2h_timed(snf_batch_impl* b, void* dst, const void* src, size_t bytes, const char* name) {
  if (!bytes) return;
  if (b->time_all) { Scope _s(b, name, (int64_t)bytes); d2h(b, dst, src, bytes); }
  else d2h(b, dst, src, bytes);
}

// LAUNCH_Q: tiny kernels are only bracketed by events when SNF_TIME_ALL=1 (two event records cost more host time
// than the launch itself and the stage A-C region is launch-bound)
#define LAUNCH_Q(kern, view, n, bytes)                                                        \
  do {                                                                                        \
    int64_t _n = (n);                                                                         \
    if (_n > 0) {                                                                             \
      if (b->time_all) { Scope _s(b, #kern, (bytes));                                         \
        hipLaunchKernelGGL(kern, dim3((unsigned)((_n + 255) / 256)), dim3(256), 0, b->cur, view, _n); } \
      else hipLaunchKernelGGL(kern, dim3((unsigned)((_n + 255) / 256)), dim3(256), 0, b->cur, view, _n); \
      SNF_HIP(hipGetLastError());                                                             \
    }                                                                                         \
  } while (0)
#define LAUNCH(kern, view, n, bytes)                                                          \
  do {                                                                                        \
    int64_t _n = (n);                                                                         \
    if (_n > 0) {                                                                             \
      Scope _s(b, #kern, (bytes));                                                            \
      hipLaunchKernelGGL(kern, dim3((unsigned)((_n + 255) / 256)), dim3(256), 0, b->cur, view, _n); \
      SNF_HIP(hipGetLastError());                                                             \
    }                                                                                         \
  } while (0)

// kernels of snf_fused.h: n elements, 256 per block; bracketed by timing events only with SNF_TIME_ALL
#define FUSED(kern, n)                                                                                  \
  do {                                                                                                  \
    int64_t _n = (n);                                                                                   \
    if (_n > 0) {                                                                                       \
      if (b->time_all) { Scope _s(b, #kern, 0);                                                         \
        hipLaunchKernelGGL(kern, dim3((unsigned)((_n + 255) / 256)), dim3(256), 0, b->cur, v, _n); }    \
      else hipLaunchKernelGGL(kern, dim3((unsigned)((_n + 255) / 256)), dim3(256), 0, b->cur, v, _n);   \
      SNF_HIP(hipGetLastError());                                                                       \
    }                                                                                                   \
  } while (0)

// single-launch scan chains (snf_fused.h chain_scan)
#define CHAIN(kern, n) FUSED(kern, n)

// ---- primitives: stable radix sort (key,value) and exclusive scans ----
template <class K>
void prim_sort_pairs(snf_batch_impl* b, const K* kin, K* kout, const uint32_t* vin, uint32_t* vout, int64_t n,
                     int end_bit, const char* name) {
  if (n <= 0) return;
  size_t need = 0;
  // (rocPRIM's default takes a block sort + ~20 merge launches up to 1 M items: 140 us for the 0.7 M pairs behind the prefilter)
  using SortCfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 65536>;
  SNF_HIP(rocprim::radix_sort_pairs<SortCfg>(nullptr, need, kin, kout, vin, vout, (size_t)n, 0, end_bit, b->cur));
  void*& tmp = b->sort_tmp[b->cur_slot]; size_t& tmpb = b->sort_tmp_bytes[b->cur_slot];
  if (need > tmpb) { tmp = dalloc<uint8_t>(b, need); tmpb = need; }      // (from the batch's slab: a hipMalloc + hipFree per task cost more than a contig's upload copies; an outgrown region stays in the slab)
  Scope s(b, name, n * 2 * (int64_t)(sizeof(K) + 4));
  SNF_HIP(rocprim::radix_sort_pairs<SortCfg>(tmp, need, kin, kout, vin, vout, (size_t)n, 0, end_bit, b->cur));
}
template <class T>
void prim_exscan(snf_batch_impl* b, const T* in, T* out, int64_t n, const char* name) {
  if (n <= 0) return;
  size_t need = 0;
  SNF_HIP(rocprim::exclusive_scan(nullptr, need, in, out, (T)0, (size_t)n, rocprim::plus<T>(), b->cur));
  void*& tmp = b->sort_tmp[b->cur_slot]; size_t& tmpb = b->sort_tmp_bytes[b->cur_slot];
  if (need > tmpb) { tmp = dalloc<uint8_t>(b, need); tmpb = need; }
  if (b->time_all) { Scope s(b, name, n * 2 * (int64_t)sizeof(T));
    SNF_HIP(rocprim::exclusive_scan(tmp, need, in, out, (T)0, (size_t)n, rocprim::plus<T>(), b->cur)); }
  else SNF_HIP(rocprim::exclusive_scan(tmp, need, in, out, (T)0, (size_t)n, rocprim::plus<T>(), b->cur));
}

// ---- genotype table: exactly genotyping.py:124-171 for every (normalised support, coverage) ----
double likelihood_ratio(double q1, double q2) {
  if (q1 / q2 > 0) return std::log(q1 / q2) / std::log(10.0);  // math.log(x, 10)
  return 0;
}
// (the table depends on two scalars of the configuration only and costs ~0.4 M libm calls - 2 ms, two thirds of a contig task's upload:
//  built once per process and configuration)
std::vector<GtEntry> build_gt_lut_uncached(const snf_config_t& cfg);
const std::vector<GtEntry>& build_gt_lut(const snf_config_t& cfg) {
  static std::mutex mu;
  static std::map<std::pair<uint64_t, int>, std::vector<GtEntry>> known;
  uint64_t bits; memcpy(&bits, &cfg.genotype_error, sizeof(bits));
  std::lock_guard<std::mutex> g(mu);
  auto it = known.find({bits, (int)cfg.genotype_ploidy});
  if (it == known.end()) it = known.emplace(std::make_pair(bits, (int)cfg.genotype_ploidy), build_gt_lut_uncached(cfg)).first;
  return it->second;      // (entries are never erased: the reference stays valid)
}
std::vector<GtEntry> build_gt_lut_uncached(const snf_config_t& cfg) {
  std::vector<GtEntry> lut((size_t)SNF_GT_N * SNF_GT_N);
  double p[3] = {cfg.genotype_error, 1.0 / (double)cfg.genotype_ploidy, 1.0 - cfg.genotype_error};
  for (int ns = 0; ns < SNF_GT_N; ns++)
    for (int ncv = 0; ncv < SNF_GT_N; ncv++) {
      GtEntry e{0, 0, 0, 0};
      if (ncv >= ns) {
        double q[3]; int order[3] = {0, 1, 2};
        for (int g = 0; g < 3; g++) q[g] = std::pow(p[g], (double)ns) * std::pow(1.0 - p[g], (double)(ncv - ns));
        for (int a = 1; a < 3; a++) {  // list.sort(key=q, reverse=True) is stable
          int o = order[a], bb = a - 1;
          while (bb >= 0 && q[order[bb]] < q[o]) { order[bb + 1] = order[bb]; bb--; }
          order[bb + 1] = o;
        }
        double sum = 0; for (int g = 0; g < 3; g++) sum += q[order[g]];
        double nq[3]; for (int g = 0; g < 3; g++) nq[g] = q[order[g]] / sum;
        double qz = 0; for (int g = 0; g < 3; g++) if (order[g] == 0) { qz = nq[g]; break; }
        long z = (long)((-10.0) * likelihood_ratio(qz, nq[0])); if (z > 60) z = 60;
        long gq = (long)((-10.0) * likelihood_ratio(nq[1], nq[0])); if (gq > 60) gq = 60;
        e.order0 = (int8_t)order[0]; e.gq = (int8_t)gq; e.z = (int8_t)z;
      }
      lut[(size_t)ns * SNF_GT_N + ncv] = e;
    }
  return lut;
}

int bits_for(uint64_t x) { int b = 0; while (x) { b++; x >>= 1; } return b ? b : 1; }

// ---------------------------------------------------------------------------------------------- upload
// Host -> HBM in three steps: (1) host threads validate every task and copy its columns into ONE pinned staging arena,
// at their final place in the batch-wide columns (task after task); (2) two large copies put the arena and the sequence
// pool into HBM; (3) kernels derive what the pipeline wants besides the columns: the interleaved per-lead records
// (a6_scatter gathers one 64-B record instead of 20 scattered words), the task of every lead / read.
// The staging arena is process-wide and grow-only: pinning memory costs more than the copy, so it is paid once.
struct StageArena {
  std::mutex mu; void* p = nullptr; size_t cap = 0;
  void* ensure(size_t bytes) {
    if (bytes <= cap && p) return p;
    if (p) (void)hipHostFree(p);
    cap = bytes + bytes / 8 + (1u << 20);
    SNF_HIP(hipHostMalloc(&p, cap, hipHostMallocDefault));
    return p;
  }
};
StageArena g_stage;

enum { IC_REF_START = 0, IC_REF_END, IC_QRY_START, IC_QRY_END, IC_SVLEN, IC_READ_LEN, IC_QNAME, IC_READ_ID, IC_PS, IC_MATE_CONTIG,
       IC_MATE_POS, IC_SEQ_LEN, IC_SEQ_OFF, IC_NM, IC_SVTYPE, IC_STRAND, IC_MAPQ, IC_SOURCE, IC_HAP, IC_IS_SA, IC_FIRST, IC_REV,
       IC_RSTART, IC_REND, IC_RHP, IC_COUNT };
const int kInElem[IC_COUNT] = {4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 8, 8, 1, 1, 1, 1, 1, 1, 1, 1, 4, 4, 1};

struct PackView {   // kernels of the upload
  const int64_t* t_lead_off; const int64_t* t_read_off; int32_t T; int64_t N, R;
  const int32_t *ref_start, *ref_end, *qry_start, *qry_end, *svlen, *read_len, *ps, *mate_contig, *mate_pos, *seq_len;
  const uint32_t *qname, *read_id; const int64_t* seq_off;
  const uint8_t *svtype, *strand, *mapq, *source, *hap, *is_sa, *first, *rev;
  LeadRec* rec; int32_t* lead_task; int32_t* r_task;
};
}  // namespace
namespace snf {
SNF_HD int32_t task_of(const int64_t* off, int32_t T, int64_t i) {   // last t with off[t] <= i (empty tasks skipped)
  int32_t lo = 0, hi = T - 1;
  while (lo < hi) { const int32_t mid = (lo + hi + 1) >> 1; if (off[mid] <= i) lo = mid; else hi = mid - 1; }
  return lo;
}
SNF_HD void u1_pack_body(int64_t i, const PackView& q) {
  LeadRec r;
  r.ref_start = q.ref_start[i]; r.ref_end = q.ref_end[i]; r.qry_start = q.qry_start[i]; r.qry_end = q.qry_end[i];
  r.svlen = q.svlen[i];
  const bool hs = q.seq_len[i] >= 0;
  r.seq_len = hs ? q.seq_len[i] : -1; r.seq_off = hs ? q.seq_off[i] : 0;
  r.qname = q.qname[i]; r.read_id = q.read_id[i]; r.ps = q.ps[i]; r.mate_pos = q.mate_pos[i];
  r.mate_contig = q.mate_contig[i]; r.read_len = q.read_len[i]; r.orig = (uint32_t)i;
  r.strand = q.strand[i]; r.mapq = q.mapq[i]; r.source = q.source[i]; r.hap = q.hap[i];
  r.is_sa = q.is_sa[i]; r.first = q.first[i]; r.rev = q.rev[i]; r.svtype = q.svtype[i];
  q.rec[i] = r;
  q.lead_task[i] = task_of(q.t_lead_off, q.T, i);
}
SNF_HD void u2_readtask_body(int64_t r, const PackView& q) { q.r_task[r] = task_of(q.t_read_off, q.T, r); }
}  // namespace snf
SNF_KERNEL(u1_pack, PackView)
SNF_KERNEL(u2_readtask, PackView)
struct RebaseView { int64_t* seq_off; const int32_t* seq_len; int64_t base; };
namespace snf {
SNF_HD void u0_rebase_body(int64_t i, const RebaseView& q) { q.seq_off[i] = q.seq_len[i] >= 0 ? q.seq_off[i] + q.base : 0; }
}  // namespace snf
SNF_KERNEL(u0_rebase, RebaseView)

namespace {
double now_ms() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return 1e3 * (double)ts.tv_sec + 1e-6 * (double)ts.tv_nsec; }

// one task's columns -> their slices of the staging arena; every check of the old element-wise add_task happens here
// Staging of one task's input into the pinned arena, in pieces that host threads take independently:
//   stage_leads: rows [i0, i1) of the 22 lead columns (validated, sequence offsets rebased into the batch pool)
//   stage_pool:  bytes [b0, b1) of the task's sequence pool (checked for the consensus gap symbol)
//   stage_reads: the task's read table (sortedness enforced: the rank queries need ascending starts)
void stage_leads(const snf_task_input_t& t, uint8_t* st, const size_t* off, int64_t l0, int64_t p0, int64_t i0, int64_t i1) {
  const void* src[22] = {t.ref_start, t.ref_end, t.qry_start, t.qry_end, t.svlen, t.read_len, t.qname_id, t.read_id, t.ps_rank,
                         t.mate_contig, t.mate_ref_start, t.seq_len, t.seq_off, t.nm, t.svtype, t.strand, t.mapq, t.source, t.hap,
                         t.is_sa, t.bnd_is_first, t.bnd_is_reverse};
  for (int c = 0; c < 22; c++) {
    if (c == IC_SEQ_OFF) continue;
    memcpy(st + off[c] + (size_t)(l0 + i0) * kInElem[c], (const uint8_t*)src[c] + (size_t)i0 * kInElem[c], (size_t)(i1 - i0) * kInElem[c]);
  }
  int64_t* so = (int64_t*)(st + off[IC_SEQ_OFF]) + l0;
  for (int64_t i = i0; i < i1; i++) {
    if (t.svtype[i] >= SNF_NTYPES) fail("svtype code out of range");
    if (t.hap[i] > 2) fail("hap must be 0, 1 or 2 (leadprov.py:403)");
    // the packed per-lead record keeps these in bit fields (snf_view.h LeadRec): anything wider would be cut silently
    if (t.strand[i] > 1 || t.source[i] > 3 || t.is_sa[i] > 1 || t.bnd_is_first[i] > 1 || t.bnd_is_reverse[i] > 1)
      fail("strand / is_sa / bnd_is_first / bnd_is_reverse must be 0 or 1, source one of the four lead sources");
    const int32_t sl = t.seq_len[i];
    if (sl >= 0 && (t.seq_off[i] < 0 || t.seq_off[i] + sl > t.seq_pool_len)) fail("seq_off/seq_len outside seq_pool");
    so[i] = sl >= 0 ? t.seq_off[i] + p0 : 0;      // rebased into the batch pool
  }
}
void stage_pool(const snf_task_input_t& t, uint8_t* st, size_t pool_at, int64_t p0, int64_t b0, int64_t b1, std::atomic<int>* has_dash) {
  // the reference's consensus uses '-' as its gap symbol (consensus.py:317-380): a read base '-' IS a gap there.  BAM sequences
  // cannot hold one; a batch whose pool does (Lead objects built by hand) takes the literal thread kernels e4 / e5 / e6 for every
  // consensus call - they keep the reference's rows with '-' as the gap byte - instead of the LDS-vote kernels (View::cons_thread_only)
  if (memchr(t.seq_pool + b0, '-', (size_t)(b1 - b0))) has_dash->store(1);
  memcpy(st + pool_at + (size_t)(p0 + b0), t.seq_pool + b0, (size_t)(b1 - b0));
}
void stage_reads(const snf_task_input_t& t, uint8_t* st, const size_t* off, int64_t r0, int32_t* rend_max) {
  const int64_t r = t.n_reads;
  // reads: BAM order == ascending start; enforce (stable) so the rank queries are valid
  int32_t* rs = (int32_t*)(st + off[IC_RSTART]) + r0; int32_t* re = (int32_t*)(st + off[IC_REND]) + r0;
  uint8_t* rh = st + off[IC_RHP] + r0;
  bool sorted = true;
  for (int64_t i = 1; i < r; i++) if (t.read_start[i] < t.read_start[i - 1]) { sorted = false; break; }
  if (sorted) {
    if (r) { memcpy(rs, t.read_start, (size_t)r * 4); memcpy(re, t.read_end, (size_t)r * 4); memcpy(rh, t.read_hp, (size_t)r); }
  } else {
    std::vector<int64_t> ord((size_t)r);
    for (int64_t i = 0; i < r; i++) ord[(size_t)i] = i;
    std::stable_sort(ord.begin(), ord.end(), [&](int64_t x, int64_t y) { return t.read_start[x] < t.read_start[y]; });
    for (int64_t i = 0; i < r; i++) { const int64_t k = ord[(size_t)i]; rs[i] = t.read_start[k]; re[i] = t.read_end[k]; rh[i] = t.read_hp[k]; }
  }
  int32_t rmax = 0;
  for (int64_t i = 0; i < r; i++) {
    const int32_t s = rs[i], e = re[i];
    if (s < 0 || s >= t.contig_len || e < s) fail("read interval outside the task region (leadprov.py:497-498)");
    if (rh[i] > 2) fail("read hp must be 0, 1 or 2");
    if (e > rmax) rmax = e;
  }
  *rend_max = rmax;
}

void enqueue_read_index(snf_batch_impl* b);
void enqueue_pass_init(snf_batch_impl* b);
void enqueue_keys(snf_batch_impl* b);

void do_upload(snf_batch_impl* b) {
  SNF_TRACE("snf_batch_upload");
  View& v = b->v;
  const double t_begin = now_ms();
  int T = (int)b->tasks.size();
  int64_t N = b->h_lead_off.back(), R = b->h_read_off.back(), NTR = (int64_t)b->h_trs.size();
  if (T >= (1 << 16)) fail("too many tasks in one batch (max 65535)");
  if (N >= (int64_t)1 << 31 || R >= (int64_t)1 << 31) fail("batch too large for 32-bit lead/read indices");
  v.cfg = b->cfg; v.T = T; v.N = N; v.R = R; v.NTR = NTR; v.run_gap = b->run_gap;
  v.wave_path = getenv("SNF_NO_WAVE") ? 0 : 1;
  v.prof = getenv("SNF_PROF") ? 1 : 0;
  v.merge_reread = (getenv("SNF_MERGE_REREAD") && atoi(getenv("SNF_MERGE_REREAD")) != 0) ? 1 : 0;
  const bool sort64 = getenv("SNF_SORT64") != nullptr;  // tests: force the wide-key sorts
  {  // lead sort key: (task*8 + svtype) << bin_bits | bin, one more bit marks leads outside their contig (sorted last)
    int64_t max_bins = 1;
    for (auto& t : b->tasks) { int64_t nb = t.contig_len / (b->cfg.cluster_binsize > 0 ? b->cfg.cluster_binsize : 1) + 1; if (nb > max_bins) max_bins = nb; }
    v.key_bin_bits = bits_for((uint64_t)max_bins);
    v.key_nbits = v.key_bin_bits + bits_for((uint64_t)(8 * (T > 0 ? T : 1)));
    v.key32 = (!sort64 && v.key_nbits + 1 <= 32) ? 1 : 0;
  }
  v.pool_len = b->h_pool_off.back(); v.pool_cap = 2 * v.pool_len + 16;
  v.pool_extra_base = v.pool_len; v.pool_slice = 0;
  // Fused sequences (merge_inner) go behind the input sequences.  All of them together are at most pool_len bytes, and that
  // much is kept for reservations through the shared counter; in front of it every resident wave of d1w_refine owns a private
  // slice it fills without any atomic (the shared counter is one address: ~10^4 returning atomics per pass queue up in L2 and
  // were a third of that kernel's time).  A wave whose slice is full falls back to the counter.
  if (b->slots_d1w > 0 && getenv("SNF_NO_POOL_SLICES") == nullptr) {
    v.pool_slice = (v.pool_len / b->slots_d1w) & ~(int64_t)15;
    v.pool_extra_base = v.pool_len + v.pool_slice * b->slots_d1w;
    v.pool_cap = v.pool_extra_base + v.pool_len + 16;
  }
  // everything allocated below fits one slab of this size (per-lead arrays ~1.9 KB/lead, the pool twice, the reads);
  // anything beyond it simply opens another slab
  // (+ the ALT pool, the prefilter's columns and bitmap, and the output block sized for half of the leads behind the sort;
  // whatever does not fit opens a second, small slab)
  {
    size_t cells = 0;
    const int bs0 = b->cfg.cluster_binsize > 0 ? b->cfg.cluster_binsize : 1;
    for (auto& t : b->tasks) cells += (size_t)SNF_NTYPES * ((size_t)t.contig_len / bs0 + 1);
    b->slab_next = (size_t)N * (1408 + 104 + 16 + 8) + 2 * (size_t)v.pool_cap + (size_t)R * 96 + cells / 4 + ((size_t)N / 2) * (sizeof(snf_call_t) + 32) + ((size_t)8 << 20);
  }
  v.cnt = dalloc<Counts>(b, 1);
  {  // pinned result block: Counts | call offsets [T+1] | coverage averages [T] | status [T]
    size_t bytes = sizeof(Counts) + 8 + ((size_t)T + 1) * 8 + (size_t)T * 8 + (size_t)T * 4 + 8 + sizeof(OutHdr) + 8;
    uint8_t* hp = (uint8_t*)b->hb_res.ensure(bytes);
    memset(hp, 0, bytes);
    b->h_cnt = (Counts*)hp;
    v.res_cnt = (Counts*)hp; v.res_rn_total = (int64_t*)(hp + sizeof(Counts)); b->h_rn_total = v.res_rn_total;
    v.res_off = v.res_rn_total + 1; v.res_cov = (double*)(v.res_off + T + 1);
    v.res_status = (int32_t*)(v.res_cov + T);
    v.res_out = (OutHdr*)(((uintptr_t)(v.res_status + T) + 7) & ~(uintptr_t)7);
  }
  std::vector<int32_t> tid(T), svs(T), clen(T), psn(T); std::vector<double> nmt(T);
  for (int t = 0; t < T; t++) {
    tid[t] = b->tasks[t].task_id; svs[t] = b->tasks[t].sv_id_start; clen[t] = b->tasks[t].contig_len;
    psn[t] = b->tasks[t].ps_null_rank; nmt[t] = b->tasks[t].qc_nm_threshold;
  }
  v.t_task_id = upload_vec(b, tid); v.t_sv_id_start = upload_vec(b, svs); v.t_contig_len = upload_vec(b, clen);
  v.t_ps_null = upload_vec(b, psn); v.t_qc_nm_thr = upload_vec(b, nmt);
  v.t_lead_off = upload_vec(b, b->h_lead_off); v.t_read_off = upload_vec(b, b->h_read_off); v.t_tr_off = upload_vec(b, b->h_tr_off);
  v.t_has_tr = upload_vec(b, b->h_has_tr);
  v.t_status = dalloc<int32_t>(b, T + 1); v.t_call_off = dalloc<int64_t>(b, T + 2); v.t_cov_avg = dalloc<double>(b, T + 1);
  v.t_stale_end = dalloc<int32_t>(b, T + 1); v.t_cov_sum = dalloc<unsigned long long>(b, T + 1);
  dsync(b);   // the small host vectors above go out of scope
  // ---- (1) stage: arena layout = the batch-wide columns back to back (256-B aligned), then the sequence pool
  size_t off[IC_COUNT]; size_t at = 0;
  for (int c = 0; c < IC_COUNT; c++) { off[c] = at; at += (((size_t)(c < IC_RSTART ? N : R) * kInElem[c]) + 255) & ~(size_t)255; }
  const size_t col_bytes = at, pool_at = at;
  at += ((size_t)v.pool_len + 255) & ~(size_t)255;
  b->h_rend_max.assign((size_t)T, 0);
  uint8_t* d_in = dalloc<uint8_t>(b, col_bytes + 256);
  v.pool = dalloc<uint8_t>(b, (size_t)v.pool_cap);
  double t_staged = 0, t_copied = 0;
  std::vector<int32_t> top;
  {
    std::lock_guard<std::mutex> hold(g_stage.mu);   // one upload at a time stages through the arena
    uint8_t* st = (uint8_t*)g_stage.ensure(at + 256);
    // Work items of at most ~4 MB: lead rows and read tables first (the columns' copy to HBM then runs while the sequence
    // pools are still being staged), pool pieces second.  Up to 32 host threads take them from a counter: a whole genome is
    // ~0.46 GB, one thread copies ~5 GB/s including the checks.
    struct Item { int t; int kind; int64_t lo, hi; };
    std::vector<Item> items[2];
    for (int t = 0; t < T; t++) {
      if (b->task_on_device[(size_t)t]) {   // born in HBM (extraction): copied device-to-device below; its reads end inside the contig
        b->h_rend_max[(size_t)t] = b->tasks[(size_t)t].contig_len;
        continue;
      }
      const snf_task_input_t& q = b->tasks[(size_t)t];
      for (int64_t i = 0; i < q.n_leads; i += 49152) items[0].push_back({t, 0, i, i + 49152 < q.n_leads ? i + 49152 : q.n_leads});
      items[0].push_back({t, 2, 0, 0});
      for (int64_t p = 0; p < q.seq_pool_len; p += (4 << 20)) items[1].push_back({t, 1, p, p + (4 << 20) < q.seq_pool_len ? p + (4 << 20) : q.seq_pool_len});
    }
    int nth = (int)std::thread::hardware_concurrency(); if (nth > 32) nth = 32; if (nth < 1) nth = 1;
    if (const char* e = getenv("SNF_UPLOAD_THREADS")) { nth = atoi(e); if (nth < 1) nth = 1; }
    std::mutex emu; std::string err;
    std::atomic<int> has_dash{0};
    auto run_items = [&](const std::vector<Item>& its) {
      std::atomic<size_t> next{0};
      auto work = [&]() {
        for (;;) {
          const size_t k = next.fetch_add(1);
          if (k >= its.size()) return;
          const Item& it = its[k];
          const snf_task_input_t& q = b->tasks[(size_t)it.t];
          try {
            if (it.kind == 0) stage_leads(q, st, off, b->h_lead_off[(size_t)it.t], b->h_pool_off[(size_t)it.t], it.lo, it.hi);
            else if (it.kind == 1) stage_pool(q, st, pool_at, b->h_pool_off[(size_t)it.t], it.lo, it.hi, &has_dash);
            else stage_reads(q, st, off, b->h_read_off[(size_t)it.t], &b->h_rend_max[(size_t)it.t]);
          } catch (const snf::Error& e) { std::lock_guard<std::mutex> g(emu); if (err.empty()) err = e.msg; }
        }
      };
      const int n_use = (size_t)nth < its.size() ? nth : (int)its.size();
      if (n_use <= 1) work();
      else {
        std::vector<std::thread> ths;
        for (int k = 0; k < n_use; k++) ths.emplace_back(work);
        for (auto& th : ths) th.join();
      }
      if (!err.empty()) fail(err);
    };
    run_items(items[0]);
    h2d(b, d_in, st, col_bytes);                       // (asynchronous: pinned source) overlaps the staging of the pools
    try { run_items(items[1]); }
    catch (...) {
      // the copy above still reads the shared arena: it must have landed before the lock is released to another upload, and
      // the caller's arrays are not ours to keep after a failed upload
      dsync(b);
      for (auto& t : b->tasks) { snf_task_input_t e{}; e.task_id = t.task_id; e.contig_len = t.contig_len; t = e; }
      throw;
    }
    t_staged = now_ms();
    v.cons_thread_only = has_dash.load();
    h2d(b, v.pool, st + pool_at, (size_t)v.pool_len);
    // tasks whose columns are already in HBM (snf_batch_add_task_device): device-to-device into their slices, sequence
    // offsets rebased into the batch pool by a kernel
    bool any_dev = false;
    for (int t = 0; t < T; t++) {
      if (!b->task_on_device[(size_t)t]) continue;
      any_dev = true;
      const snf_task_input_t& q = b->tasks[(size_t)t];
      const int64_t l0 = b->h_lead_off[(size_t)t], r0 = b->h_read_off[(size_t)t], p0 = b->h_pool_off[(size_t)t];
      const void* src[22] = {q.ref_start, q.ref_end, q.qry_start, q.qry_end, q.svlen, q.read_len, q.qname_id, q.read_id, q.ps_rank,
                             q.mate_contig, q.mate_ref_start, q.seq_len, q.seq_off, q.nm, q.svtype, q.strand, q.mapq, q.source, q.hap,
                             q.is_sa, q.bnd_is_first, q.bnd_is_reverse};
      for (int c = 0; c < 22; c++) d2d(b, d_in + off[c] + (size_t)l0 * kInElem[c], src[c], (size_t)q.n_leads * kInElem[c]);
      d2d(b, v.pool + p0, q.seq_pool, (size_t)q.seq_pool_len);
      d2d(b, d_in + off[IC_RSTART] + (size_t)r0 * 4, q.read_start, (size_t)q.n_reads * 4);
      d2d(b, d_in + off[IC_REND] + (size_t)r0 * 4, q.read_end, (size_t)q.n_reads * 4);
      d2d(b, d_in + off[IC_RHP] + (size_t)r0, q.read_hp, (size_t)q.n_reads);
      RebaseView rv{(int64_t*)(d_in + off[IC_SEQ_OFF]) + l0, (const int32_t*)(d_in + off[IC_SEQ_LEN]) + l0, p0};
      LAUNCH_Q(u0_rebase, rv, q.n_leads, q.n_leads * 12);
    }
    // top level of the read-start index (every 256th start): from the staging arena, or back from HBM when some of the
    // starts never were on the host
    if (!any_dev) {
      const int32_t* rs_all = (const int32_t*)(st + off[IC_RSTART]);
      for (int64_t r = 0; r < R; r += (1 << SNF_TOP_SHIFT)) top.push_back(rs_all[r]);
    } else {
      for (int64_t r = 0; r < R; r += (1 << SNF_TOP_SHIFT)) top.push_back(0);
      // every 256th start in ONE strided copy (width 4 bytes, source pitch 1 KB) instead of a 4-byte copy per entry
      if (!top.empty()) SNF_HIP(hipMemcpy2DAsync(top.data(), 4, d_in + off[IC_RSTART], (size_t)4 << SNF_TOP_SHIFT, 4, top.size(), hipMemcpyDeviceToHost, b->cur));
    }
    dsync(b);
    for (auto& t : b->tasks) {   // the borrowed arrays are not referenced after this point
      snf_task_input_t s{}; s.task_id = t.task_id; s.sv_id_start = t.sv_id_start; s.contig_len = t.contig_len; s.ps_null_rank = t.ps_null_rank;
      s.qc_nm_threshold = t.qc_nm_threshold; s.n_leads = t.n_leads; s.n_reads = t.n_reads; s.n_tr = t.n_tr; s.seq_pool_len = t.seq_pool_len;
      t = s;
    }
    t_copied = now_ms();
  }
  v.in_ref_start = (const int32_t*)(d_in + off[IC_REF_START]); v.in_ref_end = (const int32_t*)(d_in + off[IC_REF_END]);
  v.in_qry_start = (const int32_t*)(d_in + off[IC_QRY_START]); v.in_qry_end = (const int32_t*)(d_in + off[IC_QRY_END]);
  v.in_svlen = (const int32_t*)(d_in + off[IC_SVLEN]); v.in_read_len = (const int32_t*)(d_in + off[IC_READ_LEN]);
  v.in_qname = (const uint32_t*)(d_in + off[IC_QNAME]); v.in_read_id = (const uint32_t*)(d_in + off[IC_READ_ID]);
  v.in_ps = (const int32_t*)(d_in + off[IC_PS]); v.in_mate_contig = (const int32_t*)(d_in + off[IC_MATE_CONTIG]);
  v.in_mate_pos = (const int32_t*)(d_in + off[IC_MATE_POS]); v.in_seq_len = (const int32_t*)(d_in + off[IC_SEQ_LEN]);
  v.in_seq_off = (const int64_t*)(d_in + off[IC_SEQ_OFF]); v.in_nm = (const double*)(d_in + off[IC_NM]);
  v.in_svtype = d_in + off[IC_SVTYPE]; v.in_strand = d_in + off[IC_STRAND]; v.in_mapq = d_in + off[IC_MAPQ];
  v.in_source = d_in + off[IC_SOURCE]; v.in_hap = d_in + off[IC_HAP]; v.in_is_sa = d_in + off[IC_IS_SA];
  v.in_first = d_in + off[IC_FIRST]; v.in_rev = d_in + off[IC_REV];
  v.r_start = (const int32_t*)(d_in + off[IC_RSTART]); v.r_end = (const int32_t*)(d_in + off[IC_REND]); v.r_hp = d_in + off[IC_RHP];
  {  // ---- (3) derived on the device
    PackView q{};
    q.t_lead_off = v.t_lead_off; q.t_read_off = v.t_read_off; q.T = T; q.N = N; q.R = R;
    q.ref_start = v.in_ref_start; q.ref_end = v.in_ref_end; q.qry_start = v.in_qry_start; q.qry_end = v.in_qry_end; q.svlen = v.in_svlen;
    q.read_len = v.in_read_len; q.ps = v.in_ps; q.mate_contig = v.in_mate_contig; q.mate_pos = v.in_mate_pos; q.seq_len = v.in_seq_len;
    q.qname = v.in_qname; q.read_id = v.in_read_id; q.seq_off = v.in_seq_off;
    q.svtype = v.in_svtype; q.strand = v.in_strand; q.mapq = v.in_mapq; q.source = v.in_source; q.hap = v.in_hap; q.is_sa = v.in_is_sa;
    q.first = v.in_first; q.rev = v.in_rev;
    LeadRec* rec = dalloc<LeadRec>(b, (size_t)N); int32_t* lt = dalloc<int32_t>(b, (size_t)N); int32_t* rt = dalloc<int32_t>(b, (size_t)R);
    q.rec = rec; q.lead_task = lt; q.r_task = rt;
    LAUNCH_Q(u1_pack, q, N, N * 136);
    LAUNCH_Q(u2_readtask, q, R, R * 4);
    v.in_rec = rec; v.lead_task = lt; v.r_task = rt;
  }
  v.rk_in = dalloc<uint64_t>(b, R); v.rk_out = dalloc<uint64_t>(b, R); v.rv_in = dalloc<uint32_t>(b, R); v.rv_out = dalloc<uint32_t>(b, R);
  v.re_sorted = dalloc<int32_t>(b, R);
  v.pc_s2 = dalloc<uint64_t>(b, R + 1); v.pc_e2 = dalloc<uint64_t>(b, R + 1);
  v.rs_top = upload_vec(b, top, 1); v.re_top = dalloc<int32_t>(b, top.size() + 1);
  v.rs_mid = dalloc<int32_t>(b, (size_t)(R + 15) / 16 + 16); v.re_mid = dalloc<int32_t>(b, (size_t)(R + 15) / 16 + 16);
  dsync(b);
  ReadPrep& rp = b->rp;
  {
    std::vector<uint64_t> base((size_t)T + 1, 0);
    for (int t = 0; t < T; t++) base[t + 1] = base[t] + (uint64_t)(uint32_t)b->h_rend_max[t] + 1;
    rp.t_base = upload_vec(b, base);
    b->read_key_bits = bits_for(base[T]);
    rp.key32 = (!sort64 && b->read_key_bits <= 32) ? 1 : 0;
  }
  rp.r_end = v.r_end; rp.r_hp = v.r_hp; rp.r_task = v.r_task; rp.r_start = v.r_start; rp.rk_in = v.rk_in; rp.rv_in = v.rv_in;
  rp.rk_out = v.rk_out; rp.rv_out = v.rv_out; rp.re_sorted = v.re_sorted; rp.re_top = v.re_top; rp.R = R;
  rp.rs_mid = v.rs_mid; rp.re_mid = v.re_mid;
  rp.fs2 = dalloc<uint64_t>(b, R + 1); rp.fe2 = dalloc<uint64_t>(b, R + 1);
  v.tr_start = upload_vec(b, b->h_trs); v.tr_end = upload_vec(b, b->h_tre); v.tr_pmax = upload_vec(b, b->h_trp);
  v.nm_start = nullptr; v.nm_end = nullptr; v.t_nm_off = nullptr; v.t_cov_exact = nullptr;
  if (!b->h_nms.empty()) { v.nm_start = upload_vec(b, b->h_nms); v.nm_end = upload_vec(b, b->h_nme); v.t_nm_off = upload_vec(b, b->h_nm_off); }
  b->d_max_depth = dalloc<int32_t>(b, (size_t)T + 1);
  dzero(b, b->d_max_depth, sizeof(int32_t) * ((size_t)T + 1));
  size_t N1 = (size_t)N + 1;
  v.key_in = dalloc<uint64_t>(b, N); v.key_out = dalloc<uint64_t>(b, N); v.val_in = dalloc<uint32_t>(b, N); v.val_out = dalloc<uint32_t>(b, N);
  {  // occupancy prefilter (snf_stage_cluster.h a0_*): only worth it when a cell with one lead can never seed a cluster
    v.NS = N; v.prefilter = 0;
    std::vector<int64_t> cell_off((size_t)T + 1, 0);
    const int bs = b->cfg.cluster_binsize > 0 ? b->cfg.cluster_binsize : 1;
    for (int t = 0; t < T; t++) cell_off[(size_t)t + 1] = cell_off[(size_t)t] + (int64_t)SNF_NTYPES * ((int64_t)b->tasks[(size_t)t].contig_len / bs + 1);
    const int64_t cells = cell_off[(size_t)T];
    if (N > 0 && b->cfg.dev_min_leads_cluster >= 2 && getenv("SNF_NO_PREFILTER") == nullptr && cells < ((int64_t)1 << 36)) {
      v.prefilter = 1;
      v.t_cell_off = upload_vec(b, cell_off);
      b->pf_words = (((cells >> 19) + 1) << 19) / 16 + 2;   // (whole 2^19-cell blocks: pf_slot permutes inside a block)
      v.pf_spread = (getenv("SNF_PF_SPREAD") && atoi(getenv("SNF_PF_SPREAD")) == 1) ? 1 : 0;   // (measured: a1_keys 0.146 ms spread / 0.111 adjacent)
      v.pf_bm = dalloc<uint32_t>(b, (size_t)b->pf_words);
      dzero(b, v.pf_bm, (size_t)b->pf_words * 4);
      v.pf_key = dalloc<uint64_t>(b, N); v.pf_keep = dalloc<uint32_t>(b, N1); v.pf_scan = dalloc<uint32_t>(b, N1);
      dsync(b);   // cell_off goes out of scope
    }
  }
  // window front end (snf_stage_window.h): same precondition as the prefilter (a one-lead bin can never seed a cluster); the window
  // width is chosen below, once the leads can be counted per window - the arrays are sized for the narrowest window tried
  const int WIN_W_MAX = getenv("SNF_WIN_BITS_MAX") ? atoi(getenv("SNF_WIN_BITS_MAX")) : 10, WIN_W_MIN = 6;   // (measured on the 30x genome: 10 beats 9 and 8)
  auto win_layout = [&](int W, std::vector<int64_t>& off) {
    const int bs = b->cfg.cluster_binsize > 0 ? b->cfg.cluster_binsize : 1;
    off.assign((size_t)T + 1, 0);
    for (int t = 0; t < T; t++) off[(size_t)t + 1] = off[(size_t)t] + (int64_t)SNF_NTYPES * ((((int64_t)b->tasks[(size_t)t].contig_len / bs + 1) >> W) + 1);
    return off[(size_t)T];
  };
  int64_t win_slots_max = 0;
  const bool front_wanted = v.prefilter && v.wave_path && getenv("SNF_NO_WINFRONT") == nullptr && getenv("SNF_NO_FUSE") == nullptr && N <= ((int64_t)1 << 25);
  if (front_wanted) {
    std::vector<int64_t> off;
    win_slots_max = win_layout(WIN_W_MIN, off);
    if (win_slots_max >= ((int64_t)1 << 31)) win_slots_max = 0;
  }
  v.front = 0; v.NW = 0;
  if (win_slots_max > 0) {
    const size_t W1 = (size_t)win_slots_max + 1;
    v.wcnt = dalloc<uint32_t>(b, W1); v.wfill = dalloc<uint32_t>(b, W1); v.wbase = dalloc<uint32_t>(b, W1);
    const size_t occ_max = (size_t)(N < win_slots_max ? N : win_slots_max) + 2;
    const size_t nblk = (size_t)N / 64 + 2;
    v.wlist = dalloc<uint32_t>(b, 4 * occ_max); v.blk_k0 = dalloc<uint32_t>(b, nblk);
    v.ws_seeds = dalloc<uint32_t>(b, nblk); v.ws_nf = dalloc<uint32_t>(b, nblk); v.ws_nl = dalloc<uint32_t>(b, nblk);
    v.whead = dalloc<uint64_t>(b, N1); v.whead2 = dalloc<uint64_t>(b, N1);
    dzero(b, v.wcnt, W1 * 4); dzero(b, v.wfill, W1 * 4);
  }
  uint32_t** u32s[] = {&v.headflag, &v.headscan, &v.eligflag, &v.eligscan, &v.fN, &v.pN, &v.fL, &v.pL, &v.runflag, &v.runscan,
                       &v.clflag, &v.clscan, &v.rcflag, &v.rcscan, &v.cdflag, &v.cdscan, &v.rnf, &v.rnp};
  for (auto pp : u32s) *pp = dalloc<uint32_t>(b, N1);
  v.seqnull = dalloc<uint8_t>(b, N); v.bin_lo = dalloc<int32_t>(b, N1); v.bin_key = dalloc<uint64_t>(b, N);
  v.bin_hap = dalloc<uint16_t>(b, 3 * (size_t)N); v.bin_elig = dalloc<uint8_t>(b, N);
  v.grp_first_bin = dalloc<int32_t>(b, 8 * (size_t)T + 8);
  v.L = dalloc<uint32_t>(b, N); v.LL = dalloc<uint32_t>(b, N); v.Lrec = dalloc<LeadRec>(b, N1); v.chdr = dalloc<ClusterHdr>(b, N1);
  int32_t** i32s[] = {&v.seed_bin, &v.seed_lo, &v.seed_hi, &v.seedL_lo, &v.seedL_hi, &v.seed_start, &v.seed_grp, &v.c_last, &v.c_end,
                      &v.nxt, &v.prv, &v.run_first, &v.run_last_head, &v.cl_head, &v.w0, &v.w1, &v.w2, &v.w3, &v.w4, &v.w5, &v.w6, &v.w7,
                      &v.F_orig, &v.F_svlen, &v.F_seq_len, &v.FI, &v.F_lpos, &v.rc_n_s, &v.rc_cl_s, &v.rc_lo, &v.rc_n, &v.rc_cluster};
  for (auto pp : i32s) *pp = dalloc<int32_t>(b, N1);
  double** f64s[] = {&v.s_mean0, &v.s_stdev0, &v.c_mean, &v.c_stdev, &v.run_b_stdev, &v.run_b_absmean};
  for (auto pp : f64s) *pp = dalloc<double>(b, N1);
  uint8_t** u8s[] = {&v.s_repeat0, &v.c_repeat, &v.run_b_repeat, &v.F_sel, &v.rc_keeplong_s, &v.rc_keeplong};
  for (auto pp : u8s) *pp = dalloc<uint8_t>(b, N1);
  v.c_ms = dalloc<ClusterSums>(b, N1);
  v.grp_dirty = dalloc<int32_t>(b, 8 * (size_t)T + 8); v.grp_seed_lo = dalloc<int32_t>(b, 8 * (size_t)T + 8);
  v.grp_seed_hi = dalloc<int32_t>(b, 8 * (size_t)T + 8);
  v.F_seq_off = dalloc<int64_t>(b, N1);
  v.cand = dalloc<snf_call_t>(b, N1); v.candx = dalloc<CallX>(b, N1);
  v.calls = dalloc<snf_call_t>(b, N1); v.callx = dalloc<CallX>(b, N1);
  v.rnames = dalloc<uint32_t>(b, 2 * (size_t)N + 1);
  const auto& lut = build_gt_lut(b->cfg);
  v.gt_lut = upload_vec(b, lut);
  v.cons_call = dalloc<int32_t>(b, N1);
  v.stripes = dalloc<unsigned long long>(b, 4 * 64 * 16);
  v.tile_stride = (int64_t)((N1 > (size_t)win_slots_max + 1 ? N1 : (size_t)win_slots_max + 1) / 256 + 2); v.tile_sums = dalloc<unsigned long long>(b, (size_t)v.tile_stride * TS_SLOTS);
  v.super_stride = v.tile_stride / 64 + 2; v.tile_super = dalloc<unsigned long long>(b, (size_t)v.super_stride * TS_SLOTS);
  v.chain_stride = v.tile_stride; v.chain = dalloc<unsigned long long>(b, (size_t)v.chain_stride * TS_SLOTS * 2);
  v.chain_ticket = dalloc<uint32_t>(b, TS_SLOTS + 2); v.chain_epoch = v.chain_ticket + TS_SLOTS;
  dzero(b, v.chain, (size_t)v.chain_stride * TS_SLOTS * 16); dzero(b, v.chain_ticket, (TS_SLOTS + 2) * 4);   // (a recycled slab may hold another batch's tags)
  // Single-launch chains and the pass graph pay where a pass is launch-bound, i.e. for small batches (measured on MI355X, same box:
  // chr20 alone, 60 k signatures: 0.435 against 0.462 ms per step replayed from a graph; whole genome, 2.87 M signatures: the
  // look-back chains cost what their launch pairs cost - 2 700 tiles queue behind each other at ~22 ns a tile: c4 59 us against
  // 7.6 + 6.8 - and two replayed passes next to each other are 2.5 % SLOWER than two eager ones, 1.67 against 1.63 ms per step).
  // SNF_CHAIN / SNF_GRAPH = 0 / 1 force either way.
  const bool small_batch = N <= 400000;
  v.chain_on = getenv("SNF_CHAIN") ? (atoi(getenv("SNF_CHAIN")) != 0) : (getenv("SNF_NO_CHAIN") ? 0 : (small_batch ? 1 : 0));
  b->graph_mode = getenv("SNF_GRAPH") ? (atoi(getenv("SNF_GRAPH")) != 0 ? 2 : 0) : (getenv("SNF_NO_GRAPH") ? 0 : (small_batch ? 2 : 0));
  v.big_cap = (int64_t)(N1 / 64 + 2); v.big_cnt = dalloc<uint32_t>(b, 3 * 64 * 16); v.big_list = dalloc<int32_t>(b, (size_t)(3 * 64 * v.big_cap));
  v.big_wave = v.wave_path;
  { const int hn = getenv("SNF_HEAVY_N") ? atoi(getenv("SNF_HEAVY_N")) : 24; v.heavy_n = hn > 8 && hn < 64 ? hn : 0; }      // (0 / out of range: one class, as before)
  { const int eb = getenv("SNF_E1_BATCH") ? atoi(getenv("SNF_E1_BATCH")) : 64; v.e1_batch = (eb == 2 || eb == 4 || eb == 8 || eb == 16 || eb == 32) ? eb : 64; }
  v.stage_cap = getenv("SNF_NO_BIG_STAGE") ? 0 : 1;   // x_big<0>: clusters up to SNF_BIG_STAGE_CAP leads are kept in LDS
  v.cdesc = dalloc<ConsDesc>(b, N1); v.crl_off = dalloc<int64_t>(b, N1); v.crl_len = dalloc<int32_t>(b, N1); v.aln_kept_w = dalloc<uint8_t>(b, N1);
  for (int k = 0; k < 8; k++) v.cls_list[k] = k == 6 ? nullptr : dalloc<int32_t>(b, N1);
  v.d2cap = (int64_t)(N1 / 64 + 64);
  for (int k = 0; k < 3; k++) v.d2_list[k] = dalloc<int32_t>(b, (size_t)(64 * v.d2cap));
  v.d2cnt = dalloc<uint32_t>(b, 3 * 64 * 16);
  v.d2_from_list = 0; v.d1_from_list = 0;
#ifdef SNF_ITRACE
  if (!v.itrace) { SNF_HIP(hipMalloc((void**)&v.itrace, (size_t)SNF_IT_SLOTS * SNF_IT_CAP * 8)); SNF_HIP(hipMemset(v.itrace, 0, (size_t)SNF_IT_SLOTS * SNF_IT_CAP * 8)); }
#endif
#ifdef SNF_WG_TRACE
  if (!v.wgtrace) { SNF_HIP(hipMalloc((void**)&v.wgtrace, (size_t)(1 << 20) * 24)); SNF_HIP(hipMemset(v.wgtrace, 0, (size_t)(1 << 20) * 24)); }
#endif
  v.cons_tab_off = dalloc<int64_t>(b, N1 + 1); v.cons_aln_off = dalloc<int64_t>(b, N1 + 1); v.cons_read_off = dalloc<int64_t>(b, N1 + 1);
  v.cons_tab_sz = dalloc<int64_t>(b, N1 + 1);
  v.sz_tab = dalloc<int64_t>(b, N1 + 1); v.sz_aln = dalloc<int64_t>(b, N1 + 1); v.sz_rd = dalloc<int64_t>(b, N1 + 1);
  v.sc_tab = dalloc<int64_t>(b, N1 + 1); v.sc_aln = dalloc<int64_t>(b, N1 + 1); v.sc_rd = dalloc<int64_t>(b, N1 + 1);
  b->readprep_each_pass = getenv("SNF_READPREP_EACH_PASS") != nullptr;
  const double t_alloc = now_ms();
  if (v.prefilter) {
    // how many leads the prefilter keeps is a property of the input: counted once here (the same three kernels every pass
    // runs), so that every later pass can size its sort and launches on the host without a round trip
    const bool tm = b->timing; b->timing = false;
    enqueue_pass_init(b);
    enqueue_keys(b);
    b->timing = tm;
    Counts hc{};
    d2h(b, &hc, v.cnt, sizeof(Counts));
    dsync(b);
    v.NS = hc.n_kept;
    if (v.prof) fprintf(stderr, "[SNF_PROF] prefilter: %lld of %lld leads share their (svtype, bin) cell with another lead\n", (long long)v.NS, (long long)N);
  }
  const double t_pfcount = now_ms();
  if (win_slots_max > 0 && v.NS > 0) {
    // window width: the widest whose largest window fits the 256-lead instance of the window kernels, else the widest that fits the
    // 1024-lead one; occupied windows and the largest window are properties of the input (counted here, like NS): the passes size
    // their launches on the host
    int forced = getenv("SNF_WIN_BITS") ? atoi(getenv("SNF_WIN_BITS")) : 0;
    int best_w = -1; int64_t best_occ = 0, best_max = 0, best_big = 0;
    std::vector<int64_t> off;
    if (forced && (forced < WIN_W_MIN || forced > WIN_W_MAX)) forced = 0;
    for (int W = forced ? forced : WIN_W_MAX; W >= (forced ? forced : WIN_W_MIN); W--) {
      v.NW = win_layout(W, off); v.win_bits = W;
      const int64_t* d_off = upload_vec(b, off);
      v.t_win_off = d_off;
      const bool tm = b->timing; b->timing = false;
      enqueue_pass_init(b);
      LAUNCH_Q(w1_hist, v, N, 0);
      LAUNCH_Q(w0_stats, v, v.NW, 0);
      b->timing = tm;
      Counts hc{};
      d2h(b, &hc, v.cnt, sizeof(Counts));
      dzero(b, v.wcnt, ((size_t)v.NW + 1) * 4);
      dsync(b);
      if (hc.max_win <= 256 || (best_w < 0 && hc.max_win <= SNF_WIN_MAXCAP)) { best_w = W; best_occ = hc.n_occ; best_max = hc.max_win; best_big = hc.n_big64; }
      if (hc.max_win <= 256) break;
    }
    if (best_w >= 0) {
      if (v.win_bits != best_w) { v.NW = win_layout(best_w, off); v.win_bits = best_w; v.t_win_off = upload_vec(b, off); dsync(b); }
      v.front = 1; b->h_n_occ = best_occ; b->win_cap = best_max <= 64 ? 64 : best_max <= 256 ? 256 : SNF_WIN_MAXCAP;
      b->h_n_big64 = best_big; v.w4_list = dalloc<int32_t>(b, (size_t)best_big + 64); v.w4_mode = 0;
      b->w4_split = getenv("SNF_W4_SPLIT") && atoi(getenv("SNF_W4_SPLIT")) != 0;
    }
    if (v.prof) fprintf(stderr, "[SNF_PROF] window front end: %s (W = %d: %lld windows, %lld occupied, largest %lld leads, %lld of more than 64: w4s_segment in %s)\n", v.front ? "on" : "off",
                        v.win_bits, (long long)v.NW, (long long)best_occ, (long long)best_max, (long long)best_big,
                        (b->w4_split && b->win_cap > 64 && best_big > 0 && best_big * 8 < (N + 63) / 64) ? "two launches" : "one launch");
  }
  const double t_winsel = now_ms();
  {  // ALT stage output (HBM; every ALT is the sequence of one lead of its cluster, so all of them together fit the pool) and
     // the output stage: at most one record per position behind the sort
    v.alt_cap = v.pool_cap; v.alt_pool = dalloc<uint8_t>(b, (size_t)v.alt_cap + 32);
    const size_t NO = (size_t)v.NS + 1;
    // (indexed by positions behind the sort: sized by N, not NS - snf_batch_fetch_clusters turns the prefilter off and the plain-scan
    // form of the output stage then walks N + 1 positions)
    v.o_scan = dalloc<uint32_t>(b, N1 + 1); v.o_src = dalloc<int32_t>(b, N1); v.o_dst = dalloc<int32_t>(b, N1); v.o_key = dalloc<int32_t>(b, N1);
    v.o_rn = dalloc<int64_t>(b, N1);
    v.out_hdr = dalloc<OutHdr>(b, 1);
    dzero(b, v.out_hdr, sizeof(OutHdr));
    v.out_dev_cap = (int64_t)(NO * sizeof(snf_call_t) + (2 * (size_t)N + 1) * 4 + 1024);
    v.out_dev = dalloc<uint8_t>(b, (size_t)v.out_dev_cap);
    v.out_mode = b->out_mode; v.out_valid = 0; v.out_pin = nullptr; v.out_pin_cap = 0; v.alt_pin = nullptr; v.alt_pin_cap = 0;
  }
  const double t_index0 = now_ms();
  { const bool tm = b->timing; b->timing = false; enqueue_read_index(b); b->timing = tm; }   // (no event brackets outside a pass)
  {  // can the uint16 coverage vector wrap (leadprov.py:451)?  Only where 65536 reads overlap: the largest depth of every task,
     // once - tasks with fewer reads than that are skipped on the host
    bool any_deep = false;
    for (int t = 0; t < T; t++) any_deep |= b->tasks[(size_t)t].n_reads >= 65536;
    std::vector<int32_t> maxd((size_t)T, 0);
    if (any_deep && R > 0) {
      MaxDepth md{v.r_start, v.re_sorted, v.rs_top, v.re_top, v.r_task, v.t_read_off, b->d_max_depth};
      const bool tm = b->timing; b->timing = false;
      LAUNCH_Q(r3_maxdepth, md, R, R * 8);
      b->timing = tm;
      d2h(b, maxd.data(), b->d_max_depth, sizeof(int32_t) * (size_t)T);
      dsync(b);
    }
    b->h_cov_exact.assign((size_t)T, 0);
    bool any = false;
    for (int t = 0; t < T; t++) {
      const bool masked = b->h_nm_off[(size_t)t + 1] > b->h_nm_off[(size_t)t];
      if (masked || maxd[(size_t)t] >= 65536 || getenv("SNF_COV_EXACT")) { b->h_cov_exact[(size_t)t] = 1; any = true; }
    }
    if (any) v.t_cov_exact = upload_vec(b, b->h_cov_exact);
  }
  dsync(b);
  b->reads_ready = true;
  b->uploaded = true;
  if (v.prof) fprintf(stderr, "[SNF_PROF] read index (sorted ends, hap prefix counts): %.2f ms\n", now_ms() - t_index0);
  if (v.prof) fprintf(stderr, "[SNF_PROF] upload: %.1f ms (stage %.1f, H2D %.1f [%.1f MB], allocations + derived %.1f = packing + arrays %.2f, prefilter count %.2f, "
                              "window width %.2f, read index %.2f; %zu device allocations)\n",
                      now_ms() - t_begin, t_staged - t_begin, t_copied - t_staged, (double)at / 1e6, now_ms() - t_copied, t_alloc - t_copied, t_pfcount - t_alloc,
                      t_winsel - t_pfcount, now_ms() - t_winsel, b->bufs.size());
}

// ---------------------------------------------------------------------------------------------- pipeline
// passes that have been enqueued and not yet waited for, over all handles of the process (a pass = snf_batch_pass, or
// call_candidates .. the fetch / sync that waits for it)
// All of it is state of ONE device (what overlaps is the work on that device and the traffic on its PCIe link): a process that drives
// several devices from one handle each must not see one device's passes switch another device's handle to the staged result or pace
// its starts (tests/test_output_modes.py::test_pacing_state_is_per_device).
struct DevicePacing {
  std::atomic<int> passes_in_flight{0};
  std::atomic<long long> last_overlap_ms{-1000000};      // when two passes were last in flight together (now_ms clock)
  std::mutex copy_mu, pace_mu;
  double last_pass_start_ms = -1e12, pass_latency_ms = 0.0;      // (under pace_mu) start of the latest pass; smoothed enqueue -> waited time
  // the chain gate (SNF_CHAIN_GATE=1; under pace_mu): the passes of a device take turns with their clustering / calling chains ON THE DEVICE - a pass's
  // main stream waits for the event the pass before it recorded where its chain ends (in front of its ALT stage), while that pass's ALT
  // and output stages run beside the new chain
  hipEvent_t chain_ev[2] = {nullptr, nullptr}; unsigned long long chain_seq = 0;
};
DevicePacing g_pacing[SNF_MAX_DEVICES];
DevicePacing& pacing_of(const snf_batch_impl* b) { return g_pacing[(b->device >= 0 && b->device < SNF_MAX_DEVICES) ? b->device : 0]; }
// Pacing of overlapping passes.  Two passes in flight run best OUT OF PHASE - one computes while the other's result crosses PCIe
// (staged result, run_finalize).  Left alone, two host threads fall into step in about one run of five and stay there: both passes start
// together, share the device through all their kernels, reach their copies together and share the link too - 1.24-1.38 ms per step
// instead of 0.95-1.0 (profiles/r05_pace.log: 2 of 10 runs; 3 of 12 in ab_r05_5.log).  Two cheap rules keep them apart, each sufficient
// in 8 of 8 runs, both together the default: (1) the copies of a staged result are taken one pass at a time (copy_mu: the link is
// shared anyway; the pass that waited starts its next pass later - a stagger), (2) a pass does not START sooner than a quarter of the
// recent pass latency after the other in-flight pass did (the second of two simultaneous starts waits ~0.4 ms once; passes that are
// half a period apart never wait).  SNF_PACE=0 turns both off (2: rule 1 only, 3: rule 2 only), SNF_PACE_FRAC sets the fraction.
// SNF_CHAIN_GATE=1 (off by default): passes of one device take turns with their chains ON THE DEVICE instead of being kept apart by the two
// timing rules below.  Measured on the last day of round 6 (tools/regime.sh, profiles/r06_regime*.log): 8 of 8 and 8 of 8 runs at 0.954-0.977
// ms per step (with rule 1) / 0.959-0.961 (without any rule) on one box - and, made the default, 12 of 16 runs at 0.955-0.97 but FOUR at
// 1.32-1.49 (the passes one after the other) on the next.  The rules: 37 of 40 default runs of that day at 0.94-0.97, three at 1.15-1.19.
// Not understood in the time left; the rules stay the default.
bool chain_gate_on() { static const bool g = getenv("SNF_CHAIN_GATE") && atoi(getenv("SNF_CHAIN_GATE")) != 0; return g; }
int pace_mode() { static const int m = getenv("SNF_PACE") ? atoi(getenv("SNF_PACE")) : 1; return m; }      // 0 off, 1 both rules, 2 copies in turn only, 3 spaced starts only
bool pace_on() { return pace_mode() == 1 || pace_mode() == 3; }
bool pace_copy_turn() { return pace_mode() == 1 || pace_mode() == 2; }
double pace_frac() { static const double f = getenv("SNF_PACE_FRAC") ? atof(getenv("SNF_PACE_FRAC")) : 0.25; return f; }
// GPU slots (SNF_GPU_SLOTS=n, off by default): at most n passes - of any process of this user - drive a device at a time.  The
// reference's deployment is a pool of worker PROCESSES (one per contig at most: 24); each of them brings its own HIP context, and the
// hardware schedules a limited number of queues: two dozen processes with four streams each are time-sliced by the driver and a
// pass that takes a millisecond alone takes tens (profiles/r05_workers_hw_queues.log).  A slot is an exclusive lock on one of n
// files under /dev/shm, taken when a pass is enqueued and dropped when its result has been waited for: the workers' host work -
// records into objects - goes on beside the n passes that hold the device.
int gpu_slots() { static const int n = getenv("SNF_GPU_SLOTS") ? atoi(getenv("SNF_GPU_SLOTS")) : 0; return n; }
int slot_acquire(int device) {
  const int n = gpu_slots();
  if (n <= 0) return -1;
  char path[160];
  const int start = (int)((unsigned)getpid() % (unsigned)n);
  for (int k = 0; k < n; k++) {       // any free slot, starting at one that depends on the process
    snprintf(path, sizeof path, "/dev/shm/snf_gpu_slot_%u_d%d_%d", (unsigned)getuid(), device, (start + k) % n);
    const int fd = open(path, O_CREAT | O_RDWR | O_CLOEXEC, 0600);
    if (fd < 0) return -1;
    if (flock(fd, LOCK_EX | LOCK_NB) == 0) return fd;
    close(fd);
  }
  snprintf(path, sizeof path, "/dev/shm/snf_gpu_slot_%u_d%d_%d", (unsigned)getuid(), device, start);
  const int fd = open(path, O_CREAT | O_RDWR | O_CLOEXEC, 0600);
  if (fd < 0) return -1;
  while (flock(fd, LOCK_EX) != 0) if (errno != EINTR) { close(fd); return -1; }
  return fd;
}
void pass_begins(snf_batch_impl* b) {
  if (b->in_flight) return;
  b->in_flight = true;
  if (b->slot_fd < 0) b->slot_fd = slot_acquire(b->device);
  DevicePacing& P = pacing_of(b);
  const bool other = P.passes_in_flight.fetch_add(1) >= 1;
  if (other) P.last_overlap_ms.store((long long)now_ms());
  if (!pace_on() || chain_gate_on()) { b->pass_start_ms = now_ms(); return; }
  double wait = 0.0;
  {
    std::lock_guard<std::mutex> g(P.pace_mu);
    const double now = now_ms();
    if (other && P.pass_latency_ms > 0.0) wait = P.last_pass_start_ms + pace_frac() * P.pass_latency_ms - now;
    if (wait > 2.0) wait = 2.0;                 // (never more than 2 ms, whatever the history says)
    if (wait < 0.0) wait = 0.0;
    P.last_pass_start_ms = now + wait;          // (the slot is taken; the wait itself happens outside the lock)
  }
  if (wait > 0.0) { struct timespec ts; ts.tv_sec = 0; ts.tv_nsec = (long)(wait * 1e6); nanosleep(&ts, nullptr); }
  b->pass_start_ms = now_ms();
}
void pass_waited(snf_batch_impl* b) {
  if (!b->in_flight) return;
  b->in_flight = false;
  if (b->slot_fd >= 0) { close(b->slot_fd); b->slot_fd = -1; }
  DevicePacing& P = pacing_of(b);
  if (P.passes_in_flight.fetch_sub(1) >= 2) P.last_overlap_ms.store((long long)now_ms());
  if (pace_on() && b->pass_start_ms > 0.0) {
    std::lock_guard<std::mutex> g(P.pace_mu);
    const double lat = now_ms() - b->pass_start_ms;
    P.pass_latency_ms = P.pass_latency_ms > 0.0 ? 0.75 * P.pass_latency_ms + 0.25 * lat : lat;
  }
}
// is this process driving several passes at a time?  (Asked when a pass is enqueued: the other handle may be between its fetch and its
// next pass at that very moment - what counts is whether passes overlapped a moment ago.)
bool passes_overlap(const snf_batch_impl* b) {
  DevicePacing& P = pacing_of(b);
  return P.passes_in_flight.load() > 1 || (long long)now_ms() - P.last_overlap_ms.load() < 100;
}
void reset_timing(snf_batch_impl* b) {
  b->ev_used = 0;
  b->timings.clear();
}
void begin_pass_timing(snf_batch_impl* b) {   // a pass begins: is it one of the sampled ones?
  reset_timing(b);
  b->time_now = b->timing && (b->time_all || b->time_every == 1 || (b->time_every > 1 && b->pass_count % (uint64_t)b->time_every == 0));
  b->pass_count++;
}

// reads: sorted ends + per-haplotype prefix counts = the device form of the LeadProvider's coverage vector and REF hap
// tables (leadprov.py:387-398, 451, 510).  The reference builds those while it extracts the signatures, BEFORE
// Task.call_candidates, and they depend on nothing but the read table: built once per batch, at upload
// (SNF_READPREP_EACH_PASS=1 rebuilds them in every call_candidates, as round 1 did; bench.py reports both).
void enqueue_read_index(snf_batch_impl* b) {
  View& v = b->v;
  const int64_t R = v.R;
  if (R <= 0) return;
  LAUNCH(r1_endkeys, b->rp, R, R * 13);
  if (b->rp.key32) prim_sort_pairs<uint32_t>(b, (uint32_t*)v.rk_in, (uint32_t*)v.rk_out, v.rv_in, v.rv_out, R, b->read_key_bits, "sort_read_ends");
  else prim_sort_pairs<uint64_t>(b, v.rk_in, v.rk_out, v.rv_in, v.rv_out, R, b->read_key_bits, "sort_read_ends");
  LAUNCH_Q(r2_unpack, b->rp, R, R * 16);
  // haplotype prefix counts: HP 1 and HP 2 packed in one 64-bit scan per order (HP 0 = rank - both)
  prim_exscan<uint64_t>(b, b->rp.fs2, v.pc_s2, R + 1, "scan_hap_prefix");
  prim_exscan<uint64_t>(b, b->rp.fe2, v.pc_e2, R + 1, "scan_hap_prefix");
}

void enqueue_read_prep(snf_batch_impl* b) {
  SNF_TRACE("coverage mean (side stream)");
  View& v = b->v;
  int64_t R = v.R; int T = v.T;
  // independent of the lead pipeline until d4_coverage, so it runs on the side stream
  {
  SideStream side(b);
  if (R > 0) {
    if (b->readprep_each_pass) enqueue_read_index(b);
    // Task.coverage_average_total = coverage.mean(): sum of the clipped read lengths (exact) / contig length
    { Scope _s(b, "d5w_covsum", R * 12);
      int64_t grid = (R + 4095) / 4096; if (grid > 2048) grid = 2048;
      hipLaunchKernelGGL(d5w_covsum, dim3((unsigned)grid), dim3(256), 0, b->cur, v, (int64_t)0);
      SNF_HIP(hipGetLastError()); }
  }
  if (v.t_cov_exact)
    for (int t = 0; t < T; t++) if (b->h_cov_exact[(size_t)t]) {   // (rare: masked / wrapping tasks) the exact sum of the masked uint16 vector
      CovExact p{};
      p.q.r_start = v.r_start; p.q.re_sorted = v.re_sorted; p.q.rs_top = v.rs_top; p.q.re_top = v.re_top;
      p.q.lo = b->h_read_off[(size_t)t]; p.q.hi = b->h_read_off[(size_t)t + 1]; p.q.L = b->tasks[(size_t)t].contig_len;
      p.q.nm_start = v.nm_start; p.q.nm_end = v.nm_end; p.q.nm_lo = b->h_nm_off[(size_t)t]; p.q.nm_hi = b->h_nm_off[(size_t)t + 1];
      p.out = v.t_cov_sum + t;
      LAUNCH_Q(d5x_covexact, p, (p.q.L + SNF_COVX_CHUNK - 1) / SNF_COVX_CHUNK, 0);
    }
  LAUNCH_Q(d5_covavg, v, T, 0);
  }
}

// the five coverage samples of every call: a thread per (call, sample) over a grid that covers a genome's calls in one round
// (d4s_coverage, snf_wave_call.h); SNF_D4=thread: the former thread-per-call kernel with its hinted binary searches
void launch_coverage(snf_batch_impl* b, int64_t N) {
  View& v = b->v;
  static const bool per_call = getenv("SNF_D4") && strcmp(getenv("SNF_D4"), "thread") == 0;
  if (per_call || !v.wave_path) { LAUNCH(d4_coverage, v, N, 0); return; }
  Scope _s(b, "d4_coverage", 0);
  int64_t grid = (5 * N + 255) / 256; if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(d4s_coverage, dim3((unsigned)grid), dim3(256), 0, b->cur, v, (int64_t)0);
  SNF_HIP(hipGetLastError());
}

// start of a pass: every small reset (one launch on the fused path)
void enqueue_pass_init(snf_batch_impl* b) {
  View& v = b->v;
  const int64_t N = v.N; const int T = v.T;
  b->fused = getenv("SNF_NO_FUSE") == nullptr && N <= ((int64_t)1 << 25);
  if (b->fused) {
    int64_t n0 = 8 * (int64_t)T + 8;
    if (TS_SLOTS * v.super_stride > n0) n0 = TS_SLOTS * v.super_stride;
    if ((int64_t)(sizeof(Counts) / 8) > n0) n0 = (int64_t)(sizeof(Counts) / 8);
    if (3 * 64 * 16 > n0) n0 = 3 * 64 * 16;
    FUSED(z0_init, n0);
  } else {
    dzero(b, v.cnt, sizeof(Counts));
    dzero(b, v.t_cov_sum, sizeof(unsigned long long) * (T + 1));
    dzero(b, v.t_status, sizeof(int32_t) * (T + 1));
    dzero(b, v.t_call_off, sizeof(int64_t) * (T + 2));
    dzero(b, v.grp_first_bin, sizeof(int32_t) * (8 * T + 8), 0xff);
    dzero(b, v.grp_seed_lo, sizeof(int32_t) * (8 * T + 8), 0xff);
    dzero(b, v.grp_seed_hi, sizeof(int32_t) * (8 * T + 8), 0xff);
    dzero(b, v.grp_dirty, sizeof(int32_t) * (8 * T + 8));
  }
}
// sort keys of the leads; with the occupancy prefilter: marks, keep flags and the compacted (key, index) pairs of the kept leads
void enqueue_keys(snf_batch_impl* b) {
  SNF_TRACE("A0: keys + occupancy prefilter");
  View& v = b->v;
  const int64_t N = v.N;
  if (N <= 0) return;
  LAUNCH(a1_keys, v, N, N * 21);
  if (!v.prefilter) return;
  if (b->fused) {
    { Scope _s(b, "a0_keep", N * 8); FUSED(a0k_keep, N); }
    FUSED(a0k_compact, N);
  } else {
    LAUNCH(a0_keep, v, N, N * 8);
    prim_exscan<uint32_t>(b, v.pf_keep, v.pf_scan, N + 1, "scan_keep");
    LAUNCH_Q(a0_compact, v, N, N * 12);
  }
}

// the window front end (snf_stage_window.h): from the input leads to the seed table and the packed lead records in six launches
void enqueue_window_front(snf_batch_impl* b) {
  SNF_TRACE("A: window front end");
  View& v = b->v;
  const int64_t N = v.N, NW = v.NW, n_occ = b->h_n_occ;
  if (N <= 0 || n_occ <= 0) return;
  Scope _all(b, "front_window", N * 21);
  LAUNCH_Q(w1_hist, v, N, 0);
  if (v.chain_on) CHAIN(w2c_offsets, NW);
  else { FUSED(w2a_sums, NW); FUSED(w2b_offsets, NW); }
  LAUNCH_Q(w3_scatter, v, N, 0);
  const int64_t n_blk = (N + 63) / 64;      // waves of w4s_segment: one per 64 positions of the bucket array
  // SNF_W4_SPLIT=1: two launches when few blocks need more than the 64-lead instance (82 registers and 3.6 KB of LDS against 98 and 8.6 KB:
  // six waves per SIMD instead of four, two unrolled rounds instead of five): the small instance over every block, the large one over the
  // list of blocks the small one left - at most as many as the input has windows of more than 64 leads (counted at upload).  Measured on the
  // 30x genome (profiles/ab_r06_19.log): 4 700-4 900 workgroups in flight instead of 3 800-4 000, each 5-10 % slower - the kernel is bound
  // by instruction issue -, 6 % of the blocks go through the list behind a gap: 84.7 against 82.2 us, the step 0.911-0.913 against
  // 0.917-0.939 ms with two passes in flight, 1.344-1.351 against 1.325-1.344 with one.  Not the default.
  const bool split = b->w4_split && b->win_cap > 64 && b->h_n_big64 > 0 && b->h_n_big64 * 8 < n_blk;
  {
    Scope* sc = b->time_all ? new Scope(b, "w4s_segment", 0) : nullptr;
    if (split) {
      v.w4_mode = 1;
      hipLaunchKernelGGL(w4s_segment<64>, dim3((unsigned)n_blk), dim3(64), 0, b->cur, v, (int64_t)0);
      v.w4_mode = 2;
      if (b->win_cap == 256) hipLaunchKernelGGL(w4s_segment<256>, dim3((unsigned)b->h_n_big64), dim3(64), 0, b->cur, v, (int64_t)0);
      else hipLaunchKernelGGL(w4s_segment<SNF_WIN_MAXCAP>, dim3((unsigned)b->h_n_big64), dim3(64), 0, b->cur, v, (int64_t)0);
      v.w4_mode = 0;
    }
    else if (b->win_cap == 64) hipLaunchKernelGGL(w4s_segment<64>, dim3((unsigned)n_blk), dim3(64), 0, b->cur, v, (int64_t)0);
    else if (b->win_cap == 256) hipLaunchKernelGGL(w4s_segment<256>, dim3((unsigned)n_blk), dim3(64), 0, b->cur, v, (int64_t)0);
    else hipLaunchKernelGGL(w4s_segment<SNF_WIN_MAXCAP>, dim3((unsigned)n_blk), dim3(64), 0, b->cur, v, (int64_t)0);
    delete sc;
    SNF_HIP(hipGetLastError());
  }
  if (v.chain_on) CHAIN(w5c_offsets, n_blk);
  else { FUSED(w5a_sums, n_blk); FUSED(w5b_offsets, n_blk); }
  LAUNCH_Q(w6t_emit, v, N, 0);
}

void run_call_candidates(snf_batch_impl* b) {
  SNF_TRACE("snf_batch_call_candidates (enqueue)");
  b->pass_idle = false;
  View& v = b->v;
  int T = v.T;
  const int64_t N = v.NS;   // positions behind the sort (the prefilter's count is known since the upload)
  b->reads_ready = true; b->cov_avg_ready = true; b->finalized = false;
  begin_pass_timing(b);
  pass_begins(b);
  b->chain_slot = -1;
  if (chain_gate_on() && !b->capturing) {
    DevicePacing& P = pacing_of(b);
    std::lock_guard<std::mutex> g(P.pace_mu);
    const unsigned long long my = P.chain_seq++;
    hipEvent_t& mine = P.chain_ev[my & 1]; hipEvent_t prev = P.chain_ev[(my + 1) & 1];
    if (!mine) SNF_HIP(hipEventCreateWithFlags(&mine, hipEventDisableTiming));
    b->chain_slot = (int)(my & 1);
    if (prev && P.passes_in_flight.load() > 1) SNF_HIP(hipStreamWaitEvent(b->stream, prev, 0));
  }
  // SNF_OUT_EXECUTE (set before this call): the names of the supporting reads are only written for the calls that pass QC,
  // once finalize knows them (a stage-0 fetch writes them all, late)
  v.rn_defer = ((v.out_mode & SNF_OUT_EXECUTE) && !v.cfg.no_qc && getenv("SNF_NO_RN_DEFER") == nullptr && getenv("SNF_NO_FUSE") == nullptr && N <= ((int64_t)1 << 25)) ? 1 : 0;
  b->rn_state = v.rn_defer ? 1 : 0;
  if (b->timeline) SNF_HIP(hipEventRecord(b->ev_base, b->stream));
  // fused chains: two-level tile sums cost O(N / 16384) loads per block, fine up to a few 10^7 elements; beyond that
  // (and in the emulation build) the plain device-wide scans are used
  enqueue_pass_init(b);
  if (v.wave_path && !b->fused) dzero(b, v.big_cnt, sizeof(uint32_t) * 3 * 64 * 16);   // (fused: z0_init)
  if (v.wave_path && !b->fused) dzero(b, v.d2cnt, sizeof(uint32_t) * 3 * 64 * 16);
  b->finalize_runs = 0;
  fork_mark(b);  // the read-preparation branch may start here, wherever it is enqueued below
  if (b->sched_readprep == 0) enqueue_read_prep(b);
  const bool front = v.front && b->fused && N > 0;
  if (front) enqueue_window_front(b); else enqueue_keys(b);
  if (N > 0) {
    if (!b->fused) {
      uint32_t* tails[] = {v.headflag, v.eligflag, v.fN, v.fL, v.runflag, v.clflag, v.rcflag, v.cdflag};
      for (auto p : tails) dzero(b, p + N, sizeof(uint32_t));
    }
    if (!front) {
    if (v.key32) prim_sort_pairs<uint32_t>(b, (uint32_t*)v.key_in, (uint32_t*)v.key_out, v.val_in, v.val_out, N, v.key_nbits + 1, "sort_lead_keys");
    else prim_sort_pairs<uint64_t>(b, v.key_in, v.key_out, v.val_in, v.val_out, N, v.key_nbits + 1, "sort_lead_keys");
    }
    if (b->fused) {
    // flag -> device-wide scan -> emit chains as "flags + tile sums" / "tile prefix + block scan + emit" kernel pairs
    // (snf_fused.h): 13 launches for stages A-C instead of 26 (each rocPRIM scan is an init kernel + a scan kernel)
    if (!front) {
    FUSED(a2k_heads, N);
    FUSED(a3k_bins, N);
    { Scope _s(b, "a4_binstats", N * 16); FUSED(a4k_binstats, N); }
    FUSED(a5k_leadflags, N);
    { Scope _s(b, "a6_scatter", N * 16); FUSED(a6k_scatter, N); }
    FUSED(a7k_seeds, N);
    }
    if (v.chain_on) { Scope _s(b, "b1_seedmetrics", N * 8); CHAIN(b12c_seedruns, N); }
    else {
    { Scope _s(b, "b1_seedmetrics", N * 8); FUSED(b1k_seedmetrics, N); }
    FUSED(b2k_runs, N);
    }
    LAUNCH(c1_mergeruns, v, N, N * 8);
    LAUNCH_Q(c2_validate, v, N, 0);
    LAUNCH_Q(c3_serial, v, 8 * (int64_t)T, 0);
    if (v.chain_on) CHAIN(c4c_clusters, N);
    else {
    FUSED(c4a_count, N);
    FUSED(c4k_clusters, N);
    }
    } else {
    LAUNCH_Q(a2_heads, v, N, N * 12);
    prim_exscan<uint32_t>(b, v.headflag, v.headscan, N + 1, "scan_bins");
    LAUNCH_Q(a3_bins, v, N, N * 8);
    LAUNCH(a4_binstats, v, N, N * 16);
    prim_exscan<uint32_t>(b, v.eligflag, v.eligscan, N + 1, "scan_seeds");
    LAUNCH_Q(a5_leadflags, v, N, N * 12);
    prim_exscan<uint32_t>(b, v.fN, v.pN, N + 1, "scan_leads");
    prim_exscan<uint32_t>(b, v.fL, v.pL, N + 1, "scan_leads_long");
    LAUNCH_Q(a6_scatter, v, N, N * 16);
    LAUNCH_Q(a7_seeds, v, N, N * 4);
    LAUNCH(b1_seedmetrics, v, N, N * 8);
    prim_exscan<uint32_t>(b, v.runflag, v.runscan, N + 1, "scan_runs");
    LAUNCH_Q(b2_runs, v, N, N * 4);
    LAUNCH(c1_mergeruns, v, N, N * 8);
    LAUNCH_Q(c2_validate, v, N, 0);
    LAUNCH_Q(c3_serial, v, 8 * (int64_t)T, 0);
    prim_exscan<uint32_t>(b, v.clflag, v.clscan, N + 1, "scan_clusters");
    LAUNCH_Q(c4_clusters, v, N, N * 4);
    dzero(b, v.rcflag, sizeof(uint32_t) * (N + 1));
    }
    if (b->sched_readprep == 3) fork_mark(b);   // mode 3: the read preparation may only start once stages A-C are through
    if (v.wave_path && b->d1_groups) {
      // merge_inner / resplit by cluster size (snf_wave_refine_g.h): eight clusters of <= 8 leads per wave, then d1w_refine - a wave
      // per cluster - for what that kernel handed on
      { Scope _s(b, "d1g_refine8", N * 36 / 3);
        hipLaunchKernelGGL(d1g_refine<8>, dim3(b->slots_d1w), dim3(64), 0, b->cur, v, (int64_t)0);
        SNF_HIP(hipGetLastError()); }
      { Scope _s(b, "d1w_refine", N * 36 - N * 36 / 3);
        v.d1_from_list = 1;
        hipLaunchKernelGGL(d1w_refine, dim3(b->slots_d1w), dim3(64), 0, b->cur, v, (int64_t)0);
        v.d1_from_list = 0;
        SNF_HIP(hipGetLastError()); }
    } else if (v.wave_path) {
      Scope _s(b, "d1w_refine", N * 36);
      hipLaunchKernelGGL(d1w_refine, dim3(b->slots_d1w), dim3(64), 0, b->cur, v, (int64_t)0);
      SNF_HIP(hipGetLastError());
    }
    if (!(v.wave_path && b->fused)) LAUNCH_Q(d1_refine, v, N, v.wave_path ? 0 : N * 36);   // (next to the wave kernels every item would return at once)
    b->skipped_big[0] = b->skipped_big[1] = b->skipped_big[2] = false;
    if (v.wave_path && b->have_big_hist && b->hist_big[0] == 0) b->skipped_big[0] = true;      // (same input: no cluster of more than 64 leads)
    else if (v.wave_path) {   // clusters of more than 64 leads, one wave each
      Scope _s(b, "x_big_refine", 0);
      hipLaunchKernelGGL(x_big<0>, dim3(b->slots_big), dim3(64), 0, b->cur, v, (int64_t)0);
      SNF_HIP(hipGetLastError());
    }
  }
  if (b->sched_readprep == 1 || b->sched_readprep == 3) enqueue_read_prep(b);  // while the long refine kernel keeps the main stream busy
  if (N > 0) {
    if (b->fused && v.chain_on) CHAIN(d1bc_rctable, N);
    else if (b->fused) {
      FUSED(d1a_count, N);
      FUSED(d1bk_rctable, N);
    } else {
      prim_exscan<uint32_t>(b, v.rcflag, v.rcscan, N + 1, "scan_refined");
      LAUNCH_Q(d1b_rctable, v, N, N * 4);
    }
    if (v.wave_path && b->d2_groups) {
      // call_from by cluster size (snf_wave_call_g.h): eight clusters of <= 8 leads per wave, then two of <= 32 from the list
      // the first kernel left, then d2w_call - a wave per cluster - for what the second handed on
      const bool ph = b->cfg.phase != 0;
      // (32 B per signature for the call stage, split by the share of the leads each kernel sees on a 30x genome - a third of
      //  them sit in refined clusters of at most eight leads)
      { Scope _s(b, "d2g_call8", N * 32 / 3);
        if (ph) hipLaunchKernelGGL((d2g_call<8, 4, true>), dim3(b->slots_d2w), dim3(64), 0, b->cur, v, (int64_t)0);
        else hipLaunchKernelGGL((d2g_call<8, 4, false>), dim3(b->slots_d2w), dim3(64), 0, b->cur, v, (int64_t)0);
        SNF_HIP(hipGetLastError()); }
      static const bool mid = getenv("SNF_D2_MID") != nullptr;   // A/B: clusters of 9..32 leads two per wave (measured slower than a wave each)
      if (mid) { Scope _s(b, "d2g_call32", 0);
        if (ph) hipLaunchKernelGGL((d2g_call<32, 4, true>), dim3(b->slots_d2w), dim3(64), 0, b->cur, v, (int64_t)0);
        else hipLaunchKernelGGL((d2g_call<32, 4, false>), dim3(b->slots_d2w), dim3(64), 0, b->cur, v, (int64_t)0);
        SNF_HIP(hipGetLastError()); }
      { Scope _s(b, "d2w_call", N * 32 - N * 32 / 3);
        v.d2_from_list = mid ? 2 : 1;
        hipLaunchKernelGGL(b->k_d2w, dim3(b->slots_d2w), dim3(64), 0, b->cur, v, (int64_t)0);
        v.d2_from_list = 0;
        SNF_HIP(hipGetLastError()); }
    } else if (v.wave_path) {
      Scope _s(b, "d2w_call", N * 32);
      hipLaunchKernelGGL(b->k_d2w, dim3(b->slots_d2w), dim3(64), 0, b->cur, v, (int64_t)0);
      SNF_HIP(hipGetLastError());
    }
    if (!(v.wave_path && b->fused)) LAUNCH_Q(d2_call, v, N, v.wave_path ? 0 : N * 32);
    if (v.wave_path && b->have_big_hist && b->hist_big[1] == 0) b->skipped_big[1] = true;
    else if (v.wave_path) {
      Scope _s(b, "x_big_call", 0);
      hipLaunchKernelGGL(x_big<1>, dim3(b->slots_big), dim3(64), 0, b->cur, v, (int64_t)0);
      SNF_HIP(hipGetLastError());
    }
    if (b->fused && v.chain_on) CHAIN(d3cc_compact, N);
    else if (b->fused) {
      FUSED(d3a_count, N);
      FUSED(d3ck_compact, N);
    } else {
      prim_exscan<uint32_t>(b, v.cdflag, v.cdscan, N + 1, "scan_calls");
      LAUNCH_Q(d3_compact, v, N, 0);
    }
  }
  // the number of calls is known here: publish the counters (pinned block) and let the host pick them up through
  // ev_counts.  Nothing the ALT chain of finalize needs is produced after this point, so the rest of the candidate
  // stage (sv ids, supporting read names, coverage annotation) continues on the side stream, behind the read preparation
  LAUNCH_Q(d3_taskoff, v, tail_threads(v), 0);
  SNF_HIP(hipEventRecord(b->ev_counts, b->stream));
  if (b->sched_readprep == 2) enqueue_read_prep(b);
  fork_mark(b);
  if (b->fused) {
    {  // sv ids + supporting read names: own stream (nothing on the coverage -> QC -> record copy chain waits for them
       // except the copy itself, through ev_rn)
      SNF_HIP(hipStreamWaitEvent(b->stream4, b->ev_fork, 0));
      hipStream_t prev = b->cur; b->cur = b->stream4;
      if (N > 0 && v.chain_on) { Scope _s(b, "d3_rnames", 0); CHAIN(d3src_svid_rnames, N); }
      else if (N > 0) {
        FUSED(d3sk_svid, N);
        { Scope _s(b, "d3_rnames", 0); FUSED(d3rk_rnames, N); }
      } else *b->h_rn_total = 0;
      SNF_HIP(hipEventRecord(b->ev_rn, b->stream4));
      b->cur = prev;
    }
    SideStream side(b);
    if (N > 0) launch_coverage(b, N);
  } else
  {
    SideStream side(b);
    if (N > 0) {
      LAUNCH_Q(d3_svid, v, N, 0);
      prim_exscan<uint32_t>(b, v.rnf, v.rnp, N + 1, "scan_rnames");
      LAUNCH(d3_rnames, v, N, 0);
    } else *b->h_rn_total = 0;
    SNF_HIP(hipEventRecord(b->ev_rn, b->cur));
    if (N > 0) launch_coverage(b, N);
  }
  b->res_current = false;
  v.out_valid = 0;
}

// everything enqueued on any of the batch's streams has completed and the pinned result block is current
void join_fourth(snf_batch_impl* b) {  // main stream waits for the fourth stream (sv ids, read names, their copy)
  SNF_HIP(hipEventRecord(b->ev_join4, b->stream4));
  SNF_HIP(hipStreamWaitEvent(b->stream, b->ev_join4, 0));
}
void full_sync(snf_batch_impl* b) {
  join_side(b);
  join_fourth(b);
  if (!b->res_current) { LAUNCH_Q(z1_results, b->v, tail_threads(b->v), 0); b->res_current = true; }
  dsync(b);
}

void ensure_cap(snf_batch_impl* b, int64_t need, int64_t& cap, void** p, size_t elem) {
  if (need <= cap && *p) return;
  if (*p) dfree_one(b, *p);
  cap = need + need / 4 + 64;
  *p = dalloc_own<uint8_t>(b, (size_t)cap * elem);
}

// ---- output stage (snf_stage_out.h): flags / scan / rank / records + read names on the side stream as soon as the scalar
// call fields are final, the ALT bytes behind the consensus kernels.  `late`: everything on the main stream (plain-scan path,
// and the redo after the rare fallbacks of the ALT stage)
// supporting read names that the candidate stage only sized (View::rn_defer): for the calls the output keeps, or for all
void enqueue_rnames_late(snf_batch_impl* b, bool all) {
  View& v = b->v;
  if (b->rn_state == 0 || (b->rn_state == 2 && !all) || v.NS <= 0) return;
  const int keep = v.rn_defer;
  v.rn_defer = all ? 2 : 1;
  { Scope _s(b, "d3_rnames_late", 0);
    unsigned grid = (unsigned)((v.NS / 4 + 255) / 256) + 1u; if (grid > 4096u) grid = 4096u;
    hipLaunchKernelGGL(d3lk_rnames_late, dim3(grid), dim3(256), 0, b->cur, v, (int64_t)0);
    SNF_HIP(hipGetLastError()); }
  v.rn_defer = keep;
  b->rn_state = all ? 0 : 2;
}

void enqueue_output_head(snf_batch_impl* b) {
  SNF_TRACE("F: output stage (filter, rank, records + read names)");
  View& v = b->v;
  const int64_t NS = v.NS;
  const bool rn_all = !((v.out_mode & SNF_OUT_EXECUTE) && !v.cfg.no_qc);
  // deferred names of the kept calls: written by f4w_emit itself, from the leads into the block (no pass over the candidates
  // and no intermediate copy in HBM); everything else (all names wanted, the thread form) takes the late kernel
  const bool rn_src = b->rn_state == 1 && !rn_all && b->fused && NS > 0 && NS <= ((int64_t)1 << 22) * 256 && getenv("SNF_NO_RN_FUSE") == nullptr;
  if (!rn_src) enqueue_rnames_late(b, rn_all);
  if (b->fused && NS <= ((int64_t)1 << 22) * 256) {
    const unsigned grid = (unsigned)((NS + 255) / 256) > 0u ? (unsigned)((NS + 255) / 256) : 1u;
    FUSED(f1k_outflags, NS > 0 ? NS : 1);
    FUSED(f2k_outscan, NS > 0 ? NS : 1);
    if ((v.out_mode & SNF_OUT_EXECUTE) && v.cfg.sort) {
      Scope _s(b, "f3_rank", 0);
      hipLaunchKernelGGL(f3k_rank, dim3(grid < 1024u ? grid : 1024u), dim3(256), 0, b->cur, v, (int64_t)0);
      SNF_HIP(hipGetLastError());
    }
    { Scope _s(b, "f4_emit", 0);
      v.rn_from_src = rn_src ? 1 : 0;
      static const unsigned f4cap = getenv("SNF_F4_GRID") && atoi(getenv("SNF_F4_GRID")) > 0 ? (unsigned)atoi(getenv("SNF_F4_GRID")) : 2048u;   // experiments
      hipLaunchKernelGGL(f4w_emit, dim3(grid < f4cap ? grid : f4cap), dim3(256), 0, b->cur, v, (int64_t)0);
      v.rn_from_src = 0;
      SNF_HIP(hipGetLastError()); }
  } else {
    const int64_t nc = NS;   // upper bound of the number of calls (the bodies stop at the device's count)
    LAUNCH_Q(f1_flags, v, nc + 1, 0);
    prim_exscan<uint32_t>(b, v.o_scan, v.pL, nc + 1, "scan_out");
    prim_exscan<int64_t>(b, v.sz_rd, v.sc_rd, nc + 1, "scan_out");
    LAUNCH_Q(f2_scan, v, nc + 1, 0);
    if ((v.out_mode & SNF_OUT_EXECUTE) && v.cfg.sort) LAUNCH_Q(f3_rank, v, nc, 0);
    LAUNCH_Q(f4_emit, v, nc, 0);
  }
}
// Thread-kernel form of the ALT stage (e4 / e5 / e6) and the ROWS instance need scratch sized by totals that only the device
// knows: the one place where finalize waits for the device.  Taken by the emulation build (always), by SNF_NO_WAVE, and - from
// the fetch, after the fact - when a call fits none of the LDS classes or a workgroup handed its call over (escape list).
void run_alt_fallback(snf_batch_impl* b) {
  SNF_TRACE("E: slow ALT kernels (ROWS / thread form)");
  View& v = b->v;
  d2h(b, b->h_cnt, v.cnt, sizeof(Counts));
  dsync(b);
  const Counts& c = *b->h_cnt;
  if (c.n_cons <= 0) return;
  const int64_t tab_total = c.tab_total, aln_total = c.aln_total, nreads = c.n_cons_reads;
  int64_t c1 = b->tab_cap, c2 = b->tab_cap, c3 = b->tab_cap;
  ensure_cap(b, tab_total, c1, (void**)&v.tab_key, sizeof(uint64_t));
  ensure_cap(b, tab_total, c2, (void**)&v.tab_pos, sizeof(int32_t));
  ensure_cap(b, tab_total, c3, (void**)&v.tab_state, sizeof(uint8_t));
  b->tab_cap = c1; v.tab_cap = c1;
  ensure_cap(b, aln_total, b->aln_cap, (void**)&v.aln, 1); v.aln_cap = b->aln_cap;
  int64_t r1 = b->cr_cap, r2 = b->cr_cap, r3 = b->cr_cap;
  ensure_cap(b, nreads, r1, (void**)&v.aln_kept, 1);
  ensure_cap(b, nreads, r2, (void**)&v.cr_call, sizeof(int32_t));
  ensure_cap(b, nreads, r3, (void**)&v.cr_read, sizeof(int32_t));
  b->cr_cap = r1;
  const bool threads = !v.wave_path || c.n_cons_fallback > 0;
  if (threads) LAUNCH_Q(e4_anchor, v, c.n_cons, v.wave_path ? 0 : c.tab_total * 13);
  if (v.wave_path && c.n_cls[7] > 0) {   // work list 7: calls beyond the LDS vote counters and what SMALL / LARGE handed over at run time
    Scope _s(b, "e45w_consensus_rows", 0);
    const int64_t n_rows = (int64_t)c.n_cls[7];
    hipLaunchKernelGGL((K_CONS_ROWS), dim3((unsigned)(n_rows < 16384 ? n_rows : 16384)), dim3(256), 0, b->cur, v, (int64_t)0);
    SNF_HIP(hipGetLastError());
  }
  if (threads) {
    LAUNCH_Q(e5_align, v, c.n_cons_reads, v.wave_path ? 0 : c.aln_total * 2);
    LAUNCH(e6_vote, v, c.alt_total, c.aln_total + 2 * c.alt_total);
  }
}

// SMALL / LARGE / verbatim-copy kernels of the ALT stage, one launch per class.  The grids are upper bounds (exact when the
// caller knows the class counts): workgroups behind the end of a class's list return at once, and the kernels stride when
// a list is longer than the grid.
void enqueue_consensus_wave(snf_batch_impl* b, int64_t g_small, int64_t g_large, int64_t g_copy) {
  SNF_TRACE("E4/E5: INS consensus (SMALL / LARGE / verbatim)");
  View& v = b->v;
  const bool serial = getenv("SNF_SERIAL") != nullptr;  // dev: every ALT kernel alone on the device (isolated timings)
  // Order of the two consensus classes.  0: LARGE (main stream) NEXT TO SMALL (third stream) - alone on the device SMALL fills the
  // tail of LARGE's unequal calls (one batch in flight: 1.466 ms per step against 1.494 for order 2).  2: LARGE BEHIND SMALL on one
  // stream - next to each other a LARGE workgroup (70 KB of LDS, 4 x 256 VGPRs on one CU at once) only finds room when SMALL's queue
  // of one-wave workgroups runs dry: LARGE spans 0.51 ms in place against 0.22 behind SMALL (SMALL: 0.27 / 0.17), and with a second
  // pass in flight that pass fills the tails instead (two in flight: 1.238 against 1.242).  So: 2 when another pass is in flight on
  // the device, 0 otherwise; SNF_CONS_ORDER forces one (1 = SMALL behind LARGE: slower than both).
  const int order_env = getenv("SNF_CONS_ORDER") ? atoi(getenv("SNF_CONS_ORDER")) : -1;
  const int order = order_env >= 0 ? order_env : (pacing_of(b).passes_in_flight.load() > 1 ? 2 : 0);
  auto launch_large = [&]() {
    Scope _s(b, "e45w_consensus_large", 0, true);
    const dim3 gl((unsigned)(g_large < b->slots_cons_l ? g_large : b->slots_cons_l));
    if (b->cons_large_nw == 16) hipLaunchKernelGGL((K_CONS_LARGE_16W), gl, dim3(1024), 0, b->cur, v, (int64_t)0);
    else if (b->cons_large_nw == 8) hipLaunchKernelGGL((K_CONS_LARGE_8W), gl, dim3(512), 0, b->cur, v, (int64_t)0);
    else hipLaunchKernelGGL((K_CONS_LARGE), gl, dim3(256), 0, b->cur, v, (int64_t)0);
    SNF_HIP(hipGetLastError());
  };
  auto launch_small = [&]() {
    Scope _s(b, "e45w_consensus_small", 0);
    const dim3 gs((unsigned)(g_small < b->slots_cons_s ? g_small : b->slots_cons_s));
    if (b->cons_nw == 1) hipLaunchKernelGGL((K_CONS_SMALL_1W), dim3((unsigned)(g_small < b->slots_cons_s1 ? g_small : b->slots_cons_s1)), dim3(64), 0, b->cur, v, (int64_t)0);
    else if (b->occ_s >= 8) hipLaunchKernelGGL((K_CONS_SMALL(8)), gs, dim3(256), 0, b->cur, v, (int64_t)0);
    else if (b->occ_s == 6) hipLaunchKernelGGL((K_CONS_SMALL(6)), gs, dim3(256), 0, b->cur, v, (int64_t)0);
    else hipLaunchKernelGGL((K_CONS_SMALL(5)), gs, dim3(256), 0, b->cur, v, (int64_t)0);
    SNF_HIP(hipGetLastError());
  };
  auto launch_copy = [&]() {    // verbatim ALTs: short; on the main stream ahead of SMALL
    Scope _s(b, "e4c_copy", 0);
    hipLaunchKernelGGL(e4c_copy, dim3((unsigned)(g_copy < 32768 ? g_copy : 32768)), dim3(64), 0, b->cur, v, (int64_t)0);
    SNF_HIP(hipGetLastError());
  };
  auto on_stream3 = [&](auto&& f) { hipStream_t prev = b->cur; b->cur = b->stream3; f(); b->cur = prev; };
  // LARGE - the longer chain - stays on the main stream, right behind the kernel that built its work lists: a kernel behind a
  // cross-stream event starts 30-45 us after the event's kernel has ended, one on the same stream 5 us after
  if (serial) SNF_HIP(hipDeviceSynchronize());
  if (order == 2) {            // one stream: verbatim copies, SMALL, LARGE
    launch_copy();
    launch_small();
    if (serial) SNF_HIP(hipDeviceSynchronize());
    launch_large();
    return;
  }
  SNF_HIP(hipEventRecord(b->ev_fork3, b->stream));
  SNF_HIP(hipStreamWaitEvent(b->stream3, b->ev_fork3, 0));
  launch_large();
  if (serial) SNF_HIP(hipDeviceSynchronize());
  if (order == 1) launch_small();
  on_stream3(launch_copy);
  if (order != 1) on_stream3(launch_small);
  SNF_HIP(hipEventRecord(b->ev_join3, b->stream3));
  SNF_HIP(hipStreamWaitEvent(b->stream, b->ev_join3, 0));
}

// staged result with SNF_STAGE_COPY=kernel: a small copy kernel behind the producers of the block / the ALT section, on their stream
// (64 workgroups: enough stores in flight for the PCIe link, a handful of wave slots)
void stage_copy_block(snf_batch_impl* b) {
  if (!b->staged_kernel || !b->v.stage_out_pin) return;
  Scope _s(b, "d2h_block", 0);
  hipLaunchKernelGGL(z2_stage_copy, dim3(64), dim3(256), 0, b->cur, b->v, (int64_t)0);
  SNF_HIP(hipGetLastError());
  b->stage_block_copied = true;
}
void stage_copy_alt(snf_batch_impl* b) {
  if (!b->staged_kernel || !b->v.stage_alt_pin) return;
  Scope _s(b, "d2h_alt", 0);
  hipLaunchKernelGGL(z2_stage_copy, dim3(64), dim3(256), 0, b->cur, b->v, (int64_t)1);
  SNF_HIP(hipGetLastError());
  b->stage_alt_copied = true;
}

void run_finalize(snf_batch_impl* b) {
  SNF_TRACE("snf_batch_finalize (enqueue)");
  b->pass_idle = false;
  View& v = b->v;
  const int64_t NS = v.NS;
  b->finalized = true;
  b->res_current = false;
  if (b->chain_slot >= 0 && !b->capturing) {      // the chain of this pass ends here: the next pass of the device may start its own
    DevicePacing& P = pacing_of(b);
    std::lock_guard<std::mutex> g(P.pace_mu);
    if (P.chain_ev[b->chain_slot]) SNF_HIP(hipEventRecord(P.chain_ev[b->chain_slot], b->stream));
    b->chain_slot = -1;
  }
  // Nothing below waits for the device: grids cover upper bounds derived from NS (the kernels read the real counts in HBM
  // and stride or return), the ALT bytes go to an HBM pool sized at upload, and the fetch is the one host wait of the pass.
  {  // pinned block for the result: sized from the input, grown by the fetch when a result did not fit
    const size_t want = (size_t)((v.out_mode & SNF_OUT_EXECUTE) ? 8 : 16) * (size_t)(v.N > 0 ? v.N : 1) + ((size_t)1 << 20);
    if (!(v.out_mode & SNF_OUT_DEVICE) && !b->hb_out.external && b->hb_out.cap < want) b->hb_out.ensure(want);
    // Where the kernels store the result.  Alone on the device: straight into the pinned buffers (zero-copy stores over PCIe - the
    // shortest pass).  With ANOTHER pass in flight: into HBM, and two copies behind the kernels take it to the host.  The stores of a
    // pass are ~19 MB at ~40 GB/s = the last 0.45 ms of its kernels, during which waves that wait for PCIe hold the wave slots, LDS
    // and registers the other pass's kernels need; staged, those kernels end at HBM speed and the copy engines' traffic overlaps the
    // other pass's compute (same box, two in flight: 0.945 ms per step against 1.17; one in flight 1.48 against 1.36 - hence the
    // switch).  SNF_STAGE_OUT=1 / 0 force either.  (Not while a pass is captured: a replayed graph keeps the direct stores.)
    const int stage_env = getenv("SNF_STAGE_OUT") ? atoi(getenv("SNF_STAGE_OUT")) : -1;
    const bool stage = !(v.out_mode & SNF_OUT_DEVICE) && !b->capturing && (stage_env == 1 || (stage_env < 0 && passes_overlap(b)));
    const char* copy_env = getenv("SNF_STAGE_COPY");
    b->staged = stage; b->staged_kernel = stage && copy_env && strcmp(copy_env, "kernel") == 0;
    b->stage_block_copied = b->stage_alt_copied = false;
    const bool out_hbm = (v.out_mode & SNF_OUT_DEVICE) || stage;
    v.out_pin = out_hbm ? nullptr : (uint8_t*)b->hb_out.p;
    v.out_pin_cap = out_hbm ? 0 : (int64_t)b->hb_out.cap;
    v.stage_out_pin = b->staged_kernel ? (uint8_t*)b->hb_out.p : nullptr; v.stage_out_cap = b->staged_kernel ? (int64_t)b->hb_out.cap : 0;
    // ALT section: an eighth of the input sequence bytes (a 30x genome needs a twentieth); the fetch grows it when a pass overflowed into HBM
    const size_t want_alt = (size_t)(v.pool_len / 8) + ((size_t)1 << 20);
    if (!(v.out_mode & SNF_OUT_DEVICE) && !b->hb_alt.external && b->hb_alt.cap < want_alt) b->hb_alt.ensure(want_alt);
    const bool alt_hbm = getenv("SNF_ALT_HBM") != nullptr || stage;   // (SNF_ALT_HBM: measurement - only the ALT bytes into HBM, copied at fetch)
    v.alt_pin = ((v.out_mode & SNF_OUT_DEVICE) || alt_hbm) ? nullptr : (uint8_t*)b->hb_alt.p;
    v.alt_pin_cap = ((v.out_mode & SNF_OUT_DEVICE) || alt_hbm) ? 0 : (int64_t)b->hb_alt.cap;
    v.stage_alt_pin = b->staged_kernel ? (uint8_t*)b->hb_alt.p : nullptr; v.stage_alt_cap = b->staged_kernel ? (int64_t)b->hb_alt.cap : 0;
  }
  v.out_valid = 1;
  // Launch sizes of the data-dependent kernels.  Grids much larger than the work flood the dispatcher with empty workgroups
  // (measured: every co-running kernel of the device slows down), grids smaller make the kernels stride.  A handle that
  // has finalized before knows its sizes (same input); the first pass waits once for the counters d3_taskoff published.
  int64_t n_calls_hint = b->hist_calls;
  if (!b->have_hist && NS > 0) {
    SNF_HIP(hipEventSynchronize(b->ev_counts));
    n_calls_hint = b->h_cnt->n_calls;
  }
  if (NS > 0) {
  {  // QC / phasing / genotyping only touch the scalar call fields: side stream (behind d4_coverage, whose
     // annotations they read), overlapped with the consensus chain
    SideStream side(b);
    if (v.wave_path && b->finalize_runs++ > 0) dzero(b, v.big_cnt + 2 * 64 * 16, sizeof(uint32_t) * 64 * 16);   // (the first finalize finds the list empty; a repeated one empties it again)
    if (v.wave_path) {
      Scope _s(b, "e1w_finalize", 0);
      // one workgroup (= wave) per batch of e1_batch calls, dispatched by the hardware (the kernel strides when there are
      // more calls than the grid covers, workgroups behind the last call return at once)
      int64_t grid = (n_calls_hint + v.e1_batch - 1) / v.e1_batch + 1;
      if (grid > (1 << 20)) grid = 1 << 20;
      hipLaunchKernelGGL(b->k_e1w, dim3((unsigned)grid), dim3(64), 0, b->cur, v, (int64_t)0);
      SNF_HIP(hipGetLastError());
    }
    if (!v.wave_path) LAUNCH_Q(e1_finalize, v, NS, 0);
    if (v.wave_path && b->have_big_hist && b->hist_big[2] == 0 && b->finalize_runs == 1) b->skipped_big[2] = true;
    else if (v.wave_path) {
      Scope _s(b, "x_big_finalize", 0);
      hipLaunchKernelGGL(x_big<2>, dim3(b->slots_big), dim3(64), 0, b->cur, v, (int64_t)0);
      SNF_HIP(hipGetLastError());
    }
  }
  const bool fast_alt = v.wave_path && b->fused && NS <= ((int64_t)1 << 22) * 256;
  if (fast_alt)
  {  // E2 sizes -> offsets -> E3 work items in two launches (snf_fused.h) instead of a size kernel, five scans and E3
    const unsigned grid = (unsigned)((NS + 255) / 256);
    if (b->time_all) { Scope _s(b, "e2a_sizes", 0); hipLaunchKernelGGL(e2a_sizes, dim3(grid), dim3(256), 0, b->cur, v, (int64_t)0); }
    else hipLaunchKernelGGL(e2a_sizes, dim3(grid), dim3(256), 0, b->cur, v, (int64_t)0);
    SNF_HIP(hipGetLastError());
    if (b->time_all) { Scope _s(b, "e3b_offsets", 0); hipLaunchKernelGGL(e3b_offsets, dim3(grid), dim3(256), 0, b->cur, v, (int64_t)0); }
    else hipLaunchKernelGGL(e3b_offsets, dim3(grid), dim3(256), 0, b->cur, v, (int64_t)0);
    SNF_HIP(hipGetLastError());
    SNF_HIP(hipEventRecord(b->ev_e3, b->stream));
    {  // records + read names leave as soon as e1 (side stream), the sv ids / read names (fourth stream) and the ALT
       // lengths (e2a, main) are there - next to the consensus kernels
      SideStream side(b);
      SNF_HIP(hipStreamWaitEvent(b->stream2, b->ev_e3, 0));
      SNF_HIP(hipStreamWaitEvent(b->stream2, b->ev_rn, 0));
      enqueue_output_head(b);
      stage_copy_block(b);      // (staged result: the block leaves behind f4w_emit, next to the consensus kernels)
    }
    if (!b->have_hist) {   // first finalize of this handle: the class sizes e3b has just written, one host wait
      d2h(b, b->h_cnt, v.cnt, sizeof(Counts));
      dsync(b);
      const Counts& c = *b->h_cnt;
      b->hist_small = (int64_t)c.n_cls[1]; b->hist_large = (int64_t)(c.n_cls[2] + c.n_cls[3] + c.n_cls[4] + c.n_cls[5]); b->hist_copy = (int64_t)c.n_cls[0];
      b->hist_calls = c.n_calls; b->have_hist = true;
    }
    enqueue_consensus_wave(b, b->hist_small + 1, b->hist_large + 1, b->hist_copy + 1);
    stage_copy_alt(b);          // (staged result: the ALT section leaves behind the last ALT kernel)
    join_side(b);
    join_fourth(b);
  }
  else
  {
    // plain-scan path (emulation build, SNF_NO_FUSE, SNF_NO_WAVE): sized by the number of calls, one host wait
    d2h(b, b->h_cnt, v.cnt, sizeof(Counts));
    dsync(b);
    const int64_t nc = b->h_cnt->n_calls;
    dzero(b, v.stripes, sizeof(unsigned long long) * 4 * 64 * 16);
    LAUNCH(e2_best, v, nc + 1, 0);
    prim_exscan<uint32_t>(b, v.fN, v.pN, nc + 1, "scan_alt");
    prim_exscan<uint32_t>(b, v.fL, v.pL, nc + 1, "scan_cons");
    prim_exscan<int64_t>(b, v.sz_tab, v.sc_tab, nc + 1, "scan_cons_sizes");
    prim_exscan<int64_t>(b, v.sz_aln, v.sc_aln, nc + 1, "scan_cons_sizes");
    prim_exscan<int64_t>(b, v.sz_rd, v.sc_rd, nc + 1, "scan_cons_sizes");
    LAUNCH_Q(e3_conslist, v, nc + 1, 0);
    if (v.wave_path) {
      d2h(b, b->h_cnt, v.cnt, sizeof(Counts));
      dsync(b);
      const Counts& c = *b->h_cnt;
      enqueue_consensus_wave(b, (int64_t)c.n_cls[1] + 1, (int64_t)(c.n_cls[2] + c.n_cls[3] + c.n_cls[4] + c.n_cls[5]) + 1, (int64_t)c.n_cls[0] + 1);
    }
    run_alt_fallback(b);     // (host wait: scratch sizes; thread kernels, and ROWS for work list 7, complete only now)
    join_side(b);
    join_fourth(b);
    enqueue_output_head(b);
  }
  } else {
    join_side(b);
    join_fourth(b);
    enqueue_output_head(b);   // (no positions behind the sort: an empty block)
  }
  LAUNCH_Q(z1_results, v, tail_threads(v), 0);
  b->res_current = true;
}

// ---- one pass = Task.call_candidates + Task.finalize_candidates back to back (CallTask.execute, parallel.py:264-266), as a HIP graph
// once the launch sizes of this handle are known.  The pass is ~40 dependent launches on four streams; replayed from a graph the
// host enqueues it with one call and the device-side launch-to-launch gaps shrink.  Everything a pass needs from the host is
// constant for a handle (same input): grids, pointers, modes - except the memory the result lands in and the output mode, which
// key the graph.  Kernels take the pass-dependent state (counters, chain tags, window cursors) from HBM.
bool pass_graph_ok(snf_batch_impl* b) {
  const View& v = b->v;
  if (b->graph_mode == 1 && pacing_of(b).passes_in_flight.load() > 1) return false;      // (this batch itself is counted)
  return b->graph_mode && !b->graph_failed && b->have_hist && b->fused && v.front && v.wave_path && !b->timeline && !b->time_all &&
         v.NS > 0 && !b->readprep_each_pass && getenv("SNF_SERIAL") == nullptr;
}
void run_pass(snf_batch_impl* b) {
  View& v = b->v;
  pass_begins(b);
  if (!pass_graph_ok(b)) { run_call_candidates(b); run_finalize(b); return; }
  // (the pinned blocks of the result are chosen by run_finalize from b->hb_out / hb_alt: stable once they have been sized by a first pass)
  const auto key = std::make_tuple((void*)b->hb_out.p, (void*)b->hb_alt.p, v.out_mode);
  auto& g = b->graphs[key];
  if (g.exec && (g.cap_out != b->hb_out.cap || g.cap_alt != b->hb_alt.cap)) {      // the same address with another size: a new configuration
    (void)hipGraphExecDestroy(g.exec); (void)hipGraphDestroy(g.graph); g = snf_batch_impl::PassGraph{};
  }
  g.passes++;
  // (a pass that carries the event brackets - every time_every-th of the handle - runs eagerly: a replayed graph has no events)
  const bool sampled = b->timing && (b->time_all || b->time_every == 1 || (b->time_every > 1 && b->pass_count % (uint64_t)b->time_every == 0));
  if (g.exec && !sampled) {
    b->pass_idle = false;
    reset_timing(b); b->pass_count++;
    SNF_HIP(hipGraphLaunch(g.exec, b->stream));
    // host-side state the two calls leave behind
    b->reads_ready = true; b->cov_avg_ready = true; b->finalized = true; b->finalize_runs = 1; b->res_current = true;
    b->rn_state = g.rn_state; v.rn_defer = g.rn_defer; v.out_valid = 1;
    return;
  }
  if (g.exec || g.passes < 2) { run_call_candidates(b); run_finalize(b); return; }    // eager: first pass of the configuration / timing pass
  // capture
  const bool tm = b->timing;
  b->timing = false;
  hipGraph_t graph = nullptr;
  bool ok = hipStreamBeginCapture(b->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
  if (ok) {
    b->capturing = true;
    // (an exception while capturing - a call that is illegal in a capture, a host wait only this path reaches - does not mean the
    // pass cannot run: the capture is ended, the handle stays eager from here on and the pass is enqueued the plain way below)
    // The message of what was swallowed is kept (SNF_PROF prints it; a genuine failure surfaces again, with its own message, from the eager
    // run below) and the pass counter the capture attempt advanced is put back, so that the sampled-timing cadence does not shift.
    const uint64_t count_before = b->pass_count;
    try { run_call_candidates(b); run_finalize(b); }
    catch (const std::exception& e) {
      (void)hipStreamEndCapture(b->stream, &graph); if (graph) (void)hipGraphDestroy(graph); graph = nullptr; ok = false; (void)hipGetLastError();
      b->pass_count = count_before;
      if (v.prof) fprintf(stderr, "[SNF_PROF] pass graph: capture gave up (%s); the pass runs eagerly\n", e.what());
    }
    catch (...) {
      (void)hipStreamEndCapture(b->stream, &graph); if (graph) (void)hipGraphDestroy(graph); graph = nullptr; ok = false; (void)hipGetLastError();
      b->pass_count = count_before;
    }
    b->capturing = false;
    if (ok) ok = hipStreamEndCapture(b->stream, &graph) == hipSuccess && graph != nullptr;
  }
  b->timing = tm;
  hipGraphExec_t exec = nullptr;
  if (ok) ok = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess && exec != nullptr;
  if (!ok) {
    (void)hipGetLastError();
    if (graph) (void)hipGraphDestroy(graph);
    b->graph_failed = true;
    if (v.prof) fprintf(stderr, "[SNF_PROF] pass graph: capture failed, passes stay eager\n");
    run_call_candidates(b); run_finalize(b);
    return;
  }
  g.graph = graph; g.exec = exec; g.rn_state = b->rn_state; g.rn_defer = v.rn_defer; g.cap_out = b->hb_out.cap; g.cap_alt = b->hb_alt.cap;
  if (v.prof) { size_t nn = 0; (void)hipGraphGetNodes(graph, nullptr, &nn); fprintf(stderr, "[SNF_PROF] pass graph: captured (%zu nodes)\n", nn); }
  SNF_HIP(hipGraphLaunch(g.exec, b->stream));
}
void destroy_graphs(snf_batch_impl* b) {
  for (auto& kv : b->graphs) { if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec); if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph); }
  b->graphs.clear();
}

void collect_timings(snf_batch_impl* b) {
  b->timings.clear();
#ifdef SNF_ITRACE
  if (getenv("SNF_PROF") && b->v.itrace) {      // per kernel slot: the workgroups' starts and durations of the pass that just ended
    static const char* nm[SNF_IT_SLOTS] = {"w1_hist", "w3_scatter", "w4s_segment", "w6t_emit", "d1g_refine", "d1w_refine", "d2g_call", "d2w_call", "e3b_offsets",
                                           "e1w_finalize", "cons SMALL", "cons LARGE", "d4s_coverage", "f4w_emit", "cons ROWS", ""};
    std::vector<uint32_t> h((size_t)SNF_IT_SLOTS * SNF_IT_CAP * 2);
    (void)hipMemcpy(h.data(), b->v.itrace, h.size() * 4, hipMemcpyDeviceToHost);
    (void)hipMemset(b->v.itrace, 0, h.size() * 4);
    for (int k = 0; k < SNF_IT_SLOTS; k++) {
      struct E { uint32_t t0, dur, wg; };
      std::vector<E> es;
      for (int64_t i = 0; i < SNF_IT_CAP; i++) { const uint32_t* o = &h[2 * ((size_t)k * SNF_IT_CAP + i)]; if (o[1] & 0x80000000u) es.push_back({o[0], o[1] & 0x7fffffffu, (uint32_t)i}); }
      if (es.empty()) continue;
      uint32_t base = es[0].t0;
      for (auto& e : es) if ((int32_t)(e.t0 - base) < 0) base = e.t0;
      uint32_t span = 0; double sum = 0;
      for (auto& e : es) { e.t0 -= base; span = std::max(span, e.t0 + e.dur); sum += e.dur; }
      std::vector<uint32_t> d; for (auto& e : es) d.push_back(e.dur);
      std::sort(d.begin(), d.end());
      auto pc = [&](double q) { return d[(size_t)(q * (d.size() - 1))] * 0.01; };
      fprintf(stderr, "[SNF_ITRACE] %-13s %6zu workgroups, span %7.1f us, sum %9.1f us = %6.0f in flight on average; duration us p50 %.1f p90 %.1f p99 %.1f max %.1f; in flight at 12 points:",
              nm[k], es.size(), span * 0.01, sum * 0.01, sum / (span ? span : 1), pc(0.5), pc(0.9), pc(0.99), pc(1.0));
      for (int q = 0; q < 12; q++) { const uint32_t t = (uint32_t)((uint64_t)span * (2 * q + 1) / 24); int act = 0; for (auto& e : es) act += e.t0 <= t && t < e.t0 + e.dur; fprintf(stderr, " %d", act); }
      std::sort(es.begin(), es.end(), [](const E& a, const E& c) { return a.t0 + a.dur > c.t0 + c.dur; });
      fprintf(stderr, "; last to end:");
      for (size_t q = 0; q < es.size() && q < 4; q++) fprintf(stderr, " wg %u +%.1f for %.1f", es[q].wg, es[q].t0 * 0.01, es[q].dur * 0.01);
      fprintf(stderr, "\n");
    }
  }
#endif
#ifdef SNF_WG_TRACE
  if (getenv("SNF_PROF") && b->v.wgtrace) {
    const int64_t n = 1 << 19;
    std::vector<unsigned long long> h((size_t)n * 6);
    (void)hipMemcpy(h.data(), b->v.wgtrace, h.size() * 8, hipMemcpyDeviceToHost);
    (void)hipMemset(b->v.wgtrace, 0, h.size() * 8);
    bool any_wg = false;
    for (int64_t i = 0; i < n && !any_wg; i++) any_wg = h[2 * i + 1] != 0;
    if (const char* path = any_wg ? getenv("SNF_WG_TRACE_FILE") : nullptr) {   // every workgroup: class, start, duration (100 MHz ticks), L, others, phases, HW_ID, XCC_ID
      if (FILE* f = fopen(path, "w")) {
        for (int64_t i = 0; i < n; i++) { const unsigned long long m = h[2 * i + 1]; if (!m) continue;
          fprintf(f, "%d %llu %llu %d %d %llu %llu %llu\n", (int)((m >> 28) & 15), h[2 * i], m >> 32, (int)(m & 0xffff), (int)((m >> 16) & 0xff), h[2 * (i + n)], h[2 * (i + n) + 1], h[2 * (i + 2 * n)]); }
        fclose(f);
      }
    }
    for (int cls = 1; cls <= 2; cls++) {
      struct E { unsigned long long t0, dur; int L, no; unsigned long long ph, rp; };
      std::vector<E> es;
      for (int64_t i = 0; i < n; i++) { const unsigned long long m = h[2 * i + 1]; if (m && (int)((m >> 28) & 15) == cls) es.push_back({h[2 * i], m >> 32, (int)(m & 0xffff), (int)((m >> 16) & 0xff), h[2 * (i + n)], h[2 * (i + n) + 1]}); }
      if (es.empty()) continue;
      unsigned long long t_min = ~0ull, t_max = 0, sum = 0;
      for (auto& e : es) { t_min = std::min(t_min, e.t0); t_max = std::max(t_max, e.t0 + e.dur); sum += e.dur; }
      std::sort(es.begin(), es.end(), [](const E& a, const E& c) { return a.dur > c.dur; });
      fprintf(stderr, "[SNF_WG_TRACE] %s: %zu workgroups, span %.1f us, sum of durations %.1f us (mean %.2f), longest:\n", cls == 1 ? "SMALL" : "LARGE", es.size(),
              (t_max - t_min) * 0.01, sum * 0.01, sum * 0.01 / es.size());
      for (size_t k = 0; k < es.size() && k < 8; k++)
        fprintf(stderr, "[SNF_WG_TRACE]    start +%.1f us, %.1f us, L %d, others %d | wave 0: setup %.1f reads %.1f barrier %.1f vote+store %.1f | reads: probes %.1f scan(+filter) %.1f segments %.1f geom+prefetch %.1f votes %.1f\n", (es[k].t0 - t_min) * 0.01, es[k].dur * 0.01, es[k].L, es[k].no,
          (es[k].ph & 0xffff) * 0.01, ((es[k].ph >> 16) & 0xffff) * 0.01, ((es[k].ph >> 32) & 0xffff) * 0.01, (es[k].ph >> 48) * 0.01,
          (es[k].rp & 0xfff) * 0.1, ((es[k].rp >> 12) & 0xfff) * 0.1, ((es[k].rp >> 24) & 0xfff) * 0.1, ((es[k].rp >> 36) & 0xfff) * 0.1, ((es[k].rp >> 48) & 0xfff) * 0.1);
      { std::vector<E> byt = es; std::sort(byt.begin(), byt.end(), [](const E& a, const E& c) { return a.t0 < c.t0; });
        for (size_t k = 0; k < byt.size(); k += byt.size() / 10 + 1) fprintf(stderr, "[SNF_WG_TRACE]    #%zu by start: +%.1f us, %.1f us, L %d, others %d | setup %.1f reads %.1f barrier %.1f vote+store %.1f\n", k, (byt[k].t0 - t_min) * 0.01, byt[k].dur * 0.01, byt[k].L, byt[k].no,
          (byt[k].ph & 0xffff) * 0.01, ((byt[k].ph >> 16) & 0xffff) * 0.01, ((byt[k].ph >> 32) & 0xffff) * 0.01, (byt[k].ph >> 48) * 0.01); }
      const int NB = 12; std::vector<int> act(NB, 0), fin(NB, 0);
      for (auto& e : es) for (int k = 0; k < NB; k++) { const unsigned long long t = t_min + (t_max - t_min) * (2 * k + 1) / (2 * NB); if (e.t0 <= t && t < e.t0 + e.dur) act[k]++; }
      fprintf(stderr, "[SNF_WG_TRACE]    workgroups in flight at 12 points of the span:");
      for (int k = 0; k < NB; k++) fprintf(stderr, " %d", act[k]);
      fprintf(stderr, "\n");
    }
  }
#endif
#ifdef SNF_C1_PROFILE
  if (getenv("SNF_PROF")) { const unsigned long long* c = b->h_cnt->c1p; const double w = (double)(c[8] ? c[8] : 1);
    fprintf(stderr, "[SNF_C1_PROFILE] c1_mergeruns, mean us per wave (%llu waves): first node %.2f | next node %.2f criterion %.2f | merge: stores %.2f metrics %.2f movement %.2f | loop exit %.2f tail %.2f; longest wave %.1f us\n",
            c[8], c[0] * 0.01 / w, c[1] * 0.01 / w, c[2] * 0.01 / w, c[3] * 0.01 / w, c[4] * 0.01 / w, c[5] * 0.01 / w, c[6] * 0.01 / w, c[7] * 0.01 / w, c[9] * 0.01); }
#endif
#ifdef SNF_CONS_PROFILE
  if (getenv("SNF_PROF")) {
    static const char* nm[9] = {"setup+table", "kmers+probes", "chain", "segments", "run filter", "votes", "barrier+vote+store", "longest workgroup", "workgroups"};
    for (int c = 0; c < 2; c++)
      for (int k = 0; k < 9; k++) if (c * 16 + k < 24) fprintf(stderr, "[SNF_CONS_PROFILE] %s %-20s %llu\n", c ? "LARGE" : "SMALL", nm[k], b->h_cnt->dbg[c * 16 + k]);
    static const char* dn[7] = {"loads + svlen sort", "names sort + a1", "ref_start sort + stdevs", "sums + record", "BND block", "INS best lead", "aggregates"};
    for (int k = 0; k < 7; k++) fprintf(stderr, "[SNF_CONS_PROFILE] d2g8 %-26s %llu\n", dn[k], b->h_cnt->dbg[9 + k]);
    static const char* rn[6] = {"load + first appearance", "rank + permute", "gathers + fuse scans", "reserve + copy + compact", "F stores", "resplit + next"};
    for (int k = 0; k < 6; k++) fprintf(stderr, "[SNF_CONS_PROFILE] d1w %-26s %llu\n", rn[k], b->h_cnt->dbg[24 + k]);
  }
#endif
  for (size_t i = 0; i < b->ev_used; i++) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, b->evs[i].a, b->evs[i].b) != hipSuccess) ms = -1;
    if (b->timeline) {
      float off = 0;
      if (hipEventElapsedTime(&off, b->ev_base, b->evs[i].a) != hipSuccess) off = -1;
      fprintf(stderr, "[SNF_TIMELINE] %9.1f %8.1f  %s\n", off * 1e3, ms * 1e3, b->evs[i].name);
    }
    bool found = false;
    for (auto& t : b->timings)
      if (strcmp(t.name, b->evs[i].name) == 0) { t.ms += ms; t.bytes += b->evs[i].bytes; t.launches++; found = true; break; }
    if (!found) b->timings.push_back({b->evs[i].name, ms, b->evs[i].bytes, 1});
  }
  for (auto& t : b->timings) {  // the consensus kernels count their own algorithmic bytes (read back with the counters)
    if (strcmp(t.name, "e45w_consensus_small") == 0) t.bytes = (int64_t)b->h_cnt->cons_bytes[1];
    if (strcmp(t.name, "e45w_consensus_large") == 0) t.bytes = (int64_t)b->h_cnt->cons_bytes[2];
    if (strcmp(t.name, "e4c_copy") == 0) t.bytes = (int64_t)b->h_cnt->cons_bytes[3];
    // a finalized call is one 240-byte record written; coverage = 8 B per read (start, end) + 20 B of queries per call
    if (strcmp(t.name, "e1w_finalize") == 0) t.bytes = (int64_t)sizeof(snf_call_t) * b->h_cnt->n_calls;
    if (strcmp(t.name, "d4_coverage") == 0) t.bytes = 8 * b->v.R + 20 * b->h_cnt->n_calls;
    if (strcmp(t.name, "f4_emit") == 0 && b->v.res_out) t.bytes = 2 * ((int64_t)sizeof(snf_call_t) * b->v.res_out->n_out + 4 * b->v.res_out->rn_out);
  }
  // running sums since snf_batch_timing_mean_reset: the mean launch duration of every kernel over the passes in between
  for (const auto& t : b->timings) {
    bool found = false;
    for (auto& a : b->timing_acc)
      if (strcmp(a.name, t.name) == 0) { a.ms += t.ms; a.bytes = t.bytes; a.launches++; found = true; break; }
    if (!found) b->timing_acc.push_back({t.name, t.ms, t.bytes, 1});
  }
}

// After everything of the pass has run: did the ALT stage leave work for the slow kernels (a call that fits none of the LDS
// classes, or a workgroup that handed its call to work list 7)?  Rare; they run now, and the ALT bytes are copied out again.
bool settle_alt_stage(snf_batch_impl* b) {      // true: ALT bytes were produced here, behind the pass
  View& v = b->v;
  if (!v.wave_path || !b->fused || v.NS <= 0) return false;   // (the plain path ran them inside finalize)
  const Counts& c = *b->h_cnt;
  if (c.n_cls[7] == 0 && c.n_cons_fallback == 0) return false;
  run_alt_fallback(b);     // (they store into the pass's ALT section like the fast kernels)
  LAUNCH_Q(z1_results, v, tail_threads(v), 0);
  dsync(b);
  return true;
}

void do_fetch(snf_batch_impl* b, int stage, snf_result_t* out) {
  SNF_TRACE(stage >= 1 ? "snf_batch_fetch(1): wait + result block" : "snf_batch_fetch(0): wait + candidates");
  View& v = b->v;
  int T = v.T;
  b->r_status.assign(T, 0); b->r_off.assign(T + 1, 0); b->r_cov.assign(T, NAN);
  full_sync(b);  // everything enqueued so far, incl. z1_results -> the pinned result block is current
  pass_waited(b);
  if (b->h_cnt->overflow) fail(b->h_cnt->overflow & 2 ? "internal: hand-over list of the call kernels overflowed" : "internal: fused-sequence pool overflow");
  if (v.prof) {
    const Counts& c = *b->h_cnt;
    fprintf(stderr, "[SNF_PROF] counts: valid %lld bins %lld seeds %lld clusters %lld refined %lld calls %lld | cons calls %lld reads %lld "
                    "fallback %lld alt bytes %lld fused bytes through the shared counter %llu (the rest in the waves' own slices; the split depends on scheduling, the total and the capacity do not) | ALT lists copy %llu small %llu large %llu/%llu/%llu/%llu thread %llu rows %llu\n", (long long)c.n_valid, (long long)c.n_bins, (long long)c.n_seeds, (long long)c.n_clusters,
            (long long)c.n_rc, (long long)c.n_calls, (long long)c.n_cons, (long long)c.n_cons_reads, (long long)c.n_cons_fallback,
            (long long)c.alt_total, c.pool_extra_used, c.n_cls[0], c.n_cls[1], c.n_cls[2], c.n_cls[3], c.n_cls[4], c.n_cls[5], c.n_cls[6], c.n_cls[7]);
  }
  memcpy(b->r_status.data(), v.res_status, (size_t)T * sizeof(int32_t));
  memcpy(b->r_cov.data(), v.res_cov, (size_t)T * sizeof(double));
  if (stage >= 1 && b->finalized) {
    // ---- the block of the output stage: already in pinned host memory, or one copy away
    const bool alt_late = settle_alt_stage(b);
    {
      const Counts& c = *b->h_cnt;
      bool skipped_work = false;
      for (int k = 0; k < 3; k++) {
        if (b->skipped_big[k] && c.n_big[k] > 0) skipped_work = true;
        b->hist_big[k] = c.n_big[k];        // (refreshed first: the next pass of this handle launches what this one skipped)
      }
      if (skipped_work) { b->pass_idle = true; fail("internal: a pass skipped x_big although items were handed to it (the handle's next pass launches it)"); }
      b->have_big_hist = true;
      b->hist_small = (int64_t)c.n_cls[1]; b->hist_large = (int64_t)(c.n_cls[2] + c.n_cls[3] + c.n_cls[4] + c.n_cls[5]); b->hist_copy = (int64_t)c.n_cls[0];
      b->hist_calls = c.n_calls; b->have_hist = true;
    }
    const OutHdr h = *v.res_out;
    const uint8_t* base = (const uint8_t*)b->hb_out.p;
    const int64_t alt_total_now = b->h_cnt->alt_total;
    // staged result, copied by the pass's own copy kernels (they read the sizes on the device and do nothing when a section does not fit)
    const bool block_there = !h.in_pinned && b->staged_kernel && b->stage_block_copied && v.stage_out_pin == (uint8_t*)b->hb_out.p && h.bytes <= v.stage_out_cap;
    const bool alt_there = !b->h_cnt->alt_in_pinned && b->staged_kernel && b->stage_alt_copied && !alt_late && v.stage_alt_pin == (uint8_t*)b->hb_alt.p && alt_total_now <= v.stage_alt_cap;
    bool need_sync = false;
    // (the copies of a staged result: one pass at a time - see pass_begins)
    std::unique_lock<std::mutex> copy_turn(pacing_of(b).copy_mu, std::defer_lock);
    if (b->staged && pace_copy_turn() && ((!h.in_pinned && !block_there) || (!b->h_cnt->alt_in_pinned && !alt_there))) copy_turn.lock();
    if (!h.in_pinned && !block_there) {
      v.out_pin = nullptr; v.out_pin_cap = 0;   // (a larger pinned block replaces the old one: the next finalize takes it)
      base = (const uint8_t*)b->hb_out.ensure((size_t)h.bytes + 256);
      d2h_timed(b, (void*)base, v.out_dev, (size_t)h.bytes, "d2h_block");
      need_sync = true;
    }
    const int64_t alt_total = b->h_cnt->alt_total;
    const uint8_t* alt = (const uint8_t*)b->hb_alt.p;
    if (!b->h_cnt->alt_in_pinned && !alt_there) {
      v.alt_pin = nullptr; v.alt_pin_cap = 0;
      alt = (const uint8_t*)b->hb_alt.ensure((size_t)alt_total + 256);
      if (alt_total) { d2h_timed(b, (void*)alt, v.alt_pool, (size_t)alt_total, "d2h_alt"); need_sync = true; }
    }
    if (need_sync) dsync(b);      // (both copies behind each other on the stream, one wait)
    if (copy_turn.owns_lock()) copy_turn.unlock();
    memcpy(b->r_off.data(), v.res_off, ((size_t)T + 1) * sizeof(int64_t));
    collect_timings(b);
    out->n_calls = h.n_out; out->calls = (const snf_call_t*)base;
    out->alt_pool_len = alt_total; out->alt_pool = alt;
    out->rnames_len = h.rn_out; out->rnames = (const uint32_t*)(base + h.off_rn);
    out->n_tasks = T; out->task_status = b->r_status.data(); out->task_call_off = b->r_off.data();
    out->coverage_average_total = b->r_cov.data();
    b->pass_idle = true;
    return;
  }
  // ---- candidates (stage 0, or no finalize yet): records and read names as the candidate stage left them in HBM
  if (b->rn_state != 0) { enqueue_rnames_late(b, true); dsync(b); }   // (names were deferred for an execute-mode finalize: all of them now)
  int64_t nc = v.NS > 0 ? b->h_cnt->n_calls : 0;
  int64_t rn_total = v.NS > 0 ? b->h_cnt->rn_total : 0;
  snf_call_t* calls = (snf_call_t*)b->hb_calls.ensure((size_t)(nc + 1) * sizeof(snf_call_t));
  uint32_t* rn = (uint32_t*)b->hb_rn.ensure((size_t)(rn_total + 1) * sizeof(uint32_t));
  d2h_timed(b, calls, v.calls, (size_t)nc * sizeof(snf_call_t), "d2h_calls");
  if (rn_total) d2h_timed(b, rn, v.rnames, (size_t)rn_total * sizeof(uint32_t), "d2h_rnames");
  dsync(b);
  if (v.out_valid) {   // (a stage-0 fetch after finalize: the result block holds the offsets of the output block)
    std::vector<int64_t> off((size_t)T + 1);
    d2h(b, off.data(), v.t_call_off, ((size_t)T + 1) * sizeof(int64_t)); dsync(b);
    b->r_off = off;
  } else memcpy(b->r_off.data(), v.res_off, ((size_t)T + 1) * sizeof(int64_t));
  if (v.NS <= 0) std::fill(b->r_off.begin(), b->r_off.end(), 0);
  // tasks whose reference run raises (SNF_TASK_ERR_*) yield no calls: squeeze them out (rare; in place)
  bool any_err = false;
  for (int t = 0; t < T; t++) any_err |= b->r_status[t] != SNF_TASK_OK;
  if (any_err) {
    int64_t w = 0;
    std::vector<int64_t> noff((size_t)T + 1, 0);
    for (int t = 0; t < T; t++) {
      noff[t] = w;
      if (b->r_status[t] != SNF_TASK_OK) continue;
      for (int64_t i = b->r_off[t]; i < b->r_off[t + 1]; i++) calls[w++] = calls[i];
    }
    noff[T] = w; b->r_off = noff; nc = w;
  }
  for (int64_t i = 0; i < nc; i++) { calls[i].alt_len = -1; calls[i].alt_off = 0; }
  collect_timings(b);
  out->n_calls = nc; out->calls = calls;
  out->alt_pool_len = 0; out->alt_pool = (const uint8_t*)calls;
  out->rnames_len = rn_total; out->rnames = rn;
  out->n_tasks = T; out->task_status = b->r_status.data(); out->task_call_off = b->r_off.data();
  out->coverage_average_total = b->r_cov.data();
  b->pass_idle = true;
}

void do_add_task(snf_batch_impl* b, const snf_task_input_t* t) {
  if (b->uploaded) fail("snf_batch_add_task after snf_batch_upload");
  int64_t n = t->n_leads, r = t->n_reads;
  if (n < 0 || r < 0 || t->contig_len < 0 || t->seq_pool_len < 0) fail("negative sizes in task input");
  if (n > 0 && (!t->ref_start || !t->ref_end || !t->qry_start || !t->qry_end || !t->svlen || !t->read_len || !t->qname_id ||
                !t->read_id || !t->ps_rank || !t->mate_contig || !t->mate_ref_start || !t->seq_len || !t->seq_off || !t->nm ||
                !t->svtype || !t->strand || !t->mapq || !t->source || !t->hap || !t->is_sa || !t->bnd_is_first || !t->bnd_is_reverse))
    fail("null lead column");
  if (r > 0 && (!t->read_start || !t->read_end || !t->read_hp)) fail("null read column");
  if (t->seq_pool_len > 0 && !t->seq_pool) fail("null sequence pool");
  int64_t ntr = t->n_tr > 0 ? t->n_tr : 0;
  int32_t pm = INT32_MIN;
  for (int64_t i = 0; i < ntr; i++) {
    b->h_trs.push_back(t->tr_start[i]); b->h_tre.push_back(t->tr_end[i]);
    if (t->tr_end[i] > pm) pm = t->tr_end[i];
    b->h_trp.push_back(pm);
  }
  b->h_has_tr.push_back(ntr > 0 ? 1 : 0);  // tr None or [] -> no TR handling (cluster.py:229-235)
  {
    const int64_t nn = t->n_nmask > 0 ? t->n_nmask : 0;
    if (nn > 0 && (!t->nmask_start || !t->nmask_end)) fail("null mask arrays");
    int32_t prev = 0;
    for (int64_t i = 0; i < nn; i++) {
      const int32_t a = t->nmask_start[i], e = t->nmask_end[i];
      if (a < prev || e <= a || e > t->contig_len) fail("mask intervals must be sorted, disjoint, non-empty and inside the contig (leadprov.py:434-441)");
      b->h_nms.push_back(a); b->h_nme.push_back(e); prev = e;
    }
    b->h_nm_off.push_back((int64_t)b->h_nms.size());
  }
  b->h_lead_off.push_back(b->h_lead_off.back() + n);
  b->h_read_off.push_back(b->h_read_off.back() + r);
  b->h_pool_off.push_back(b->h_pool_off.back() + t->seq_pool_len);
  b->h_tr_off.push_back((int64_t)b->h_trs.size());
  b->tasks.push_back(*t);   // the content (values, order of the reads) is checked when the upload stages it
  b->task_on_device.push_back(0);
}

}  // namespace

namespace {
void do_consensus_batch(int device, int klen, const uint8_t* seq_pool, int64_t seq_pool_len, int64_t n_problems,
                        const int64_t* best_off, const int32_t* best_len, const int32_t* skip, const int32_t* skip_rep, const int64_t* others_index,
                        const int64_t* others_off, const int32_t* others_len, uint8_t* out_pool, const int64_t* out_off) {
    if (n_problems <= 0) return;
    if (!seq_pool || !best_off || !best_len || !skip || !others_index || !others_off || !others_len || !out_pool || !out_off)
      fail("snf_consensus_batch: null argument");
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0 || device < 0 || device >= nd) fail("no such HIP device");
    SNF_HIP(hipSetDevice(device));
    if (klen < 1 || klen > 8) fail("snf_consensus_batch: klen must be 1..8 (a k-mer is compared as one 64-bit word)");
    // '-' is the reference's gap symbol (consensus.py:317-380): a pool that holds one goes to the literal thread kernels as a whole
    const bool has_dash = seq_pool_len > 0 && memchr(seq_pool, '-', (size_t)seq_pool_len) != nullptr;
    const int64_t np = n_problems, n_reads = others_index[np];
    std::vector<ConsDesc> descs((size_t)np);
    std::vector<int32_t> lists[8];
    std::vector<int64_t> generic;      // problems for the thread kernels e4 / e5 / e6: '-' bytes, skip_repetitive != skip, beyond the workgroup kernels' limits
    Counts hc{};
    int64_t aln_total = 0, alt_total = 0;
    for (int64_t p = 0; p < np; p++) {
      const int64_t L = best_len[p]; const int64_t no = others_index[p + 1] - others_index[p];
      if (L < 0 || no < 0 || best_off[p] < 0 || best_off[p] + L > seq_pool_len) fail("best sequence outside the pool");
      if (out_off[p + 1] - out_off[p] != L) fail("out_off must be the prefix sums of best_len");
      if (skip[p] < 1 || (skip_rep && skip_rep[p] < 1)) fail("sampling steps must be >= 1 (range() of the reference raises on 0)");
      for (int64_t k = others_index[p]; k < others_index[p + 1]; k++)
        if (others_len[k] < 0 || others_off[k] < 0 || others_off[k] + others_len[k] > seq_pool_len) fail("other sequence outside the pool");
      const int cls = (has_dash || (skip_rep && skip_rep[p] != skip[p])) ? 0 : cons_class_of(1, klen, skip[p], L, (int32_t)no);
      if (cls == 0) { generic.push_back(p); continue; }
      ConsDesc d{};
      d.best_off = best_off[p]; d.alt_off = out_off[p]; d.aln_off = aln_total; d.read_off = others_index[p];
      d.L = (int32_t)L; d.n_others = (int32_t)no; d.skip = skip[p]; d.cls = cls;
      descs[(size_t)p] = d;
      int lid = cls;
      if (cls == 2) { const int64_t work = no * L; lid = work >= 32768 ? 2 : work >= 16384 ? 3 : work >= 8192 ? 4 : 5; }
      else if (cls == 4) lid = 7;
      lists[lid].push_back((int32_t)p); hc.n_cls[lid]++;
      aln_total += no * L;
    }
    alt_total = out_off[np];
    std::vector<void*> frees;
    auto dev = [&](size_t bytes) { void* q = nullptr; SNF_HIP(hipMalloc(&q, bytes ? bytes : 1)); frees.push_back(q); return q; };
    auto up = [&](const void* h, size_t bytes) { void* q = dev(bytes); if (bytes) SNF_HIP(hipMemcpy(q, h, bytes, hipMemcpyHostToDevice)); return q; };
    struct Guard { std::vector<void*>& f; ~Guard() { for (void* q : f) (void)hipFree(q); } } guard{frees};
    View v{};
    v.cfg.consensus_kmer_len = klen;
    v.wave_path = 1;
    uint8_t* dpool = (uint8_t*)dev((size_t)seq_pool_len + 32);
    SNF_HIP(hipMemset(dpool, 0, (size_t)seq_pool_len + 32));
    if (seq_pool_len) SNF_HIP(hipMemcpy(dpool, seq_pool, (size_t)seq_pool_len, hipMemcpyHostToDevice));
    v.pool = dpool; v.pool_len = seq_pool_len; v.pool_cap = seq_pool_len + 32;
    v.alt_pool = (uint8_t*)dev((size_t)alt_total + 16);
    if ((int64_t)generic.size() < np) {
    v.cdesc = (ConsDesc*)up(descs.data(), descs.size() * sizeof(ConsDesc));
    for (int k = 1; k < 6; k++) v.cls_list[k] = (int32_t*)up(lists[k].data(), lists[k].size() * sizeof(int32_t));
    lists[7].resize((size_t)np, 0);   // room for every problem: SMALL / LARGE may hand calls over at run time
    v.cls_list[7] = (int32_t*)up(lists[7].data(), lists[7].size() * sizeof(int32_t));
    v.cnt = (Counts*)up(&hc, sizeof(Counts));
    v.crl_off = (int64_t*)up(others_off, (size_t)n_reads * sizeof(int64_t));
    v.crl_len = (int32_t*)up(others_len, (size_t)n_reads * sizeof(int32_t));
    v.aln = (uint8_t*)dev((size_t)aln_total + 16);
    v.aln_kept_w = (uint8_t*)dev((size_t)n_reads + 16);
    v.stripes = (unsigned long long*)dev(4 * 64 * 16 * sizeof(unsigned long long));
    SNF_HIP(hipMemset(v.stripes, 0, 4 * 64 * 16 * sizeof(unsigned long long)));
    const int64_t n_small = (int64_t)hc.n_cls[1], n_large = (int64_t)(hc.n_cls[2] + hc.n_cls[3] + hc.n_cls[4] + hc.n_cls[5]);
    if (n_large > 0) hipLaunchKernelGGL((K_CONS_LARGE), dim3((unsigned)(n_large < 16384 ? n_large : 16384)), dim3(256), 0, 0, v, (int64_t)0);
    if (n_small > 0) hipLaunchKernelGGL((K_CONS_SMALL(5)), dim3((unsigned)(n_small < 16384 ? n_small : 16384)), dim3(256), 0, 0, v, (int64_t)0);
    // list 7: calls beyond the LDS vote counters, plus whatever the two kernels above handed over (null stream: ordered)
    hipLaunchKernelGGL((K_CONS_ROWS), dim3((unsigned)(np < 4096 ? np : 4096)), dim3(256), 0, 0, v, (int64_t)0);
    SNF_HIP(hipGetLastError());
    }
    if (!generic.empty()) {
      // The literal thread kernels of the batch pipeline (e4_anchor: anchor table with `skip_repetitive`; e5_align: one thread per other
      // read, rows with '-' as the gap byte; e6_vote: one thread per column) over a synthetic call table: call k = generic problem k,
      // F slots = its sequences (best first), alt offsets = the caller's out_off.
      const int64_t ng = (int64_t)generic.size();
      std::vector<CallX> cx((size_t)ng);
      std::vector<int32_t> ccall((size_t)ng), fi, flen, sk((size_t)ng), skr((size_t)ng);
      std::vector<int64_t> foff, roff((size_t)ng + 1), toff((size_t)ng + 1), tsz((size_t)ng), aoff((size_t)ng + 1);
      std::vector<uint32_t> pn((size_t)np + 1);
      // e6_vote finds the call of a column by its alt offset: every problem is a call there, the generic ones flagged for consensus
      std::vector<CallX> call_all((size_t)np);
      for (int64_t p = 0; p < np; p++) { pn[(size_t)p] = (uint32_t)out_off[p]; CallX x{}; x.do_cons = 0; x.best = -1; x.cons_id = -1; call_all[(size_t)p] = x; }
      pn[(size_t)np] = (uint32_t)out_off[np];
      if (out_off[np] >= ((int64_t)1 << 32)) fail("snf_consensus_batch: more than 4 GB of output");
      int64_t r_tot = 0, t_tot = 0, a_tot = 0;
      for (int64_t k = 0; k < ng; k++) {
        const int64_t p = generic[(size_t)k], L = best_len[p], no = others_index[p + 1] - others_index[p];
        CallX x{};
        x.flo = (int32_t)fi.size(); x.fn = (int32_t)(no + 1); x.best = x.flo; x.n_others = (int32_t)no; x.do_cons = 1; x.cons_id = (int32_t)k;
        fi.push_back((int32_t)fi.size()); flen.push_back((int32_t)L); foff.push_back(best_off[p]);
        for (int64_t q = others_index[p]; q < others_index[p + 1]; q++) { fi.push_back((int32_t)fi.size()); flen.push_back(others_len[q]); foff.push_back(others_off[q]); }
        if (fi.size() >= ((size_t)1 << 31)) fail("snf_consensus_batch: too many sequences");
        call_all[(size_t)p] = x; ccall[(size_t)k] = (int32_t)p;
        sk[(size_t)k] = skip[p]; skr[(size_t)k] = skip_rep ? skip_rep[p] : skip[p];
        const int64_t npos = cons_npos(L, klen, skr[(size_t)k]);
        int64_t hs = 16; while (hs < 2 * npos + 2) hs <<= 1;
        roff[(size_t)k] = r_tot; toff[(size_t)k] = t_tot; tsz[(size_t)k] = hs; aoff[(size_t)k] = a_tot;
        r_tot += no; t_tot += hs; a_tot += no * L;
      }
      roff[(size_t)ng] = r_tot; toff[(size_t)ng] = t_tot; aoff[(size_t)ng] = a_tot;
      View g{};
      g.cfg.consensus_kmer_len = klen; g.wave_path = 1; g.cons_thread_only = 1;   // (e6_vote leaves the columns of the other problems alone)
      g.pool = dpool; g.pool_len = seq_pool_len; g.pool_cap = seq_pool_len + 32;
      Counts gc{}; gc.n_cons = ng; gc.n_cons_reads = r_tot; gc.alt_total = out_off[np]; gc.n_calls = np; gc.alt_in_pinned = 0;
      g.cnt = (Counts*)up(&gc, sizeof(Counts));
      g.cons_call = (int32_t*)up(ccall.data(), ccall.size() * 4);
      g.callx = (CallX*)up(call_all.data(), call_all.size() * sizeof(CallX));
      g.F_seq_len = (int32_t*)up(flen.data(), flen.size() * 4); g.F_seq_off = (int64_t*)up(foff.data(), foff.size() * 8);
      g.FI = (int32_t*)up(fi.data(), fi.size() * 4);
      g.cons_read_off = (int64_t*)up(roff.data(), roff.size() * 8); g.cons_tab_off = (int64_t*)up(toff.data(), toff.size() * 8);
      g.cons_tab_sz = (int64_t*)up(tsz.data(), tsz.size() * 8); g.cons_aln_off = (int64_t*)up(aoff.data(), aoff.size() * 8);
      g.cons_skip_arr = (int32_t*)up(sk.data(), sk.size() * 4); g.cons_skiprep_arr = (int32_t*)up(skr.data(), skr.size() * 4);
      g.pN = (uint32_t*)up(pn.data(), pn.size() * 4);
      g.tab_key = (uint64_t*)dev((size_t)t_tot * 8); g.tab_pos = (int32_t*)dev((size_t)t_tot * 4); g.tab_state = (uint8_t*)dev((size_t)t_tot);
      g.aln = (uint8_t*)dev((size_t)a_tot + 16); g.aln_kept = (uint8_t*)dev((size_t)r_tot + 16);
      g.cr_call = (int32_t*)dev((size_t)r_tot * 4 + 4); g.cr_read = (int32_t*)dev((size_t)r_tot * 4 + 4);
      g.alt_pool = v.alt_pool;
      auto grid = [](int64_t n) { return dim3((unsigned)((n + 255) / 256)); };
      hipLaunchKernelGGL(e4_anchor, grid(ng), dim3(256), 0, 0, g, ng);
      if (r_tot > 0) hipLaunchKernelGGL(e5_align, grid(r_tot), dim3(256), 0, 0, g, r_tot);
      if (out_off[np] > 0) hipLaunchKernelGGL(e6_vote, grid(out_off[np]), dim3(256), 0, 0, g, out_off[np]);
      SNF_HIP(hipGetLastError());
    }
    SNF_HIP(hipDeviceSynchronize());
    if (alt_total) SNF_HIP(hipMemcpy(out_pool, v.alt_pool, (size_t)alt_total, hipMemcpyDeviceToHost));
}
}  // namespace

// ---------------------------------------------------------------------------------------------- C ABI
void do_coverage_calls(snf_batch_t* bb, int32_t task_index, int64_t n, const int32_t* svtype, const int32_t* pos,
                       const int32_t* svlen, const uint8_t* bnd_is_first, int32_t* cov, int32_t* status, double* coverage_mean) {
    auto b = reinterpret_cast<snf_batch_impl*>(bb);
    if (!b || !b->uploaded || !status || !coverage_mean) fail("batch not uploaded / null argument");
    if (!b->cov_avg_ready) fail("snf_batch_coverage_calls needs snf_batch_call_candidates first (coverage.mean() is formed there)");
    if (task_index < 0 || task_index >= b->v.T) fail("task index out of range");
    if (n < 0 || (n > 0 && (!svtype || !pos || !svlen || !bnd_is_first || !cov))) fail("invalid call arrays");
    SNF_HIP(hipSetDevice(b->device));
    full_sync(b);
    *status = 0;
    d2h(b, coverage_mean, b->v.t_cov_avg + task_index, sizeof(double));
    dsync(b);
    if (n == 0) return;
    CovCalls q{};
    q.r_start = b->v.r_start; q.re_sorted = b->v.re_sorted; q.rs_top = b->v.rs_top; q.re_top = b->v.re_top;
    q.lo = b->h_read_off[(size_t)task_index]; q.hi = b->h_read_off[(size_t)task_index + 1];
    q.L = b->tasks[(size_t)task_index].contig_len; q.n = n;
    q.nm_start = b->v.nm_start; q.nm_end = b->v.nm_end; q.nm_lo = b->h_nm_off[(size_t)task_index]; q.nm_hi = b->h_nm_off[(size_t)task_index + 1];
    q.binsize = b->cfg.coverage_binsize; q.updown = b->cfg.coverage_updown_bins;
    int32_t* d_t = dalloc_own<int32_t>(b, (size_t)n); int32_t* d_p = dalloc_own<int32_t>(b, (size_t)n); int32_t* d_l = dalloc_own<int32_t>(b, (size_t)n);
    uint8_t* d_f = dalloc_own<uint8_t>(b, (size_t)n);
    q.end = dalloc_own<int64_t>(b, (size_t)n); q.cov = dalloc_own<int32_t>(b, (size_t)n * 5); q.n_valid = dalloc_own<int32_t>(b, 2);
    h2d(b, d_t, svtype, (size_t)n * 4); h2d(b, d_p, pos, (size_t)n * 4); h2d(b, d_l, svlen, (size_t)n * 4); h2d(b, d_f, bnd_is_first, (size_t)n);
    h2d(b, q.cov, cov, (size_t)n * 20);
    q.svtype = d_t; q.pos = d_p; q.svlen = d_l; q.bnd_is_first = d_f;
    LAUNCH(s2_covends, q, 1, n * 13);
    LAUNCH(s2_covcalls, q, n, (q.hi - q.lo) * 8 + n * 40);
    int32_t nv[2] = {0, 0};
    d2h(b, cov, q.cov, (size_t)n * 20);
    d2h(b, nv, q.n_valid, 8);
    dsync(b);
    *status = nv[1];
    void* tmp[] = {d_t, d_p, d_l, d_f, q.end, q.cov, q.n_valid};
    for (void* p : tmp) dfree_one(b, p);
}

// ---- stand-alone genotyping (--reqc): genotype_sv of snf_stage_final.h over caller-supplied records
struct GenoView { snf_config_t cfg; const GtEntry* gt_lut; snf_call_t* calls; int64_t n; };
namespace snf {
SNF_HD void g1_genotype_body(int64_t i, const GenoView& q) {
  // genotype_sv only reads the configuration and the lookup table through the view
  View v{};
  v.cfg = q.cfg; v.gt_lut = q.gt_lut;
  snf_call_t c = q.calls[i];
  genotype_sv(v, c, c.gt_hp, c.gt_ps);
  q.calls[i] = c;
}
}  // namespace snf
SNF_KERNEL(g1_genotype, GenoView)
namespace {
DevArena g_geno_arenas[SNF_MAX_DEVICES];
void do_genotype_batch(const snf_config_t* cfg, int device, snf_call_t* calls, int64_t n) {
  if (!cfg || (n > 0 && !calls)) fail("null argument");
  if (n <= 0) return;
  if (cfg->genotype_ploidy != 2) fail("only genotype_ploidy 2 is supported");
  if (device < 0 || device >= SNF_MAX_DEVICES) fail("device index out of range");
  const auto& lut = build_gt_lut(*cfg);
  DevArena& A = g_geno_arenas[device];
  std::lock_guard<std::mutex> hold(A.mu);
  ArenaLayout L;
  const size_t o_lut = L.add<GtEntry>(lut.size()), o_calls = L.add<snf_call_t>((size_t)n);
  if (!A.ensure(device, L.at)) fail("no HIP device available: the sniffles_amd hot path requires an AMD GPU (there is no CPU fallback)");
  memcpy(A.h + o_lut, lut.data(), lut.size() * sizeof(GtEntry));
  memcpy(A.h + o_calls, calls, (size_t)n * sizeof(snf_call_t));
  GenoView q{};
  q.cfg = *cfg; q.gt_lut = (const GtEntry*)(A.d + o_lut); q.calls = (snf_call_t*)(A.d + o_calls); q.n = n;
  SNF_HIP(hipMemcpyAsync(A.d, A.h, L.at, hipMemcpyHostToDevice, A.stream));
  hipLaunchKernelGGL(g1_genotype, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, A.stream, q, n);
  SNF_HIP(hipGetLastError());
  SNF_HIP(hipMemcpyAsync(A.h + o_calls, A.d + o_calls, (size_t)n * sizeof(snf_call_t), hipMemcpyDeviceToHost, A.stream));
  SNF_HIP(hipStreamSynchronize(A.stream));
  memcpy(calls, A.h + o_calls, (size_t)n * sizeof(snf_call_t));
}
}  // namespace

// ---- cluster export (seam B3): everything is copied to the host and assembled there - a debugging / parity aid
template <class T>
std::vector<T> pull(snf_batch_impl* b, const T* dptr, size_t n) {
  std::vector<T> h(n ? n : 1);
  d2h(b, h.data(), dptr, n * sizeof(T));
  return h;
}
void do_fetch_clusters(snf_batch_impl* b, int stage, snf_clusters_t* out) {
  View& v = b->v;
  if (stage < 0 || stage > 2) fail("stage must be 0 (seeds), 1 (merged) or 2 (refined)");
  if (!b->cov_avg_ready) fail("snf_batch_fetch_clusters needs snf_batch_call_candidates first");
  if (v.prefilter) {
    // Cluster ids carry the seed's index among ALL occupied bins of its (task, svtype) sequence (cluster.py:238-246), which
    // the occupancy prefilter does not keep: the candidate stage is redone over every lead (this seam is a debugging /
    // parity aid; the calls are the same either way) and the batch stays unfiltered from here on
    const bool fin = b->finalized;
    full_sync(b);
    v.prefilter = 0; v.front = 0; v.NS = v.N;
    run_call_candidates(b);
    if (fin) run_finalize(b);
  }
  full_sync(b);
  const Counts c = *b->h_cnt;
  b->cl_task.clear(); b->cl_svtype.clear(); b->cl_start.clear(); b->cl_end.clear(); b->cl_seed.clear(); b->cl_seed_index.clear();
  b->cl_nlong.clear(); b->cl_lead.clear(); b->cl_lead_svlen.clear(); b->cl_repeat.clear(); b->cl_lead_off.assign(1, 0);
  const int64_t N = v.N;
  if (N > 0 && c.n_seeds > 0) {
    const size_t ns = (size_t)c.n_seeds, ncl = (size_t)c.n_clusters, nrc = (size_t)c.n_rc, nf = (size_t)c.NF;
    auto seed_grp = pull(b, v.seed_grp, ns); auto seed_start = pull(b, v.seed_start, ns); auto seed_bin = pull(b, v.seed_bin, ns);
    auto seed_lo = pull(b, v.seed_lo, ns); auto seed_hi = pull(b, v.seed_hi, ns);
    auto seedL_lo = pull(b, v.seedL_lo, ns); auto seedL_hi = pull(b, v.seedL_hi, ns);
    auto s_repeat0 = pull(b, v.s_repeat0, ns);
    auto grp_first_bin = pull(b, v.grp_first_bin, (size_t)(8 * v.T + 8));
    auto L = pull(b, v.L, nf);
    auto lead_off = pull(b, v.t_lead_off, (size_t)v.T + 1);
    auto in_svlen = pull(b, v.in_svlen, (size_t)N);
    std::vector<int32_t> cl_head, c_last, c_end, rc_lo, rc_n, rc_cluster, FI, F_orig, F_svlen; std::vector<uint8_t> c_repeat, rc_keeplong;
    if (stage >= 1) { cl_head = pull(b, v.cl_head, ncl); c_last = pull(b, v.c_last, ns); c_end = pull(b, v.c_end, ns); c_repeat = pull(b, v.c_repeat, ns); }
    if (stage == 2) {
      rc_lo = pull(b, v.rc_lo, nrc); rc_n = pull(b, v.rc_n, nrc); rc_cluster = pull(b, v.rc_cluster, nrc); rc_keeplong = pull(b, v.rc_keeplong, nrc);
      FI = pull(b, v.FI, nf); F_orig = pull(b, v.F_orig, nf); F_svlen = pull(b, v.F_svlen, nf);
    }
    dsync(b);
    auto head = [&](int32_t h, int32_t last, int32_t end, uint8_t rep, int64_t nlong) {
      const int g = seed_grp[(size_t)h];
      b->cl_task.push_back(g >> 3); b->cl_svtype.push_back(g & 7); b->cl_start.push_back(seed_start[(size_t)h]); b->cl_end.push_back(end);
      b->cl_seed.push_back(seed_start[(size_t)h]); b->cl_seed_index.push_back(seed_bin[(size_t)h] - grp_first_bin[(size_t)g]);
      b->cl_nlong.push_back((int32_t)nlong); b->cl_repeat.push_back(rep);
      (void)last;
    };
    auto add_row = [&](uint32_t orig, int32_t svlen, int task) {
      b->cl_lead.push_back((int32_t)((int64_t)orig - lead_off[(size_t)task])); b->cl_lead_svlen.push_back(svlen);
    };
    const int binsize = b->cfg.cluster_binsize;
    if (stage == 0) {
      for (size_t sidx = 0; sidx < ns; sidx++) {
        head((int32_t)sidx, (int32_t)sidx, seed_start[sidx] + binsize, s_repeat0[sidx], seedL_hi[sidx] - seedL_lo[sidx]);
        for (int32_t k = seed_lo[sidx]; k < seed_hi[sidx]; k++) add_row(L[(size_t)k], in_svlen[L[(size_t)k]], seed_grp[sidx] >> 3);
        b->cl_lead_off.push_back((int64_t)b->cl_lead.size());
      }
    } else if (stage == 1) {
      for (size_t ci = 0; ci < ncl; ci++) {
        const int32_t h = cl_head[ci], last = c_last[(size_t)h];
        int64_t nlong = 0;
        for (int32_t sidx = h; sidx <= last; sidx++) nlong += seedL_hi[(size_t)sidx] - seedL_lo[(size_t)sidx];   // merged seeds are consecutive
        head(h, last, c_end[(size_t)h], c_repeat[(size_t)h], nlong);
        for (int32_t k = seed_lo[(size_t)h]; k < seed_hi[(size_t)last]; k++) add_row(L[(size_t)k], in_svlen[L[(size_t)k]], seed_grp[(size_t)h] >> 3);
        b->cl_lead_off.push_back((int64_t)b->cl_lead.size());
      }
    } else {
      for (size_t r = 0; r < nrc; r++) {
        const int32_t ci = rc_cluster[r], h = cl_head[(size_t)ci], last = c_last[(size_t)h];
        int64_t nlong = 0;
        if (rc_keeplong[r]) for (int32_t sidx = h; sidx <= last; sidx++) nlong += seedL_hi[(size_t)sidx] - seedL_lo[(size_t)sidx];
        head(h, last, c_end[(size_t)h], c_repeat[(size_t)h], nlong);
        for (int32_t k = 0; k < rc_n[r]; k++) {
          const int32_t slot = FI[(size_t)(rc_lo[r] + k)];
          add_row((uint32_t)F_orig[(size_t)slot], F_svlen[(size_t)slot], seed_grp[(size_t)h] >> 3);
        }
        b->cl_lead_off.push_back((int64_t)b->cl_lead.size());
      }
    }
  }
  out->n_clusters = (int64_t)b->cl_task.size();
  out->task_index = b->cl_task.data(); out->svtype = b->cl_svtype.data(); out->start = b->cl_start.data(); out->end = b->cl_end.data();
  out->seed = b->cl_seed.data(); out->seed_index = b->cl_seed_index.data(); out->n_leads_long = b->cl_nlong.data(); out->repeat = b->cl_repeat.data();
  out->lead_off = b->cl_lead_off.data(); out->n_leads = (int64_t)b->cl_lead.size(); out->lead = b->cl_lead.data(); out->lead_svlen = b->cl_lead_svlen.data();
}

extern "C" int snf_extract_device_view(snf_extract_t* x, snf_task_input_t* out, int* device);
void do_add_task_device(snf_batch_impl* b, snf_extract_t* x, const snf_task_input_t* meta) {
  if (!b || !x || !meta) fail("null argument");
  snf_task_input_t t{};
  int dev = -1;
  if (snf_extract_device_view(x, &t, &dev) != 0) fail(std::string("snf_batch_add_task_device: ") + snf_extract_last_error());
  if (dev != b->device) fail("snf_batch_add_task_device: the extraction ran on another device");
  t.task_id = meta->task_id; t.sv_id_start = meta->sv_id_start; t.contig_len = meta->contig_len;
  t.n_tr = meta->n_tr; t.tr_start = meta->tr_start; t.tr_end = meta->tr_end;
  t.n_nmask = meta->n_nmask; t.nmask_start = meta->nmask_start; t.nmask_end = meta->nmask_end;
  // (ps_null_rank and qc_nm_threshold are the extraction's own)
  const uint8_t* pool = t.seq_pool;
  const int64_t pool_len = t.seq_pool_len, n = t.n_leads, r = t.n_reads;
  t.seq_pool = nullptr; t.seq_pool_len = 0; t.n_leads = 0; t.n_reads = 0;   // do_add_task checks HOST pointers: the device ones go back in below
  do_add_task(b, &t);
  snf_task_input_t& q = b->tasks.back();
  q.n_leads = n; q.n_reads = r; q.seq_pool = pool; q.seq_pool_len = pool_len;
  b->h_lead_off.back() += n; b->h_read_off.back() += r; b->h_pool_off.back() += pool_len;
  b->task_on_device.back() = 1;
}

#define SNF_TRY(body)                                   \
  try { body; return 0; }                               \
  catch (const snf::Error& e) { g_err = e.msg; return 1; } \
  catch (const std::exception& e) { g_err = e.what(); return 1; }

extern "C" {

int snf_abi_version(void) { return SNF_ABI_VERSION; }
const char* snf_last_error(void) { return g_err.c_str(); }

int snf_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// Creating a HIP stream sets up a hardware queue: about 2 ms each, 4 per batch, and as much again to destroy them - more than the
// kernels of a contig-sized batch take.  Streams of destroyed batches (synchronised, nothing pending) are kept per device and
// handed to the next batch; the pool is bounded, the rest is destroyed as before.
struct StreamPool {
  std::mutex mu;
  std::vector<hipStream_t> idle[64], idle_hi[64];   // (idle_hi: streams of the highest priority - the LARGE consensus kernel's)
  hipStream_t take(int device, bool high = false) {
    {
      std::lock_guard<std::mutex> g(mu);
      auto& v = (high ? idle_hi : idle)[device & 63];
      if (!v.empty()) { hipStream_t s = v.back(); v.pop_back(); return s; }
    }
    hipStream_t s = nullptr;
    if (high) {
      int least = 0, greatest = 0;
      SNF_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
      SNF_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, greatest));
    } else SNF_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    return s;
  }
  int trim(int device) {      // idle streams of `device` (< 0: all) are destroyed
    std::vector<std::pair<int, hipStream_t>> gone;
    {
      std::lock_guard<std::mutex> g(mu);
      for (int d = 0; d < 64; d++) if (device < 0 || (device & 63) == d) {
        for (auto s : idle[d]) gone.push_back({d, s});
        for (auto s : idle_hi[d]) gone.push_back({d, s});
        idle[d].clear(); idle_hi[d].clear();
      }
    }
    int cur = 0; (void)hipGetDevice(&cur);
    for (auto& e : gone) { (void)hipSetDevice(e.first); (void)hipStreamDestroy(e.second); }
    (void)hipSetDevice(cur);
    return (int)gone.size();
  }
  void give(int device, hipStream_t s, bool high = false) {
    {
      std::lock_guard<std::mutex> g(mu);
      auto& v = (high ? idle_hi : idle)[device & 63];
      if (v.size() < 32 && !getenv("SNF_NO_STREAM_POOL")) { v.push_back(s); return; }
    }
    (void)hipStreamDestroy(s);
  }
};
static StreamPool g_streams;

// The device's CU count and the occupancy of the three resident kernels do not change while the process lives: asked once per
// (device, instance choice) - hipGetDeviceProperties and the occupancy queries are milliseconds, a task-sized batch is not.
struct DevInfo { int cus, nb_d1w, nb_d2w, nb_e1w; };
typedef void (*WaveKernel)(const View, int64_t);
static WaveKernel pick_d2w(int o2, bool phase) {   // waves per SIMD the instance is compiled for x config.phase
  if (phase) return o2 >= 8 ? d2w_call<8, true> : o2 == 6 ? d2w_call<6, true> : o2 == 5 ? d2w_call<5, true> : d2w_call<4, true>;
  return o2 >= 8 ? d2w_call<8, false> : o2 == 6 ? d2w_call<6, false> : o2 == 5 ? d2w_call<5, false> : d2w_call<4, false>;
}
static DevInfo device_info(int device, int o2, int o1, WaveKernel k_d2w, WaveKernel k_e1w) {
  typedef std::tuple<int, int, int> Key;
  static std::mutex mu;
  static std::map<Key, DevInfo> known;
  std::lock_guard<std::mutex> g(mu);
  auto it = known.find(Key(device, o2, o1));
  if (it == known.end()) {
    hipDeviceProp_t prop;
    SNF_HIP(hipGetDeviceProperties(&prop, device));
    DevInfo d{prop.multiProcessorCount, 0, 0, 0};
    SNF_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&d.nb_d1w, d1w_refine, 64, 0));
    SNF_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&d.nb_d2w, k_d2w, 64, 0));
    SNF_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&d.nb_e1w, k_e1w, 64, 0));
    it = known.emplace(Key(device, o2, o1), d).first;
  }
  return it->second;
}

int snf_batch_create(const snf_config_t* cfg, int device, snf_batch_t** out) {
  SNF_TRY({
    if (!cfg || !out) fail("null argument");
    if (cfg->cluster_binsize <= 0 || cfg->cluster_resplit_binsize <= 0) fail("bin sizes must be positive");
    if (cfg->consensus_kmer_len < 1 || cfg->consensus_kmer_len > 8) fail("consensus_kmer_len must be in 1..8");
    if (cfg->genotype_ploidy != 2) fail("only genotype_ploidy 2 is supported");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
      fail("no HIP device available: the sniffles_amd hot path requires an AMD GPU (there is no CPU fallback)");
    if (device < 0 || device >= n) fail("device index out of range");
    SNF_HIP(hipSetDevice(device));
    auto b = std::make_unique<snf_batch_impl>();
    b->cfg = *cfg; b->device = device;
    const char* g = getenv("SNF_RUN_GAP");
    int base = cfg->cluster_merge_bnd > (int)cfg->cluster_repeat_h_max ? cfg->cluster_merge_bnd : (int)cfg->cluster_repeat_h_max;
    b->run_gap = g ? atoi(g) : (base > 1000 ? base : 1000);
    b->stream = g_streams.take(b->device);
    b->stream2 = g_streams.take(b->device);
    b->stream3_high = getenv("SNF_LARGE_PRIO") != nullptr && atoi(getenv("SNF_LARGE_PRIO")) != 0;
    b->stream3 = g_streams.take(b->device, b->stream3_high);
    b->stream4 = g_streams.take(b->device);
    SNF_HIP(hipEventCreateWithFlags(&b->ev_join4, hipEventDisableTiming));
    SNF_HIP(hipEventCreateWithFlags(&b->ev_e3, hipEventDisableTiming));
    SNF_HIP(hipEventCreateWithFlags(&b->ev_fork3, hipEventDisableTiming));
    SNF_HIP(hipEventCreate(&b->ev_base));
    SNF_HIP(hipEventCreateWithFlags(&b->ev_counts, hipEventDisableTiming));
    SNF_HIP(hipEventCreateWithFlags(&b->ev_rn, hipEventDisableTiming));
    SNF_HIP(hipEventCreateWithFlags(&b->ev_join3, hipEventDisableTiming));
    SNF_HIP(hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming));
    SNF_HIP(hipEventCreateWithFlags(&b->ev_join, hipEventDisableTiming));
    b->cur = b->stream;
    {  // grid-stride wave kernels (refine, call_from, finalize): TWO resident sets.  Rounds 1-4 launched exactly one (with uniform work per
       // workgroup a partial second round doubles the kernel time); since the grouped kernels take the small clusters, what the
       // wave-per-cluster kernels walk is a list of a few unequal items per wave, and the dispatcher balances those better than a
       // stride does: same box, two alternations, 0.937 ms per step against 0.955 (x4: 0.934; one in flight 1.35 against 1.37)
      const int mult = getenv("SNF_GRID_MULT") ? atoi(getenv("SNF_GRID_MULT")) : 2;
      const int o2 = getenv("SNF_OCC_D2") ? atoi(getenv("SNF_OCC_D2")) : 5;   // <6> and <8> spill (36 / 100 B of scratch); <5> does not and is as fast
      const int o1 = getenv("SNF_OCC_E1") ? atoi(getenv("SNF_OCC_E1")) : 5;
      b->k_d2w = pick_d2w(o2, b->cfg.phase != 0);
      b->k_e1w = o1 == 6 ? e1w_finalize<6> : o1 == 5 ? e1w_finalize<5> : e1w_finalize<4>;  // <8> trips a register-allocation bug of this hipcc
      const DevInfo di = device_info(b->device, o2, o1, b->k_d2w, b->k_e1w);
      const int cus = di.cus;
      int nb = 0;
      const int div = getenv("SNF_GRID_DIV") && atoi(getenv("SNF_GRID_DIV")) > 0 ? atoi(getenv("SNF_GRID_DIV")) : 1;   // experiments: a fraction of the resident set (room for the other pass in flight)
      if (di.nb_d1w > 0) b->slots_d1w = di.nb_d1w * cus * mult / div;
      if (di.nb_d2w > 0) b->slots_d2w = di.nb_d2w * cus * mult / div;
      if (di.nb_e1w > 0) b->slots_e1w = di.nb_e1w * cus * mult / div;
      // The consensus kernels take one call per workgroup from the hardware dispatcher: measured alone on config 1, resident
      // (persistent) grids were slower whether they strode statically (0.315 / 0.419 ms SMALL / LARGE, a tail of unequal calls)
      // or claimed calls from a counter (0.498 / 0.382 ms) - against 0.286 / 0.306 ms for plain grids.  SNF_CONS_GRID_MULT=k
      // caps the grid at k x the resident workgroups (the kernels stride) for experiments.
      b->slots_cons_s = b->slots_cons_l = 1 << 22;
      if (const char* e = getenv("SNF_CONS_GRID_MULT")) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (K_CONS_SMALL(5)), 256, 0) == hipSuccess && nb > 0) b->slots_cons_s = nb * cus * atoi(e);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (K_CONS_LARGE), 256, 0) == hipSuccess && nb > 0) b->slots_cons_l = nb * cus * atoi(e);
      }
      b->slots_cons_s1 = 65536;
      if (const char* e = getenv("SNF_CONS_SMALL_GRID")) { if (atoi(e) > 0) b->slots_cons_s1 = atoi(e); }   // experiments: the one-wave SMALL kernel strides beyond this many workgroups
      b->slots_big = ((32 * cus) / 64) * 64; if (b->slots_big < 64) b->slots_big = 64;   // x_big: a multiple of its 64 stripes
      if (getenv("SNF_PROF")) fprintf(stderr, "[SNF_PROF] resident workgroups: d1w %d d2w %d e1w %d (CUs %d)\n", b->slots_d1w, b->slots_d2w, b->slots_e1w, cus);
    }
    b->timing = getenv("SNF_NO_TIMING") == nullptr;
    b->timeline = getenv("SNF_TIMELINE") != nullptr;
    if (const char* e = getenv("SNF_TIME_EVERY")) b->time_every = atoi(e) < 0 ? 0 : atoi(e);
    b->time_all = getenv("SNF_TIME_ALL") != nullptr || b->timeline;
    if (const char* e = getenv("SNF_OCC_S")) b->occ_s = atoi(e);
    if (const char* e = getenv("SNF_CONS_NW")) b->cons_nw = atoi(e);
    b->d2_groups = getenv("SNF_NO_D2_GROUPS") == nullptr;
    b->d1_groups = getenv("SNF_NO_D1_GROUPS") == nullptr;
    if (const char* e = getenv("SNF_CONS_LARGE_NW")) b->cons_large_nw = atoi(e);
    if (const char* e = getenv("SNF_READPREP")) b->sched_readprep = atoi(e);
    *out = reinterpret_cast<snf_batch_t*>(b.release());
  })
}

int snf_batch_add_task(snf_batch_t* bb, const snf_task_input_t* task) {
  SNF_TRY({ if (!bb || !task) fail("null argument"); do_add_task(reinterpret_cast<snf_batch_impl*>(bb), task); })
}

int snf_batch_add_task_device(snf_batch_t* bb, snf_extract_t* x, const snf_task_input_t* meta) {
  SNF_TRY(do_add_task_device(reinterpret_cast<snf_batch_impl*>(bb), x, meta))
}

int snf_batch_upload(snf_batch_t* bb) {
  SNF_TRY({
    auto b = reinterpret_cast<snf_batch_impl*>(bb);
    if (!b) fail("null batch");
    if (b->uploaded) fail("batch already uploaded");
    SNF_HIP(hipSetDevice(b->device));
    do_upload(b);
  })
}

int snf_batch_open(const snf_config_t* cfg, int device, const snf_task_input_t* tasks, int32_t n_tasks, int run, snf_batch_t** out) {
  if (out) *out = nullptr;
  if (!out || (n_tasks > 0 && !tasks) || n_tasks < 0) { g_err = "snf_batch_open: null argument"; return 1; }
  if (run != SNF_RUN_NONE && run != SNF_RUN_CANDIDATES && !((run & SNF_RUN_PASS) && (run & ~SNF_RUN_PASS) <= (SNF_OUT_EXECUTE | SNF_OUT_DEVICE))) {
    g_err = "snf_batch_open: unknown run mode"; return 1;
  }
  snf_batch_t* h = nullptr;
  int rc = snf_batch_create(cfg, device, &h);
  for (int32_t t = 0; rc == 0 && t < n_tasks; t++) rc = snf_batch_add_task(h, &tasks[t]);
  if (rc == 0) rc = snf_batch_upload(h);
  if (rc == 0 && run == SNF_RUN_CANDIDATES) rc = snf_batch_call_candidates(h);
  if (rc == 0 && (run & SNF_RUN_PASS)) { rc = snf_batch_set_output(h, run & ~SNF_RUN_PASS); if (rc == 0) rc = snf_batch_pass(h); }
  if (rc != 0) { const std::string keep = g_err; if (h) snf_batch_destroy(h); g_err = keep; return rc; }      // (the first error is the one reported)
  *out = h;
  return 0;
}

void snf_batch_destroy(snf_batch_t* bb) {
  auto b = reinterpret_cast<snf_batch_impl*>(bb);
  if (!b) return;
  (void)hipSetDevice(b->device);
  if (b->stream) (void)hipStreamSynchronize(b->stream);
  if (b->stream2) (void)hipStreamSynchronize(b->stream2);
  if (b->stream3) (void)hipStreamSynchronize(b->stream3);
  if (b->stream4) (void)hipStreamSynchronize(b->stream4);
  if (b->ev_fork3) (void)hipEventDestroy(b->ev_fork3);
  if (b->ev_e3) (void)hipEventDestroy(b->ev_e3);
  if (b->ev_base) (void)hipEventDestroy(b->ev_base);
  if (b->ev_counts) (void)hipEventDestroy(b->ev_counts);
  if (b->ev_rn) (void)hipEventDestroy(b->ev_rn);
  if (b->ev_join3) (void)hipEventDestroy(b->ev_join3);
  if (b->ev_fork) (void)hipEventDestroy(b->ev_fork);
  if (b->ev_join) (void)hipEventDestroy(b->ev_join);
  for (auto& e : b->evs) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  destroy_graphs(b);
  pass_waited(b);
  dfree_all(b);
  b->hb_calls.release(); b->hb_out.release(); b->hb_alt.release(); b->hb_rn.release(); b->hb_res.release();
  for (auto& e : b->ext_ranges) (void)hipHostUnregister(e.first);
  b->ext_ranges.clear();
  if (b->stream) g_streams.give(b->device, b->stream);   // (synchronised above)
  if (b->stream2) g_streams.give(b->device, b->stream2);   // (synchronised above)
  if (b->stream3) g_streams.give(b->device, b->stream3, b->stream3_high);   // (synchronised above)
  if (b->stream4) g_streams.give(b->device, b->stream4);   // (synchronised above)
  if (b->ev_join4) (void)hipEventDestroy(b->ev_join4);
  delete b;
}

int snf_batch_call_candidates(snf_batch_t* bb) {
  SNF_TRY({
    auto b = reinterpret_cast<snf_batch_impl*>(bb);
    if (!b || !b->uploaded) fail("batch not uploaded");
    SNF_HIP(hipSetDevice(b->device));
    run_call_candidates(b);
  })
}

int snf_batch_pass(snf_batch_t* bb) {
  SNF_TRY({
    auto b = reinterpret_cast<snf_batch_impl*>(bb);
    if (!b || !b->uploaded) fail("batch not uploaded");
    SNF_HIP(hipSetDevice(b->device));
    run_pass(b);
  })
}

int snf_batch_finalize(snf_batch_t* bb) {
  SNF_TRY({
    auto b = reinterpret_cast<snf_batch_impl*>(bb);
    if (!b || !b->uploaded) fail("batch not uploaded");
    SNF_HIP(hipSetDevice(b->device));
    run_finalize(b);
  })
}

int snf_batch_fetch(snf_batch_t* bb, int stage, snf_result_t* out) {
  SNF_TRY({
    auto b = reinterpret_cast<snf_batch_impl*>(bb);
    if (!b || !b->uploaded || !out) fail("batch not uploaded / null result");
    SNF_HIP(hipSetDevice(b->device));
    do_fetch(b, stage, out);
  })
}

int snf_genotype_batch(const snf_config_t* cfg, int device, snf_call_t* calls, int64_t n) {
  SNF_TRY(do_genotype_batch(cfg, device, calls, n))
}

int snf_batch_fetch_clusters(snf_batch_t* bb, int stage, snf_clusters_t* out) {
  SNF_TRY({
    auto b = reinterpret_cast<snf_batch_impl*>(bb);
    if (!b || !b->uploaded || !out) fail("batch not uploaded / null result");
    SNF_HIP(hipSetDevice(b->device));
    do_fetch_clusters(b, stage, out);
  })
}

int64_t snf_batch_n_candidates(snf_batch_t* bb) {
  auto b = reinterpret_cast<snf_batch_impl*>(bb);
  if (!b || !b->uploaded || !b->h_cnt) return -1;
  return b->v.NS > 0 ? b->h_cnt->n_calls : 0;
}

int64_t snf_trim_caches(int device) {
  int64_t bytes = (int64_t)g_slabs.trim(device);
  bytes += (int64_t)g_pinned.trim();
  (void)g_streams.trim(device);
  return bytes;
}

int snf_batch_set_output(snf_batch_t* bb, int mode) {
  SNF_TRY({
    auto b = reinterpret_cast<snf_batch_impl*>(bb);
    if (!b) fail("null batch");
    if (mode < 0 || mode > (SNF_OUT_EXECUTE | SNF_OUT_DEVICE)) fail("unknown output mode");
    b->out_mode = mode; b->v.out_mode = mode;
  })
}

int snf_batch_set_result_memory(snf_batch_t* bb, void* block, int64_t block_bytes, void* alt, int64_t alt_bytes) {
  SNF_TRY({
    auto b = reinterpret_cast<snf_batch_impl*>(bb);
    if (!b) fail("null batch");
    if ((block == nullptr) != (alt == nullptr) || block_bytes < 0 || alt_bytes < 0) fail("snf_batch_set_result_memory: both sections or none");
    SNF_HIP(hipSetDevice(b->device));
    if (b->uploaded && !b->pass_idle) full_sync(b);     // (no pass of this batch is writing the old buffers)
    if (!block) { b->hb_out.release(); b->hb_alt.release(); }
    else { b->hb_out.adopt(block, (size_t)block_bytes, b->ext_ranges); b->hb_alt.adopt(alt, (size_t)alt_bytes, b->ext_ranges); }
  })
}

int snf_batch_export_device(snf_batch_t* bb, void* dst_device, int64_t cap_bytes, snf_export_layout_t* layout) {
  SNF_TRY({
    auto b = reinterpret_cast<snf_batch_impl*>(bb);
    if (!b || !b->uploaded || !layout) fail("batch not uploaded / null argument");
    if (!b->finalized) fail("snf_batch_export_device needs snf_batch_finalize first");
    if (!(b->v.out_mode & SNF_OUT_DEVICE)) fail("snf_batch_export_device needs snf_batch_set_output(... | SNF_OUT_DEVICE) before the finalize");
    SNF_HIP(hipSetDevice(b->device));
    full_sync(b);
    settle_alt_stage(b);
    const OutHdr h = *b->v.res_out;
    const int64_t alt_total = b->h_cnt->alt_total;
    layout->n_calls = h.n_out; layout->rnames_len = h.rn_out; layout->alt_pool_len = alt_total;
    layout->off_rnames = h.off_rn; layout->off_alt = h.bytes; layout->bytes = h.bytes + alt_total;
    if (layout->bytes > cap_bytes) fail("export buffer too small");
    if (layout->bytes > 0 && !dst_device) fail("null destination");
    d2d(b, dst_device, b->v.out_dev, (size_t)h.bytes);
    d2d(b, (uint8_t*)dst_device + h.bytes, b->v.alt_pool, (size_t)alt_total);
    dsync(b);
  })
}

int snf_batch_block_coverage(snf_batch_t* bb, int32_t task_index, int32_t binsize, int64_t first_bin, int64_t n_bins, int32_t* out) {
  SNF_TRY({
    auto b = reinterpret_cast<snf_batch_impl*>(bb);
    if (!b || !b->uploaded || !out) fail("batch not uploaded / null argument");
    if (!b->reads_ready) fail("snf_batch_block_coverage needs snf_batch_upload first");
    if (task_index < 0 || task_index >= b->v.T) fail("task index out of range");
    if (binsize <= 0 || first_bin < 0 || n_bins < 0) fail("invalid bin range");
    if (n_bins == 0) return 0;
    SNF_HIP(hipSetDevice(b->device));
    full_sync(b);
    BlockCov q{};
    q.r_start = b->v.r_start; q.re_sorted = b->v.re_sorted; q.rs_top = b->v.rs_top; q.re_top = b->v.re_top;
    q.lo = b->h_read_off[(size_t)task_index]; q.hi = b->h_read_off[(size_t)task_index + 1];
    q.L = b->tasks[(size_t)task_index].contig_len; q.first_bin = first_bin; q.binsize = binsize;
    q.nm_start = b->v.nm_start; q.nm_end = b->v.nm_end; q.nm_lo = b->h_nm_off[(size_t)task_index]; q.nm_hi = b->h_nm_off[(size_t)task_index + 1];
    q.exact = b->h_cov_exact.empty() ? 0 : b->h_cov_exact[(size_t)task_index];
    q.out = dalloc_own<int32_t>(b, (size_t)n_bins);
    LAUNCH(s1_blockcov, q, n_bins, (q.hi - q.lo) * 8 + n_bins * 4);
    d2h(b, out, q.out, (size_t)n_bins * 4);
    dsync(b);
    dfree_one(b, q.out);
  })
}

int snf_batch_coverage_calls(snf_batch_t* bb, int32_t task_index, int64_t n, const int32_t* svtype, const int32_t* pos,
                             const int32_t* svlen, const uint8_t* bnd_is_first, int32_t* cov, int32_t* status, double* coverage_mean) {
  SNF_TRY(do_coverage_calls(bb, task_index, n, svtype, pos, svlen, bnd_is_first, cov, status, coverage_mean))
}

int snf_consensus_batch(int device, int klen, const uint8_t* seq_pool, int64_t seq_pool_len, int64_t n_problems,
                        const int64_t* best_off, const int32_t* best_len, const int32_t* skip, const int32_t* skip_repetitive,
                        const int64_t* others_index, const int64_t* others_off, const int32_t* others_len, uint8_t* out_pool,
                        const int64_t* out_off) {
  SNF_TRY(do_consensus_batch(device, klen, seq_pool, seq_pool_len, n_problems, best_off, best_len, skip, skip_repetitive, others_index,
                              others_off, others_len, out_pool, out_off))
}


int snf_batch_sync(snf_batch_t* bb) {
  SNF_TRY({ auto b = reinterpret_cast<snf_batch_impl*>(bb); if (!b) fail("null batch"); if (b->uploaded) full_sync(b); else dsync(b); pass_waited(b); collect_timings(b); })
}

int snf_batch_timing_count(snf_batch_t* bb) {
  auto b = reinterpret_cast<snf_batch_impl*>(bb);
  return b ? (int)b->timings.size() : 0;
}

int snf_batch_timing_every(snf_batch_t* bb, int n) {
  auto b = reinterpret_cast<snf_batch_impl*>(bb);
  if (!b) return 1;
  b->time_every = n < 0 ? 0 : n;
  return 0;
}
int snf_batch_timing_mean_reset(snf_batch_t* bb) {
  auto b = reinterpret_cast<snf_batch_impl*>(bb);
  if (!b) return 1;
  b->timing_acc.clear();
  return 0;
}
int snf_batch_timing_mean_count(snf_batch_t* bb) {
  auto b = reinterpret_cast<snf_batch_impl*>(bb);
  return b ? (int)b->timing_acc.size() : 0;
}
int snf_batch_timing_mean_get(snf_batch_t* bb, int i, const char** name, float* ms, int64_t* algo_bytes, int* passes) {
  auto b = reinterpret_cast<snf_batch_impl*>(bb);
  if (!b || i < 0 || i >= (int)b->timing_acc.size()) { g_err = "timing index out of range"; return 1; }
  const Timing& t = b->timing_acc[(size_t)i];
  if (name) *name = t.name;
  if (ms) *ms = t.launches > 0 ? t.ms / (float)t.launches : 0.0f;
  if (algo_bytes) *algo_bytes = t.bytes;
  if (passes) *passes = t.launches;
  return 0;
}

int snf_batch_timing_get(snf_batch_t* bb, int i, const char** name, float* ms, int64_t* algo_bytes) {
  auto b = reinterpret_cast<snf_batch_impl*>(bb);
  if (!b || i < 0 || i >= (int)b->timings.size()) { g_err = "timing index out of range"; return 1; }
  if (name) *name = b->timings[i].name;
  if (ms) *ms = b->timings[i].ms;
  if (algo_bytes) *algo_bytes = b->timings[i].bytes;
  return 0;
}

}  // extern "C"
