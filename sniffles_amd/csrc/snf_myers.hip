// snf_myers.hip - batched global (NW) unit-cost edit distance: the quantity SVGroup.align_call reads from
// edlib.align(a, b)["editDistance"] with edlib's defaults (sv.py:280-289; snfp.py:103).
//
// Bit-parallel Myers / Hyyro block algorithm: the shorter string is the pattern, cut into blocks of 64 rows;
// each block keeps the vertical delta vectors (Pv, Mv) and advances one text column in ~25 ALU ops given the
// match mask Eq(c).  Eq for an arbitrary byte alphabet comes from the 8 bit-planes of the block's 64 pattern
// bytes (16 ops), so no per-pair alphabet table is needed.  Blocks of one column are chained by the horizontal
// delta hout -> hin (global alignment: hin = +1 at row 0).
//   * patterns of <= 8 blocks: one THREAD per pair, block-major, per-column carries in a byte array
//   * longer patterns: one WAVE per pair, lane = block, lanes advance along anti-diagonals (lane l works on
//     column t - l at step t and takes hin from lane l-1 by a shuffle); patterns of > 64 blocks take several
//     passes of 64 blocks linked through the same carry array.
// No MFMA: this is integer/bit work; the bound is VALU issue, reported as such.
#include "snf_myers.h"
#include "../../include/sniffles_amd.h"

#include <string>
#include <vector>

#include "snf_ctx.h"

namespace snf {

struct EdView {
  const uint8_t* a; const int64_t* a_off; const uint8_t* b; const int64_t* b_off;
  int32_t* out;
  const int32_t* kmax;                                       // per pair: cut-off (result -1 beyond it), < 0 none; may be null
  uint64_t* scratch; const int64_t* scratch_off;             // serial form: bit-planes + (Pv, Mv) per block (ed_serial_scratch_words)
  int8_t* carry; const int64_t* carry_off;                   // wide bands only (multi-pass wave form): one byte per text column
  const int32_t* list; int64_t n;                            // pair indices handled by this launch
};

SNF_HD void pair_strings(const EdView& v, int64_t pi, const uint8_t** P, int64_t* m, const uint8_t** T, int64_t* n) {
  const uint8_t* A = v.a + v.a_off[pi]; int64_t la = v.a_off[pi + 1] - v.a_off[pi];
  const uint8_t* B = v.b + v.b_off[pi]; int64_t lb = v.b_off[pi + 1] - v.b_off[pi];
  if (la <= lb) { *P = A; *m = la; *T = B; *n = lb; } else { *P = B; *m = lb; *T = A; *n = la; }
}

// thread per pair, column-major over the band (emulation form; on the GPU ed_thread_lds keeps the block states in LDS)
SNF_HD void ed_thread_body(int64_t i, const EdView& v) {
  int64_t pi = v.list[i];
  const uint8_t* A = v.a + v.a_off[pi]; int64_t la = v.a_off[pi + 1] - v.a_off[pi];
  const uint8_t* B = v.b + v.b_off[pi]; int64_t lb = v.b_off[pi + 1] - v.b_off[pi];
  v.out[pi] = (int32_t)ed_serial_k(A, la, B, lb, v.kmax ? (int64_t)v.kmax[pi] : -1, v.scratch + v.scratch_off[pi]);
}

}  // namespace snf
using namespace snf;
SNF_KERNEL(ed_thread, EdView)

// thread per pair, patterns of at most SNF_ED_THREAD_BLOCKS blocks: bit-planes and (Pv, Mv) of every block in LDS
#define SNF_ED_THREAD_BLOCKS 8
__global__ void __launch_bounds__(64) ed_thread_lds(const EdView v, int64_t n_items) {
  __shared__ uint64_t scr[64][11 * SNF_ED_THREAD_BLOCKS + 11];
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n_items) return;
  const int64_t pi = v.list[i];
  const uint8_t* A = v.a + v.a_off[pi]; int64_t la = v.a_off[pi + 1] - v.a_off[pi];
  const uint8_t* B = v.b + v.b_off[pi]; int64_t lb = v.b_off[pi + 1] - v.b_off[pi];
  v.out[pi] = (int32_t)ed_serial_k(A, la, B, lb, v.kmax ? (int64_t)v.kmax[pi] : -1, scr[threadIdx.x]);
}
// wave per pair: lane = block of the band (registers only); bands of more than 63 blocks take the multi-pass form
__global__ void __launch_bounds__(64) ed_wave(const EdView v, int64_t n_items) {
  const int lane = threadIdx.x;
  for (int64_t it = blockIdx.x; it < n_items; it += gridDim.x) {
    const int64_t pi = v.list[it];
    const uint8_t *P, *T; int64_t m, n;
    pair_strings(v, pi, &P, &m, &T, &n);
    const int64_t k = v.kmax ? (int64_t)v.kmax[pi] : -1;
    int64_t d;
    if (ed_wave_band_fits(m, n, k)) d = ed_wave_pair_k_any(P, m, T, n, k);
    else { d = ed_wave_pair(P, m, T, n, v.carry + v.carry_off[pi]); if (k >= 0 && d > k) d = -1; }
    if (lane == 0) v.out[pi] = (int32_t)d;
  }
}

namespace {
DevArena g_ed_arenas[SNF_MAX_DEVICES];
}

// pairs: strings a_pool[a_off[i] .. a_off[i+1]) and b_pool[b_off[i] .. b_off[i+1]); max_dist (may be null): per pair the
// largest distance of interest - beyond it the result is -1 (edlib's `k`), < 0: exact
static int ed_batch(int device, const uint8_t* a_pool, const int64_t* a_off, const uint8_t* b_pool, const int64_t* b_off,
                    int64_t n_pairs, const int32_t* max_dist, int32_t* out_dist) {
  if (n_pairs <= 0) return 0;
  if (!a_pool || !a_off || !b_pool || !b_off || !out_dist) return 1;
  const int64_t la = a_off[n_pairs], lb = b_off[n_pairs];
  std::vector<int64_t> carry_off((size_t)n_pairs + 1, 0), scratch_off((size_t)n_pairs + 1, 0);
  std::vector<int32_t> thread_list, wave_list;
  for (int64_t i = 0; i < n_pairs; i++) {
    int64_t x = a_off[i + 1] - a_off[i], y = b_off[i + 1] - b_off[i];
    if (x < 0 || y < 0) return 1;
    int64_t m = x < y ? x : y, n = x < y ? y : x;
    const bool small = (m + 63) / 64 <= 8;
    EdBand bd;
    const int64_t k = max_dist ? (int64_t)max_dist[i] : -1;
    bool wide = false;   // needs the multi-pass wave form (per-column carry bytes)
    if (!small && ed_band(m, n, k, &bd)) wide = !((64 + bd.dl + 2 * bd.kk) / 64 + 2 <= 63 || (m + 63) / 64 <= 63);
    carry_off[i + 1] = carry_off[i] + (wide ? n : 0);
    scratch_off[i + 1] = scratch_off[i];
    if (small) thread_list.push_back((int32_t)i); else wave_list.push_back((int32_t)i);
  }
  if (device < 0 || device >= SNF_MAX_DEVICES) return 1;
  DevArena& g_ed_arena = g_ed_arenas[device];
  std::lock_guard<std::mutex> hold(g_ed_arena.mu);
  ArenaLayout L;
  const size_t o_a = L.add<uint8_t>((size_t)la + 16), o_b = L.add<uint8_t>((size_t)lb + 16);
  const size_t o_ao = L.add<int64_t>((size_t)n_pairs + 1), o_bo = L.add<int64_t>((size_t)n_pairs + 1);
  const size_t o_co = L.add<int64_t>((size_t)n_pairs + 1), o_so = L.add<int64_t>((size_t)n_pairs + 1);
  const size_t o_k = L.add<int32_t>((size_t)n_pairs), o_tl = L.add<int32_t>(thread_list.size()), o_wl = L.add<int32_t>(wave_list.size());
  const size_t in_end = L.at;
  const size_t o_out = L.add<int32_t>((size_t)n_pairs);
  const size_t o_carry = L.add<int8_t>((size_t)carry_off[n_pairs] + 16), o_scr = L.add<uint64_t>((size_t)scratch_off[n_pairs] + 16);
  if (!g_ed_arena.ensure(device, L.at)) return 1;
  uint8_t *h = g_ed_arena.h, *d = g_ed_arena.d;
  memcpy(h + o_a, a_pool, (size_t)la); memset(h + o_a + la, 0, 16);
  memcpy(h + o_b, b_pool, (size_t)lb); memset(h + o_b + lb, 0, 16);
  memcpy(h + o_ao, a_off, ((size_t)n_pairs + 1) * 8); memcpy(h + o_bo, b_off, ((size_t)n_pairs + 1) * 8);
  memcpy(h + o_co, carry_off.data(), ((size_t)n_pairs + 1) * 8); memcpy(h + o_so, scratch_off.data(), ((size_t)n_pairs + 1) * 8);
  if (max_dist) memcpy(h + o_k, max_dist, (size_t)n_pairs * 4);
  if (!thread_list.empty()) memcpy(h + o_tl, thread_list.data(), thread_list.size() * 4);
  if (!wave_list.empty()) memcpy(h + o_wl, wave_list.data(), wave_list.size() * 4);
  EdView v{};
  v.a = d + o_a; v.a_off = (const int64_t*)(d + o_ao); v.b = d + o_b; v.b_off = (const int64_t*)(d + o_bo);
  v.out = (int32_t*)(d + o_out); v.kmax = max_dist ? (const int32_t*)(d + o_k) : nullptr;
  v.scratch = (uint64_t*)(d + o_scr); v.scratch_off = (const int64_t*)(d + o_so);
  v.carry = (int8_t*)(d + o_carry); v.carry_off = (const int64_t*)(d + o_co);
  hipStream_t st = g_ed_arena.stream;
  bool ok = hipMemcpyAsync(d, h, in_end, hipMemcpyHostToDevice, st) == hipSuccess;
  if (ok && !thread_list.empty()) {
    v.list = (const int32_t*)(d + o_tl); v.n = (int64_t)thread_list.size();
    hipLaunchKernelGGL(ed_thread_lds, dim3((unsigned)((v.n + 63) / 64)), dim3(64), 0, st, v, v.n);
  }
  if (ok && !wave_list.empty()) {
    v.list = (const int32_t*)(d + o_wl); v.n = (int64_t)wave_list.size();
    hipLaunchKernelGGL(ed_wave, dim3((unsigned)(v.n < 16384 ? v.n : 16384)), dim3(64), 0, st, v, v.n);
  }
  ok = ok && hipGetLastError() == hipSuccess;
  ok = ok && hipMemcpyAsync(h + o_out, d + o_out, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, st) == hipSuccess;
  ok = ok && hipStreamSynchronize(st) == hipSuccess;
  if (!ok) return 1;
  memcpy(out_dist, h + o_out, (size_t)n_pairs * 4);
  return 0;
}

extern "C" int snf_edit_distance_batch(int device, const uint8_t* a_pool, const int64_t* a_off, const uint8_t* b_pool,
                                       const int64_t* b_off, int64_t n_pairs, int32_t* out_dist) {
  return ed_batch(device, a_pool, a_off, b_pool, b_off, n_pairs, nullptr, out_dist);
}

extern "C" int snf_edit_distance_batch_k(int device, const uint8_t* a_pool, const int64_t* a_off, const uint8_t* b_pool,
                                         const int64_t* b_off, int64_t n_pairs, const int32_t* max_dist, int32_t* out_dist) {
  return ed_batch(device, a_pool, a_off, b_pool, b_off, n_pairs, max_dist, out_dist);
}
