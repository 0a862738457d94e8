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

namespace snf {

struct EdView {
  const uint8_t* a; const int64_t* a_off; const uint8_t* b; const int64_t* b_off;
  int32_t* out; int8_t* carry; const int64_t* carry_off;   // carry: one byte per text column per pair
  const int32_t* list; int64_t n;                           // pair indices handled by this launch
};

SNF_HD void pair_strings(const EdView& v, int64_t pi, const uint8_t** P, int64_t* m, const uint8_t** T, int64_t* n) {
  const uint8_t* A = v.a + v.a_off[pi]; int64_t la = v.a_off[pi + 1] - v.a_off[pi];
  const uint8_t* B = v.b + v.b_off[pi]; int64_t lb = v.b_off[pi + 1] - v.b_off[pi];
  if (la <= lb) { *P = A; *m = la; *T = B; *n = lb; } else { *P = B; *m = lb; *T = A; *n = la; }
}

// thread per pair, block-major
SNF_HD void ed_thread_body(int64_t i, const EdView& v) {
  int64_t pi = v.list[i];
  const uint8_t* A = v.a + v.a_off[pi]; int64_t la = v.a_off[pi + 1] - v.a_off[pi];
  const uint8_t* B = v.b + v.b_off[pi]; int64_t lb = v.b_off[pi + 1] - v.b_off[pi];
  v.out[pi] = (int32_t)ed_serial(A, la, B, lb, v.carry + v.carry_off[pi]);
}

}  // namespace snf
using namespace snf;
SNF_KERNEL(ed_thread, EdView)

#ifndef SNF_EMU
// wave per pair: lane = block of the current 64-block pass, anti-diagonal schedule
__global__ void __launch_bounds__(64) ed_wave(const EdView v, int64_t n_items) {
  const int lane = threadIdx.x;
  for (int64_t it = blockIdx.x; it < n_items; it += gridDim.x) {
    const int64_t pi = v.list[it];
    const uint8_t *P, *T; int64_t m, n;
    pair_strings(v, pi, &P, &m, &T, &n);
    const int64_t d = ed_wave_pair(P, m, T, n, v.carry + v.carry_off[pi]);
    if (lane == 0) v.out[pi] = (int32_t)d;
  }
}
#endif

namespace {
thread_local std::string g_ed_err;
}

extern "C" int snf_edit_distance_batch(int device, const uint8_t* a_pool, const int64_t* a_off, const uint8_t* b_pool,
                                       const int64_t* b_off, int64_t n_pairs, int32_t* out_dist) {
  if (n_pairs <= 0) return 0;
  const int64_t la = a_off[n_pairs], lb = b_off[n_pairs];
  std::vector<int64_t> carry_off((size_t)n_pairs + 1, 0);
  std::vector<int32_t> thread_list, wave_list;
  for (int64_t i = 0; i < n_pairs; i++) {
    int64_t x = a_off[i + 1] - a_off[i], y = b_off[i + 1] - b_off[i];
    int64_t m = x < y ? x : y, n = x < y ? y : x;
    carry_off[i + 1] = carry_off[i] + n;
    if ((m + 63) / 64 <= 8) thread_list.push_back((int32_t)i); else wave_list.push_back((int32_t)i);
  }
  EdView v{};
#ifndef SNF_EMU
  int nd = 0;
  if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0 || device < 0 || device >= nd) return 1;
  if (hipSetDevice(device) != hipSuccess) return 1;
  uint8_t *da = nullptr, *db = nullptr; int64_t *dao = nullptr, *dbo = nullptr, *dco = nullptr; int32_t *dout = nullptr, *dlist = nullptr;
  int8_t* dcarry = nullptr;
  bool ok = true;
  auto chk = [&](hipError_t e) { if (e != hipSuccess) ok = false; };
  chk(hipMalloc(&da, la + 16)); chk(hipMalloc(&db, lb + 16));
  chk(hipMalloc(&dao, (n_pairs + 1) * 8)); chk(hipMalloc(&dbo, (n_pairs + 1) * 8)); chk(hipMalloc(&dco, (n_pairs + 1) * 8));
  chk(hipMalloc(&dout, n_pairs * 4)); chk(hipMalloc(&dlist, n_pairs * 4 + 4)); chk(hipMalloc(&dcarry, carry_off[n_pairs] + 16));
  if (ok) {
    chk(hipMemcpy(da, a_pool, la, hipMemcpyHostToDevice)); chk(hipMemcpy(db, b_pool, lb, hipMemcpyHostToDevice));
    chk(hipMemcpy(dao, a_off, (n_pairs + 1) * 8, hipMemcpyHostToDevice)); chk(hipMemcpy(dbo, b_off, (n_pairs + 1) * 8, hipMemcpyHostToDevice));
    chk(hipMemcpy(dco, carry_off.data(), (n_pairs + 1) * 8, hipMemcpyHostToDevice));
    v.a = da; v.a_off = dao; v.b = db; v.b_off = dbo; v.out = dout; v.carry = dcarry; v.carry_off = dco;
    if (!thread_list.empty()) {
      chk(hipMemcpy(dlist, thread_list.data(), thread_list.size() * 4, hipMemcpyHostToDevice));
      v.list = dlist; v.n = (int64_t)thread_list.size();
      hipLaunchKernelGGL(ed_thread, dim3((unsigned)((v.n + 255) / 256)), dim3(256), 0, 0, v, v.n);
      chk(hipDeviceSynchronize());
    }
    if (!wave_list.empty()) {
      chk(hipMemcpy(dlist, wave_list.data(), wave_list.size() * 4, hipMemcpyHostToDevice));
      v.list = dlist; v.n = (int64_t)wave_list.size();
      int64_t grid = v.n < 16384 ? v.n : 16384;
      hipLaunchKernelGGL(ed_wave, dim3((unsigned)grid), dim3(64), 0, 0, v, v.n);
      chk(hipDeviceSynchronize());
    }
    chk(hipMemcpy(out_dist, dout, n_pairs * 4, hipMemcpyDeviceToHost));
  }
  hipFree(da); hipFree(db); hipFree(dao); hipFree(dbo); hipFree(dco); hipFree(dout); hipFree(dlist); hipFree(dcarry);
  return ok ? 0 : 1;
#else
  (void)device; (void)la; (void)lb;
  std::vector<int8_t> carry((size_t)carry_off[n_pairs] + 16);
  std::vector<int32_t> all;
  all.insert(all.end(), thread_list.begin(), thread_list.end());
  all.insert(all.end(), wave_list.begin(), wave_list.end());
  v.a = a_pool; v.a_off = a_off; v.b = b_pool; v.b_off = b_off; v.out = out_dist; v.carry = carry.data();
  v.carry_off = carry_off.data(); v.list = all.data(); v.n = (int64_t)all.size();
  ed_thread(v, v.n);
  return 0;
#endif
}
