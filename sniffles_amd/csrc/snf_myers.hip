// snf_myers.hip - batched global (NW) unit-cost edit distance, the quantity SVGroup.align_call reads from
// edlib.align(a, b)["editDistance"] (sv.py:280-289; snfp.py:103).  First version: one thread per pair,
// two-row DP; to be replaced by the bit-parallel Myers/Hyyro kernel.
#include "snf_exact.h"
#include "../../include/sniffles_amd.h"

namespace snf {
struct EdView {
  const uint8_t* a; const int64_t* a_off; const uint8_t* b; const int64_t* b_off; int64_t n; int32_t* out;
  int32_t* row; const int64_t* row_off;
};
SNF_HD void ed_pair_body(int64_t i, const EdView& v) {
  const uint8_t* A = v.a + v.a_off[i]; int64_t la = v.a_off[i + 1] - v.a_off[i];
  const uint8_t* B = v.b + v.b_off[i]; int64_t lb = v.b_off[i + 1] - v.b_off[i];
  int32_t* row = v.row + v.row_off[i];
  for (int64_t j = 0; j <= lb; j++) row[j] = (int32_t)j;
  for (int64_t x = 1; x <= la; x++) {
    int32_t diag = row[0];
    row[0] = (int32_t)x;
    for (int64_t j = 1; j <= lb; j++) {
      int32_t up = row[j];
      int32_t vv = diag + (A[x - 1] != B[j - 1]);
      if (up + 1 < vv) vv = up + 1;
      if (row[j - 1] + 1 < vv) vv = row[j - 1] + 1;
      diag = up; row[j] = vv;
    }
  }
  v.out[i] = row[lb];
}
}  // namespace snf
using namespace snf;
SNF_KERNEL(ed_pair, EdView)

extern "C" int snf_edit_distance_batch(int device, const uint8_t* a_pool, const int64_t* a_off, const uint8_t* b_pool,
                                       const int64_t* b_off, int64_t n_pairs, int32_t* out_dist) {
  if (n_pairs <= 0) return 0;
  int64_t la = a_off[n_pairs], lb = b_off[n_pairs];
  std::vector<int64_t> row_off((size_t)n_pairs + 1, 0);
  for (int64_t i = 0; i < n_pairs; i++) row_off[i + 1] = row_off[i] + (b_off[i + 1] - b_off[i]) + 1;
  EdView v{};
  v.n = n_pairs;
#ifndef SNF_EMU
  int nd = 0;
  if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0 || device >= nd) return 1;
  if (hipSetDevice(device) != hipSuccess) return 1;
  uint8_t *da, *db; int64_t *dao, *dbo, *dro; int32_t *dout, *drow;
  hipMalloc(&da, la + 1); hipMalloc(&db, lb + 1); hipMalloc(&dao, (n_pairs + 1) * 8); hipMalloc(&dbo, (n_pairs + 1) * 8);
  hipMalloc(&dro, (n_pairs + 1) * 8); hipMalloc(&dout, n_pairs * 4); hipMalloc(&drow, row_off[n_pairs] * 4 + 4);
  hipMemcpy(da, a_pool, la, hipMemcpyHostToDevice); hipMemcpy(db, b_pool, lb, hipMemcpyHostToDevice);
  hipMemcpy(dao, a_off, (n_pairs + 1) * 8, hipMemcpyHostToDevice); hipMemcpy(dbo, b_off, (n_pairs + 1) * 8, hipMemcpyHostToDevice);
  hipMemcpy(dro, row_off.data(), (n_pairs + 1) * 8, hipMemcpyHostToDevice);
  v.a = da; v.a_off = dao; v.b = db; v.b_off = dbo; v.out = dout; v.row = drow; v.row_off = dro;
  hipLaunchKernelGGL(ed_pair, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, 0, v, n_pairs);
  hipError_t e = hipMemcpy(out_dist, dout, n_pairs * 4, hipMemcpyDeviceToHost);
  hipFree(da); hipFree(db); hipFree(dao); hipFree(dbo); hipFree(dro); hipFree(dout); hipFree(drow);
  return e == hipSuccess ? 0 : 1;
#else
  (void)device; (void)la; (void)lb;
  std::vector<int32_t> row((size_t)row_off[n_pairs] + 1);
  v.a = a_pool; v.a_off = a_off; v.b = b_pool; v.b_off = b_off; v.out = out_dist; v.row = row.data(); v.row_off = row_off.data();
  ed_pair(v, n_pairs);
  return 0;
#endif
}
