// TEST-ONLY: the gfx950 workgroup consensus kernels (sniffles_amd/csrc/snf_wave_cons.h: e45w_consensus SMALL / LARGE /
// ROWS, e4c_copy) executed on the host through the fibre shim in hip/hip_runtime.h.  The set-up mirrors the standalone
// entry point of the product (snf_consensus_batch, snf_lib.hip do_consensus_batch) with malloc in place of hipMalloc; the
// kernels themselves are the product's template instances, compiled from the product's header.
//   mode 0: every call in the class the product picks (cons_class_of);
//   mode 2: SMALL calls are given to the LARGE instance as well (its limits contain SMALL's);
//   mode 4: every call through the ROWS instance (aligned rows in memory, no LDS vote counters).
#include "snf_wave_cons.h"

#include <vector>

using namespace snf;

#define K_CONS_SMALL e45w_consensus<1, 256, 128, 64, 5, 4, SNF_CONS_SMALL_L, 448, 96>
#define K_CONS_SMALL_1W e45w_consensus<1, 256, 128, 64, 5, 1, SNF_CONS_SMALL_L, 448, 96>
#define K_CONS_LARGE e45w_consensus<2, 1024, 512, 256, 2, 4, SNF_CONS_LARGE_L, 0, 512>
#define K_CONS_LARGE_8W e45w_consensus<2, 1024, 512, 256, 2, 8, SNF_CONS_LARGE_L, 0, 512>
#define K_CONS_ROWS e45w_consensus<4, 1024, 512, 512, 3>

static char g_err[512];
extern "C" const char* simt_last_error() { return g_err; }

// the column vote of the LDS-vote instances on its own (tests/test_emu_parity.py pins it on the plain rule)
extern "C" int snf_emu_vote_column(uint32_t cnt4, const uint32_t* esc, int n_esc, int q, int bq, int nkept) {
  return (int)vote_column(cnt4, esc, n_esc, q, (uint8_t)bq, nkept);
}

// cls_out[p]: class the call ran in (1 SMALL, 2 LARGE, 4 ROWS, 0 verbatim copy); handed_over: calls SMALL / LARGE passed on to ROWS
extern "C" int simt_consensus_batch(int mode, int nw, int grid_cap, int min_reads, int klen, const uint8_t* seq_pool, int64_t seq_pool_len,
                                    int64_t np, const int64_t* best_off, const int32_t* best_len, const int32_t* skip,
                                    const int64_t* others_index, const int64_t* others_off, const int32_t* others_len,
                                    uint8_t* out_pool, const int64_t* out_off, int32_t* cls_out, int64_t* handed_over) {
  g_err[0] = 0;
  try {
    const int64_t n_reads = others_index[np];
    std::vector<ConsDesc> descs((size_t)np);
    std::vector<int32_t> lists[8];
    Counts hc{};
    int64_t aln_total = 0, alt_total = 0;
    for (int64_t p = 0; p < np; p++) {
      const int64_t L = best_len[p], no = others_index[p + 1] - others_index[p];
      if (out_off[p + 1] - out_off[p] != L) fail("out_off must be the prefix sums of best_len");
      int cls = cons_class_of(1, klen, skip[p], L, (int32_t)no);
      if (cls == 0) fail("problem exceeds the limits of the workgroup consensus kernels");
      if (mode == 2 && cls == 1) cls = 2;
      if (mode == 4) cls = 4;
      if (no < min_reads) cls = 0;   // verbatim copy (postprocessing.py:65-66)
      ConsDesc d{};
      d.best_off = best_off[p]; d.alt_off = out_off[p]; d.aln_off = aln_total; d.read_off = others_index[p];
      d.L = (int32_t)L; d.n_others = (int32_t)no; d.skip = skip[p]; d.cls = cls;
      descs[(size_t)p] = d;
      int lid = cls;
      if (cls == 2) { const int64_t work = no * L; lid = work >= 32768 ? 2 : work >= 16384 ? 3 : work >= 8192 ? 4 : 5; }
      else if (cls == 4) lid = 7;
      lists[lid].push_back((int32_t)p); hc.n_cls[lid]++;
      cls_out[p] = cls;
      aln_total += no * L; alt_total += L;
    }
    View v{};
    v.cfg.consensus_kmer_len = klen;
    v.wave_path = 1;
    std::vector<uint8_t> pool((size_t)seq_pool_len + 32, 0);
    if (seq_pool_len) memcpy(pool.data(), seq_pool, (size_t)seq_pool_len);
    v.pool = pool.data(); v.pool_len = seq_pool_len; v.pool_cap = seq_pool_len + 32;
    v.cdesc = descs.data();
    for (int k = 0; k < 6; k++) { lists[k].push_back(0); v.cls_list[k] = lists[k].data(); }
    const unsigned long long n7 = hc.n_cls[7];
    lists[7].resize((size_t)np + 1, 0);
    v.cls_list[7] = lists[7].data();
    v.cnt = &hc;
    std::vector<int64_t> crl_off(others_off, others_off + n_reads); crl_off.push_back(0);
    std::vector<int32_t> crl_len(others_len, others_len + n_reads); crl_len.push_back(0);
    v.crl_off = crl_off.data(); v.crl_len = crl_len.data();
    std::vector<uint8_t> aln((size_t)aln_total + 16, 0xee), kept((size_t)n_reads + 16, 0), alt((size_t)alt_total + 16, 0xee);
    v.aln = aln.data(); v.aln_kept_w = kept.data(); v.alt_pool = alt.data();
    std::vector<unsigned long long> stripes(4 * 64 * 16, 0);
    v.stripes = stripes.data();
    const int64_t n_small = (int64_t)hc.n_cls[1], n_large = (int64_t)(hc.n_cls[2] + hc.n_cls[3] + hc.n_cls[4] + hc.n_cls[5]);
    auto grid = [&](int64_t n) { return (unsigned)(grid_cap > 0 && n > grid_cap ? grid_cap : n); };
    if (hc.n_cls[0]) hipLaunchKernelGGL(e4c_copy, dim3(grid((int64_t)hc.n_cls[0])), dim3(64), 0, nullptr, v, (int64_t)0);
    if (n_large > 0) {
      if (nw == 8) hipLaunchKernelGGL((K_CONS_LARGE_8W), dim3(grid(n_large)), dim3(512), 0, nullptr, v, (int64_t)0);
      else hipLaunchKernelGGL((K_CONS_LARGE), dim3(grid(n_large)), dim3(256), 0, nullptr, v, (int64_t)0);
    }
    if (n_small > 0) {
      if (nw == 1) hipLaunchKernelGGL((K_CONS_SMALL_1W), dim3(grid(n_small)), dim3(64), 0, nullptr, v, (int64_t)0);
      else hipLaunchKernelGGL((K_CONS_SMALL), dim3(grid(n_small)), dim3(256), 0, nullptr, v, (int64_t)0);
    }
    *handed_over = (int64_t)(hc.n_cls[7] - n7);
    if (hc.n_cls[7]) hipLaunchKernelGGL((K_CONS_ROWS), dim3(grid((int64_t)hc.n_cls[7])), dim3(256), 0, nullptr, v, (int64_t)0);
    if (alt_total) memcpy(out_pool, alt.data(), (size_t)alt_total);
    return 0;
  } catch (const Error& e) {
    snprintf(g_err, sizeof g_err, "%s", e.msg.c_str());
    return 1;
  }
}
