// TEST-ONLY: known answers for the execution model of hip/hip_runtime.h (tests/test_simt_tier.py).  The kernels below are
// written the way the product's kernels use each operation; what the hardware does with them is pinned by the GPU tier.
#include <hip/hip_runtime.h>

#include <vector>

namespace {

__global__ void k_scans(int* out_sum, int* out_max, int* out_prev) {
  const int lane = threadIdx.x & 63;
  unsigned x = (unsigned)((lane * 37 + 11) % 23);
  // the six-step DPP scan of snf_wave_cons.h / snf_extract.hip
  x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);
  x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);
  x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);
  x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);
  x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
  x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
  out_sum[lane] = (int)x;
  int m = (lane * 29 + 5) % 41;
  const int ctrl[6] = {0x111, 0x112, 0x114, 0x118, 0x142, 0x143}, rows[6] = {0xf, 0xf, 0xf, 0xf, 0xa, 0xc};
  for (int k = 0; k < 6; k++) { const int y = __builtin_amdgcn_update_dpp(m, m, ctrl[k], rows[k], 0xf, false); m = y > m ? y : m; }
  out_max[lane] = m;
  // the value of the lane before: row_shr:1 leaves the first lane of every row of 16 with `old`
  out_prev[lane] = __builtin_amdgcn_update_dpp(-7, lane * 3, 0x111, 0xf, 0xf, false);
}

__global__ void k_shuffles(long long* out) {
  const int lane = threadIdx.x & 63;
  long long bad = 0;
  const long long v = 1000 + lane;
  bad += __shfl(v, 17, 64) != 1017;
  bad += __shfl(v, (lane * 5) & 63, 64) != 1000 + ((lane * 5) & 63);
  bad += __shfl_up(v, 3, 64) != (lane >= 3 ? v - 3 : v);
  bad += __shfl_down(v, 5, 64) != (lane + 5 < 64 ? v + 5 : v);
  bad += __shfl_xor(v, 32, 64) != 1000 + (lane ^ 32);
  bad += __builtin_amdgcn_readlane(lane * 2, 40) != 80;
  bad += __builtin_amdgcn_readfirstlane(lane + 9) != 9;
  const double d = 0.5 * lane;
  bad += __shfl_xor(d, 1, 64) != 0.5 * (lane ^ 1);
  unsigned long long want = 0;
  for (int k = 0; k < 64; k++) if (k % 5 == 0) want |= 1ull << k;
  bad += __ballot(lane % 5 == 0) != want;
  // the in-group DPP steps of snf_wave_call_g.h: quad_perm:[1,0,3,2], quad_perm:[2,3,0,1], row_half_mirror, and row_mirror
  bad += __builtin_amdgcn_update_dpp(0, lane, 0xB1, 0xf, 0xf, false) != (lane ^ 1);
  bad += __builtin_amdgcn_update_dpp(0, lane, 0x4E, 0xf, 0xf, false) != (lane ^ 2);
  bad += __builtin_amdgcn_update_dpp(0, lane, 0x141, 0xf, 0xf, false) != ((lane & ~7) | (7 - (lane & 7)));
  bad += __builtin_amdgcn_update_dpp(0, lane, 0x140, 0xf, 0xf, false) != ((lane & ~15) | (15 - (lane & 15)));
  { int g = (lane * 13 + 3) % 17, want_g = 0;           // all-reduce over groups of 8 lanes
    for (int k = (lane & ~7); k < (lane & ~7) + 8; k++) want_g += (k * 13 + 3) % 17;
    g += __builtin_amdgcn_update_dpp(0, g, 0xB1, 0xf, 0xf, false);
    g += __builtin_amdgcn_update_dpp(0, g, 0x4E, 0xf, 0xf, false);
    g += __builtin_amdgcn_update_dpp(0, g, 0x141, 0xf, 0xf, false);
    bad += g != want_g; }
  atomicAdd(out, bad);
}

__global__ void k_divergent(long long* out) {
  const int lane = threadIdx.x & 63;
  long long bad = 0;
  if (lane % 3 == 0) {                       // only these lanes take part
    unsigned long long want = 0;
    for (int k = 0; k < 64; k++) if (k % 3 == 0 && (k & 1)) want |= 1ull << k;
    bad += __ballot(lane & 1) != want;
    bad += __builtin_amdgcn_readfirstlane(lane) != 0;
  }
  bad += __ballot(1) != ~0ull;               // behind the join: everybody
  if (lane >= 10) bad += __builtin_amdgcn_readfirstlane(lane) != 10;
  // the wave-aggregated append of snf_stage_final.h class_list_slot: lanes of one class share one counter update
  __shared__ unsigned counter[4];
  __shared__ int slot_of[64];
  if (lane < 4) counter[lane] = 0;
  __syncthreads();
  const int cls = lane % 3;                  // three classes in use
  for (int c = 0; c < 4; c++) {
    const unsigned long long m = __ballot(cls == c);
    if (cls == c) {
      const int leader = __builtin_ctzll(m);
      unsigned base = 0;
      if (lane == leader) base = atomicAdd(&counter[c], (unsigned)__builtin_popcountll(m));
      base = __shfl(base, leader, 64);
      slot_of[lane] = (int)base + __builtin_popcountll(m & ((1ull << lane) - 1ull));
    }
  }
  __syncthreads();
  bad += slot_of[lane] != lane / 3;
  if (lane == 0) bad += !(counter[0] == 22 && counter[1] == 21 && counter[2] == 21 && counter[3] == 0);
  atomicAdd(out, bad);
}

__global__ void k_block(long long* out) {
  __shared__ int part[4];
  __shared__ int total;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (wid == 3) return;                      // a wave that has left no longer counts at the barrier
  int x = tid;
  for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
  if (lane == 0) part[wid] = x;
  __syncthreads();
  if (tid == 0) total = part[0] + part[1] + part[2];
  __syncthreads();
  if (total != 191 * 192 / 2) atomicAdd(out, 1ll);
}

// 64 lanes, one serial body: on the GPU the read-modify-write happens once (the x_big kernels of snf_wave_call.h)
__global__ void x_big_selftest(long long* word, int* out_lds) {
  __shared__ int acc;
  __shared__ int row[64];
  const int lane = threadIdx.x;
  if (blockIdx.x == 0) {
    acc = 5;
    __syncthreads();
    *word += 1;                                // uniform: every lane, same value
    acc += 1;
    row[lane] = lane * lane;                   // cooperative: every lane its own element
    __syncthreads();
    int s = 0;
    for (int k = 0; k < 64; k++) s += row[k];
    if (lane == 0) { out_lds[0] = acc; out_lds[1] = s; }
  }
}
__global__ void plain_selftest(long long* word, int* out_lds) {
  const int lane = threadIdx.x;
  (void)lane; (void)out_lds;
  *word += 1;                                  // fibres: one after the other
  __syncthreads();
}

}  // namespace

// out[0..7]: mismatches of the checks; out[8] / out[9]: a word incremented by all 64 lanes in lock step / as plain fibres
extern "C" int simt_selftest(long long* out) {
  for (int k = 0; k < 16; k++) out[k] = 0;
  int *d_sum, *d_max, *d_prev, *d_lds;
  long long* d_bad;
  if (hipMalloc(&d_sum, 64 * 4) || hipMalloc(&d_max, 64 * 4) || hipMalloc(&d_prev, 64 * 4) || hipMalloc(&d_bad, 8 * 8) || hipMalloc(&d_lds, 8)) return 1;
  hipMemset(d_bad, 0, 64);
  hipLaunchKernelGGL(k_scans, dim3(1), dim3(64), 0, nullptr, d_sum, d_max, d_prev);
  {
    int acc = 0, mx = -1;
    for (int l = 0; l < 64; l++) {
      acc += (l * 37 + 11) % 23; const int m = (l * 29 + 5) % 41; mx = m > mx ? m : mx;
      out[0] += d_sum[l] != acc;
      out[1] += d_max[l] != mx;
      out[2] += d_prev[l] != ((l & 15) ? (l - 1) * 3 : -7);
    }
  }
  hipLaunchKernelGGL(k_shuffles, dim3(1), dim3(64), 0, nullptr, d_bad + 0);
  hipLaunchKernelGGL(k_divergent, dim3(1), dim3(64), 0, nullptr, d_bad + 1);
  hipLaunchKernelGGL(k_block, dim3(3), dim3(256), 0, nullptr, d_bad + 2);
  out[3] = d_bad[0]; out[4] = d_bad[1]; out[5] = d_bad[2];
  d_bad[3] = 0; d_bad[4] = 0; d_lds[0] = d_lds[1] = 0;
  hipLaunchKernelGGL(x_big_selftest, dim3(2), dim3(64), 0, nullptr, d_bad + 3, d_lds);
  hipLaunchKernelGGL(plain_selftest, dim3(1), dim3(64), 0, nullptr, d_bad + 4, d_lds);
  out[8] = d_bad[3]; out[9] = d_bad[4];
  int want_s = 0;
  for (int k = 0; k < 64; k++) want_s += k * k;
  out[6] = d_lds[0] != 6;
  out[7] = d_lds[1] != want_s;
  hipFree(d_sum); hipFree(d_max); hipFree(d_prev); hipFree(d_bad); hipFree(d_lds);
  return 0;
}
