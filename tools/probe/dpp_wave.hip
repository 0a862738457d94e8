#include <hip/hip_runtime.h>
__global__ void k(int* out) {
  int x = threadIdx.x * 3 + 1;
  int r = __builtin_amdgcn_update_dpp(0, x, 0x13C, 0xf, 0xf, false);   // wave_ror:1
  int s = __builtin_amdgcn_update_dpp(-7, x, 0x138, 0xf, 0xf, false);  // wave_shr:1
  out[threadIdx.x] = r; out[64 + threadIdx.x] = s;
}
int main() { int* d; hipMalloc(&d, 512); k<<<1,64>>>(d); int h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 64; i++) { int e = ((i + 63) & 63) * 3 + 1; if (h[i] != e) bad++; int es = i ? (i - 1) * 3 + 1 : -7; if (h[64 + i] != es) bad++; }
  printf("wave_ror/wave_shr mismatches: %d (lane0 ror=%d shr=%d, lane1 ror=%d)\n", bad, h[0], h[64], h[1]); return bad; }
