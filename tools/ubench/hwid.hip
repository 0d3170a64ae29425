// Where do the two waves of a 128-thread workgroup land?  (SIMD / wave slot / CU from HW_REG_HW_ID)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void k(unsigned* out, unsigned* xcc) {
  extern __shared__ double lds[];
  lds[threadIdx.x] = 1.0;
  const unsigned id = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));      // HW_REG_HW_ID, all 32 bits
  const unsigned x = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));       // HW_REG_XCC_ID (gfx940+), low 4 bits
  if ((threadIdx.x & 63) == 0) { out[blockIdx.x * 2 + (threadIdx.x >> 6)] = id; xcc[blockIdx.x * 2 + (threadIdx.x >> 6)] = x; }
  // keep the workgroup resident long enough for all 1024 to be co-resident
  long long t0 = clock64();
  while (clock64() - t0 < 2000000) {}
}
int main() {
  const int B = 1024;
  unsigned *d, *dx; hipMalloc(&d, B * 2 * 4); hipMalloc(&dx, B * 2 * 4);
  k<<<B, 128, 36 * 1024>>>(d, dx); hipDeviceSynchronize();
  std::vector<unsigned> h(B * 2), hx(B * 2);
  hipMemcpy(h.data(), d, B * 2 * 4, hipMemcpyDeviceToHost); hipMemcpy(hx.data(), dx, B * 2 * 4, hipMemcpyDeviceToHost);
  int pair_hist[4][4] = {};
  std::map<unsigned, std::vector<int>> mains_per_cu;   // key = xcc|se|sh|cu -> simd ids of wave 0
  int same_slot = 0;
  for (int b = 0; b < B; ++b) {
    unsigned a = h[2 * b], c = h[2 * b + 1];
    int s0 = (a >> 4) & 3, s1 = (c >> 4) & 3, w0 = a & 15, w1 = c & 15;
    pair_hist[s0][s1]++;
    same_slot += (w0 == w1);
    unsigned key = (hx[2 * b] << 16) | ((a >> 8) & 0xff);   // cu_id[11:8], sh[12], se[15:13]
    mains_per_cu[key].push_back(s0);
  }
  printf("(simd of wave0, simd of wave1) histogram:\n");
  for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) printf("%5d", pair_hist[i][j]); printf("\n"); }
  printf("workgroups whose two waves have the same wave-slot id: %d of %d; distinct CUs seen: %zu\n", same_slot, B, mains_per_cu.size());
  int hist[5] = {};
  for (auto& kv : mains_per_cu) { int cnt[4] = {}; for (int s : kv.second) cnt[s]++; int mx = 0; for (int s = 0; s < 4; ++s) mx = cnt[s] > mx ? cnt[s] : mx; hist[mx > 4 ? 4 : mx]++; }
  printf("CUs by max number of wave-0s on one SIMD: 1:%d 2:%d 3:%d 4+:%d\n", hist[1], hist[2], hist[3], hist[4]);
  for (int b = 0; b < 8; ++b) printf("wg %d: w0 simd %u slot %u cu %u se %u xcc %u | w1 simd %u slot %u\n", b, (h[2*b]>>4)&3, h[2*b]&15, (h[2*b]>>8)&15, (h[2*b]>>13)&7, hx[2*b]&15, (h[2*b+1]>>4)&3, h[2*b+1]&15);
  return 0;
}
