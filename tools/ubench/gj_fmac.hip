// Gauss-Jordan elimination (ilqr_large.hpp: GjOuter) with the column update as ONE v_fmac_f64_dpp against the compiler's
// v_mov_b64_dpp + v_fma_f64 (-DMI_GJ_MOV_FMA): cycles per inverse (one wave, rows per lane, four matrices per wave) and a
// bitwise checksum of the inverses of 256 seeded SPD matrices - the two builds must print the same checksum.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Idrake_ddp_amd/csrc tools/ubench/gj_fmac.hip -o /tmp/gj_a
//   hipcc ... -DMI_GJ_MOV_FMA ... -o /tmp/gj_b
#include "ilqr_large.hpp"
#include <cstdio>
#include <cstring>
using namespace mi;
template <int m>
__global__ void gj_kernel(const double* A, double* W, long long* cyc, int reps) {
  const int lane = threadIdx.x, lr = lane & 15, blk = blockIdx.x * 4 + (lane >> 4);
  const int si = lr < m ? lr : m - 1;
  double base[m];
  for (int j = 0; j < m; ++j) base[j] = A[(size_t)blk * m * m + si * m + j];
  double arow[m], sc = 1.0, acc = 0.0;
  long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int j = 0; j < m; ++j) arow[j] = base[j] + acc * 1e-300;
    sc = 1.0;
    GjOuter<m, 0>::run(arow, sc, si);
    acc = sc * arow[0];
  }
  long long t1 = clock64();
  if (lr < m)
    for (int j = 0; j < m; ++j) W[(size_t)blk * m * m + lr * m + j] = sc * arow[j];
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int m>
void run(const char* name) {
  const int NB = 64, NM = NB * 4, reps = 200;
  static double h[256 * 16 * 16], w[256 * 16 * 16];
  unsigned long long seed = 12345 + m;
  auto rnd = [&]() { seed = seed * 6364136223846793005ULL + 1442695040888963407ULL; return (double)(seed >> 11) / 9007199254740992.0 - 0.5; };
  for (int b = 0; b < NM; ++b) {
    double L[16][16];
    for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) L[i][j] = rnd();
    for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) { double s = i == j ? 0.05 : 0.0; for (int k = 0; k < m; ++k) s += L[i][k] * L[j][k]; h[b * m * m + i * m + j] = s; }
  }
  double *dA, *dW; long long* dc; long long hc[64];
  hipMalloc(&dA, sizeof(h)); hipMalloc(&dW, sizeof(w)); hipMalloc(&dc, sizeof(hc));
  hipMemcpy(dA, h, sizeof(h), hipMemcpyHostToDevice);
  gj_kernel<m><<<NB, 64>>>(dA, dW, dc, reps); hipDeviceSynchronize();
  gj_kernel<m><<<NB, 64>>>(dA, dW, dc, reps); hipDeviceSynchronize();
  hipMemcpy(w, dW, sizeof(w), hipMemcpyDeviceToHost); hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost);
  unsigned long long sum = 0; double worst = 0.0;
  for (int b = 0; b < NM; ++b) {
    for (int i = 0; i < m * m; ++i) { unsigned long long u; memcpy(&u, &w[b * m * m + i], 8); sum = sum * 1099511628211ULL + u; }
    for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) {           // | A W - I |
      double s = i == j ? -1.0 : 0.0; for (int k = 0; k < m; ++k) s += h[b * m * m + i * m + k] * w[b * m * m + k * m + j];
      if (fabs(s) > worst) worst = fabs(s);
    }
  }
  long long best = hc[0]; for (int i = 1; i < NB; ++i) if (hc[i] < best) best = hc[i];
  printf("%s m=%2d: %7.1f cycles per inverse (%5.1f per pivot), checksum %016llx, max|A W - I| %.2e\n", name, m, best / (double)reps, best / (double)reps / m, sum, worst);
  hipFree(dA); hipFree(dW); hipFree(dc);
}
int main() {
#ifdef MI_GJ_MOV_FMA
  const char* name = "mov_dpp+fma ";
#else
  const char* name = "fmac_dpp    ";
#endif
  run<1>(name); run<2>(name); run<3>(name); run<4>(name); run<7>(name); run<12>(name); run<16>(name);
  return 0;
}
