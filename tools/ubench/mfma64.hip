// Verify the lane layout of v_mfma_f64_16x16x4_f64 on gfx950 and time it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* B, double* D, long long* cyc) {
  // A: 16x4 row-major, B: 4x16 row-major, D: 16x16 row-major
  const int l = threadIdx.x;
  const double a = A[(l & 15) * 4 + (l >> 4)];
  const double b = B[(l >> 4) * 16 + (l & 15)];
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
  // timing: dependent chain and independent
  long long t0 = clock64();
  d4 x = c;
  for (int i = 0; i < 256; ++i) x = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, x, 0, 0, 0);
  long long t1 = clock64();
  d4 y0 = c, y1 = c, y2 = c, y3 = c;
  for (int i = 0; i < 64; ++i) {
    y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, y0, 0, 0, 0);
    y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, y1, 0, 0, 0);
    y2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, y2, 0, 0, 0);
    y3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, y3, 0, 0, 0);
  }
  long long t2 = clock64();
  if (l == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
  D[256 + l] = x[0] + y0[0] + y1[1] + y2[2] + y3[3];
}
int main() {
  double hA[64], hB[64], hD[256 + 64], ref[256];
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) hA[i * 4 + k] = 1.0 + i * 0.37 + k * 1.9;
  for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) hB[k * 16 + j] = -2.0 + k * 0.11 + j * j * 0.013;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += hA[i * 4 + k] * hB[k * 16 + j]; ref[i * 16 + j] = s; }
  double *dA, *dB, *dD; long long* dc, hc[2];
  hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, sizeof(hD)); hipMalloc(&dc, 16);
  hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dA, dB, dD, dc); hipDeviceSynchronize();
  hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost); hipMemcpy(hc, dc, 16, hipMemcpyDeviceToHost);
  double e = 0; for (int i = 0; i < 256; ++i) e = fmax(e, fabs(hD[i] - ref[i]));
  printf("mfma_f64_16x16x4 layout check: max abs err %.3e (%s)\n", e, e < 1e-12 ? "LAYOUT OK" : "LAYOUT MISMATCH");
  printf("dependent chain: %.1f cycles/mfma ; 4 independent accumulators: %.1f cycles/mfma\n", hc[0] / 256.0, hc[1] / 256.0);
  return 0;
}
