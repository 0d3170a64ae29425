// fp64 MFMA throughput per wave when 1..4 waves of a CU (one per SIMD) issue it concurrently.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(double* D, long long* cyc, int active_waves, double a, double b) {
  const int wave = threadIdx.x >> 6;
  d4 y0 = {a, b, a, b}, y1 = y0, y2 = y0;
  __syncthreads();
  long long t0 = clock64();
  if (wave < active_waves) {
    for (int i = 0; i < 128; ++i) {
      y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, y0, 0, 0, 0);
      y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, y1, 0, 0, 0);
      y2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, y2, 0, 0, 0);
    }
  }
  long long t1 = clock64();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
  D[blockIdx.x * 256 + threadIdx.x] = y0[0] + y1[1] + y2[2];
}
int main() {
  double* D; long long* c; long long h[4];
  hipMalloc(&D, 256 * 256 * 8); hipMalloc(&c, 256 * 4 * 8);
  for (int blocks : {1, 256})
    for (int w = 1; w <= 4; ++w) {
      k<<<blocks, 256>>>(D, c, w, 1e-3, 1e-3); hipDeviceSynchronize();
      hipMemcpy(h, c, 32, hipMemcpyDeviceToHost);
      printf("blocks %3d, %d waves issuing: %.1f cycles per MFMA (wave 0)\n", blocks, w, h[0] / 384.0);
    }
  return 0;
}
