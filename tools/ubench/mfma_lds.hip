// 27 fp64 MFMAs per "phase" fed from LDS exactly like the large-path backward pass (B operand:
// F[k][16c + lr], rows lk at stride FS): what does a phase cost with 1 / 3 waves of the CU active,
// with and without the LDS operand loads?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int FS = 48, KS = 9, CT = 3;
template <int MODE>   // 0: operands constant, 1: B from LDS each phase, 2: A and B from LDS
__global__ void k(double* D, long long* cyc, int active_waves, int iters) {
  extern __shared__ double lds[];
  for (int i = threadIdx.x; i < 36 * FS * 2; i += blockDim.x) lds[i] = 1e-3 * (i % 97);
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
  const double* fb = lds + lk * FS + lr;
  const double* fa = lds + 36 * FS + lr * 37 + lk;
  double av[KS], bv[CT][KS];
  for (int ks = 0; ks < KS; ++ks) { av[ks] = 1e-3 * ks + lane; for (int c = 0; c < CT; ++c) bv[c][ks] = 2e-3 * ks + c; }
  d4 acc[CT];
  double sum = 0.0;
  long long t0 = clock64();
  if (wave < active_waves) {
    for (int it = 0; it < iters; ++it) {
      if (MODE >= 1) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int c = 0; c < CT; ++c) bv[c][ks] = fb[ks * 4 * FS + 16 * c];
      }
      if (MODE >= 2) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) av[ks] = fa[4 * ks];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < CT; ++c) acc[c] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks], bv[c][ks], acc[c], 0, 0, 0);
      sum += acc[0][0] + acc[1][1] + acc[2][2];
      if (MODE == 0) { av[0] += 1e-9 * sum; }
    }
  }
  long long t1 = clock64();
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
  D[blockIdx.x * 256 + threadIdx.x] = sum;
}
int main() {
  double* D; long long* c; long long h[4];
  hipMalloc(&D, 256 * 256 * 8); hipMalloc(&c, 256 * 4 * 8);
  const int iters = 200;
  for (int w : {1, 3}) {
#define RUN(M, name) k<M><<<64, 256, 36 * FS * 2 * 8 + 4096>>>(D, c, w, iters); hipDeviceSynchronize(); hipMemcpy(h, c, 32, hipMemcpyDeviceToHost); printf("%d wave(s) active, %-28s %.0f cycles per 27-MFMA phase (%.1f per MFMA)\n", w, name, (double)h[0] / iters, (double)h[0] / iters / 27);
    RUN(0, "operands in registers:") RUN(1, "B operands from LDS:") RUN(2, "A and B operands from LDS:")
  }
  return 0;
}
