// Do the fp64 matrix pipe and the fp64 vector pipe of a SIMD run side by side?  One wave per SIMD, per loop trip:
//   (a) 4 v_mfma_f64_16x16x4 (independent accumulators)        (b) NV v_fma_f64 (8 independent chains)
//   (c) both, interleaved in ONE instruction stream              (d) two waves per SIMD, one doing (a), the other (b)
// If the pipes are separate, (c) ~ max(a, b); if the matrix instruction occupies the vector fp64 datapath, (c) ~ a + b.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_overlap.hip -o tools/ubench/mfma_valu_overlap && tools/ubench/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE, int NV>   // MODE 0: mfma, 1: valu, 2: both in one stream, 3: even waves mfma / odd waves valu (512 threads: 2 waves per SIMD)
__global__ void __launch_bounds__(512) k(double* sink, long long* clk, int iters, double a, double b, double y) {
  d4 y0 = {a, b, a, b}, y1 = y0, y2 = y0, y3 = y0;
  double x[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) x[c] = threadIdx.x * 1e-3 + c;
  const int wave = threadIdx.x >> 6;
  const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && (wave >> 2) == 0);   // waves 0..3 and 4..7 land on SIMDs 0..3 each
  const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && (wave >> 2) == 1);
  const long long c0 = __builtin_readcyclecounter();
  if (do_m && do_v) {
    for (int i = 0; i < iters; ++i) {
      y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, y0, 0, 0, 0);
#pragma unroll
      for (int c = 0; c < NV / 4; ++c) x[c % 8] = __builtin_fma(x[c % 8], y, 1e-9);
      y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, y1, 0, 0, 0);
#pragma unroll
      for (int c = NV / 4; c < NV / 2; ++c) x[c % 8] = __builtin_fma(x[c % 8], y, 1e-9);
      y2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, y2, 0, 0, 0);
#pragma unroll
      for (int c = NV / 2; c < 3 * NV / 4; ++c) x[c % 8] = __builtin_fma(x[c % 8], y, 1e-9);
      y3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, y3, 0, 0, 0);
#pragma unroll
      for (int c = 3 * NV / 4; c < NV; ++c) x[c % 8] = __builtin_fma(x[c % 8], y, 1e-9);
    }
  } else if (do_m) {
    for (int i = 0; i < iters; ++i) {
      y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, y0, 0, 0, 0);
      y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, y1, 0, 0, 0);
      y2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, y2, 0, 0, 0);
      y3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, y3, 0, 0, 0);
    }
  } else if (do_v) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int c = 0; c < NV; ++c) x[c % 8] = __builtin_fma(x[c % 8], y, 1e-9);
    }
  }
  const long long c1 = __builtin_readcyclecounter();
  double s = y0[0] + y1[1] + y2[2] + y3[3];
#pragma unroll
  for (int c = 0; c < 8; ++c) s += x[c];
  if (s == 12345.678) sink[0] = s;
  if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 8 + wave] = c1 - c0;
}

template <int MODE, int NV>
void run(const char* what, int blocks, double* sink, long long* clk) {
  const int iters = 4000, threads = MODE == 3 ? 512 : 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    k<MODE, NV><<<blocks, threads>>>(sink, clk, iters, 1e-3, 1e-3, 1.0000001);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  }
  std::vector<long long> h(blocks * 8);
  hipMemcpy(h.data(), clk, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
  double cm = 0, cv = 0; int nm = 0, nv = 0;
  for (int b = 0; b < blocks; ++b)
    for (int w = 0; w < threads / 64; ++w) { if (MODE == 3 && w >= 4) { cv += h[b * 8 + w]; ++nv; } else { cm += h[b * 8 + w]; ++nm; } }
  if (MODE == 3) printf("%-44s NV=%2d: matrix waves %7.1f cycles per trip, vector waves %7.1f   (kernel %.3f ms)\n", what, NV, cm / nm / iters, cv / nv / iters, ms);
  else printf("%-44s NV=%2d: %7.1f cycles per trip   (kernel %.3f ms)\n", what, NV, cm / nm / iters, ms);
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  double* sink; long long* clk;
  hipMalloc(&sink, 8); hipMalloc(&clk, cus * 8 * sizeof(long long));
  for (int blocks : {1, cus}) {
    printf("--- %d workgroup(s), one per CU; a trip = 4 MFMA 16x16x4 f64 and / or NV v_fma_f64\n", blocks);
    run<0, 16>("(a) matrix only", blocks, sink, clk);
    run<1, 16>("(b) vector only", blocks, sink, clk);
    run<1, 32>("(b) vector only", blocks, sink, clk);
    run<1, 64>("(b) vector only", blocks, sink, clk);
    run<2, 16>("(c) both, one instruction stream", blocks, sink, clk);
    run<2, 32>("(c) both, one instruction stream", blocks, sink, clk);
    run<2, 64>("(c) both, one instruction stream", blocks, sink, clk);
    run<3, 16>("(d) two waves per SIMD: one matrix, one vector", blocks, sink, clk);
    run<3, 64>("(d) two waves per SIMD: one matrix, one vector", blocks, sink, clk);
  }
  return 0;
}
