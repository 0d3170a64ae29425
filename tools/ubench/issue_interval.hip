// How often can ONE wave issue an fp64 instruction?  Cycles per v_fma_f64 (s_memtime around the loop, the wave's own clock)
// for 1, 2, 4, 8 independent chains per lane, unrolled, with W waves per SIMD on (a) one CU alone and (b) every CU of the chip.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/issue_interval.hip -o tools/ubench/issue_interval && tools/ubench/issue_interval
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

constexpr int kTrip = 840;     // instructions per loop trip: a multiple of every chain count below
template <int CH>
__global__ void __launch_bounds__(256) fma_chains(double* sink, long long* clk, int iters, double y) {
  double x[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) x[c] = threadIdx.x * 1e-3 + c;
  const long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int rep = 0; rep < kTrip / CH; ++rep)
#pragma unroll
      for (int c = 0; c < CH; ++c) x[c] = __builtin_fma(x[c], y, 1e-9);
  }
  const long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) s += x[c];
  if (s == 12345.678) sink[0] = s;
  if ((threadIdx.x & 63) == 0) { const int w = blockIdx.x * 4 + (threadIdx.x >> 6); clk[2 * w] = c1 - c0; clk[2 * w + 1] = r1 - r0; }
}

template <int CH>
void run(const char* where, int blocks, int wg_per_cu, double* sink, long long* clk) {
  const int iters = 800;                       // x kTrip instructions per trip
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    fma_chains<CH><<<blocks, 256>>>(sink, clk, iters, 1.0000001);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  }
  std::vector<long long> h(2 * blocks * 4);
  hipMemcpy(h.data(), clk, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
  double cyc = 0, mhz = 0;
  for (int w = 0; w < blocks * 4; ++w) { cyc += double(h[2 * w]); mhz += double(h[2 * w]) / double(h[2 * w + 1]) * 100.0; }
  cyc /= blocks * 4; mhz /= blocks * 4;
  printf("%-9s %d wave(s)/SIMD  %d chain(s): %6.2f cycles per v_fma_f64 per wave (counter), clock %.0f MHz, kernel %.3f ms -> %.2f cycles by wall time\n",
         where, wg_per_cu, CH, cyc / (double(iters) * kTrip), mhz, ms, ms * 1e-3 * mhz * 1e6 / (double(iters) * kTrip));
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  double* sink; long long* clk;
  hipMalloc(&sink, 8); hipMalloc(&clk, 2 * 8 * cus * 8 * 4 * sizeof(long long));
  for (int wg : {1, 2, 4}) {
    run<1>("one CU", wg, wg, sink, clk); run<2>("one CU", wg, wg, sink, clk); run<3>("one CU", wg, wg, sink, clk); run<4>("one CU", wg, wg, sink, clk);
    run<5>("one CU", wg, wg, sink, clk); run<6>("one CU", wg, wg, sink, clk); run<7>("one CU", wg, wg, sink, clk); run<8>("one CU", wg, wg, sink, clk);
  }
  for (int wg : {1, 2, 4, 8}) {
    run<1>("all CUs", cus * wg, wg, sink, clk); run<2>("all CUs", cus * wg, wg, sink, clk); run<4>("all CUs", cus * wg, wg, sink, clk); run<8>("all CUs", cus * wg, wg, sink, clk);
  }
  return 0;
}
