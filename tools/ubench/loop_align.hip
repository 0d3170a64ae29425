// What does a loop's back edge cost one wave, and does it depend on where the loop starts?  A loop of NB v_fma_f64 (8 independent
// chains) + s_sub / s_cmp / s_cbranch, its first instruction placed PAD dwords past a 256-byte boundary; cycles per trip from
// s_memtime, minus NB x 4.04 (the straight-line rate, tools/ubench/issue_interval.hip) = the cost of the back edge.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/loop_align.hip -o tools/ubench/loop_align && tools/ubench/loop_align
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NB, int PAD>
__global__ void __launch_bounds__(64) k(double* sink, long long* clk, int iters, double y) {
  double x0 = threadIdx.x * 1e-3, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  long long c0, c1;
  asm volatile(
      "s_memtime %[c0]\n"
      "s_waitcnt lgkmcnt(0)\n"
      "s_branch 1f\n"
      ".p2align 8\n"
      ".rept %c[pad]\n s_nop 0\n .endr\n"
      "1:\n"
      ".rept %c[nb8]\n"
      "v_fma_f64 %[x0], %[x0], %[y], %[x0]\n v_fma_f64 %[x1], %[x1], %[y], %[x1]\n v_fma_f64 %[x2], %[x2], %[y], %[x2]\n v_fma_f64 %[x3], %[x3], %[y], %[x3]\n"
      "v_fma_f64 %[x4], %[x4], %[y], %[x4]\n v_fma_f64 %[x5], %[x5], %[y], %[x5]\n v_fma_f64 %[x6], %[x6], %[y], %[x6]\n v_fma_f64 %[x7], %[x7], %[y], %[x7]\n"
      ".endr\n"
      "s_sub_u32 %[it], %[it], 1\n"
      "s_cmp_lg_u32 %[it], 0\n"
      "s_cbranch_scc1 1b\n"
      "s_memtime %[c1]\n"
      "s_waitcnt lgkmcnt(0)\n"
      : [c0] "=&s"(c0), [c1] "=&s"(c1), [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2), [x3] "+v"(x3), [x4] "+v"(x4), [x5] "+v"(x5), [x6] "+v"(x6),
        [x7] "+v"(x7), [it] "+s"(iters)
      : [y] "v"(y), [pad] "i"(PAD), [nb8] "i"(NB / 8)
      : "scc", "memory");
  const double s = ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7));
  if (s == 12345.678) sink[0] = s;
  if (threadIdx.x == 0) clk[0] = c1 - c0;
}

template <int NB, int PAD>
double run(double* sink, long long* clk) {
  const int iters = 20000;
  long long h = 0;
  for (int rep = 0; rep < 2; ++rep) { k<NB, PAD><<<1, 64>>>(sink, clk, iters, 1e-9); hipDeviceSynchronize(); }
  hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
  return double(h) / iters;
}

template <int NB>
void sweep(double* sink, long long* clk) {
  printf("loop of %3d v_fma_f64 + 3 scalar (%4d bytes): cycles per trip (back edge = that - %.0f) at PAD =", NB, NB * 8 + 12, NB * 4.04);
  const double r[] = {run<NB, 0>(sink, clk), run<NB, 1>(sink, clk), run<NB, 2>(sink, clk), run<NB, 3>(sink, clk), run<NB, 5>(sink, clk), run<NB, 7>(sink, clk),
                      run<NB, 8>(sink, clk), run<NB, 11>(sink, clk), run<NB, 13>(sink, clk), run<NB, 15>(sink, clk), run<NB, 16>(sink, clk), run<NB, 24>(sink, clk),
                      run<NB, 31>(sink, clk), run<NB, 32>(sink, clk), run<NB, 47>(sink, clk), run<NB, 63>(sink, clk)};
  const int pads[] = {0, 1, 2, 3, 5, 7, 8, 11, 13, 15, 16, 24, 31, 32, 47, 63};
  printf("\n   ");
  for (int i = 0; i < 16; ++i) printf(" %d:%.0f(%+.0f)", pads[i], r[i], r[i] - NB * 4.04);
  printf("\n");
}

int main() {
  double* sink; long long* clk;
  hipMalloc(&sink, 8); hipMalloc(&clk, 8);
  sweep<8>(sink, clk); sweep<16>(sink, clk); sweep<32>(sink, clk); sweep<64>(sink, clk); sweep<128>(sink, clk); sweep<256>(sink, clk);
  return 0;
}
