// Per-instruction issue cost for ONE wave per SIMD (gfx950): operand kinds and op types.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 128
#define BENCH(name, body)                                                      \
  __global__ void name(double* out, long long* cyc, double a, double b, double c0) { \
    double x = a + threadIdx.x, y = b + threadIdx.x, z = c0;                   \
    int k = threadIdx.x;                                                       \
    long long t0 = clock64();                                                  \
    for (int it = 0; it < 64; ++it) {                                          \
      _Pragma("unroll") for (int r = 0; r < REP; ++r) { body; }                \
    }                                                                          \
    long long t1 = clock64();                                                  \
    out[threadIdx.x] = x + y + z + k;                                          \
    if (threadIdx.x == 0) cyc[0] = t1 - t0;                                    \
  }
BENCH(fma_vvv, x = fma(x, y, z))
BENCH(fma_svv, x = fma(b, x, y))            // b is an SGPR pair (kernel arg)
BENCH(fma_vvs, x = fma(x, y, b))
BENCH(fma_lit, x = fma(x, y, 0.123456789))  // literal constant
BENCH(fma_2chain, x = fma(x, b, a); y = fma(y, a, b))
BENCH(mul_dep, x = x * y)
BENCH(add_dep, x = x + y)
BENCH(add_s, x = x + b)
BENCH(rndne, x = rint(x) + 0.5)
BENCH(cvt, k = (int)x; x = x + (double)(k & 1))
BENCH(rcp, x = __builtin_amdgcn_rcp(x))
BENCH(mov64, x = y; y = z; z = x + 1.0)
int main() {
  double* out; long long* cyc, h; hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
#define RUN(name, per) name<<<1, 64>>>(out, cyc, 1.0, 0.999, 0.5); hipDeviceSynchronize(); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-12s %.2f cycles/iter\n", #name, (double)h / (64.0 * REP));
  RUN(fma_vvv, 1) RUN(fma_svv, 1) RUN(fma_vvs, 1) RUN(fma_lit, 1) RUN(fma_2chain, 2) RUN(mul_dep, 1) RUN(add_dep, 1) RUN(add_s, 1)
  RUN(rndne, 2) RUN(cvt, 4) RUN(rcp, 1) RUN(mov64, 1)
  return 0;
}
