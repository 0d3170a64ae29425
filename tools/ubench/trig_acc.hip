// Accuracy of the short-chain fp64 primitives (fastmath.hpp) against host libm, in ulps.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -I drake_ddp_amd/csrc tools/ubench/trig_acc.hip -o tools/ubench/trig_acc
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include <random>
#include "fastmath.hpp"

__global__ void k(const double* x, double* s, double* c, double* e, double* l, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  s[i] = mi::fast_sin(x[i]);
  c[i] = mi::fast_cos(x[i]);
  e[i] = mi::fast_exp_nonpos(-fabs(x[i]));
  l[i] = mi::fast_log1p01(fabs(x[i]) - floor(fabs(x[i])));
}

static double ulps(double a, double b) {
  if (a == b) return 0;
  double u = std::nextafter(std::fabs(b), INFINITY) - std::fabs(b);
  return std::fabs(a - b) / u;
}

int main() {
  const int n = 1 << 22;
  std::vector<double> x(n);
  std::mt19937_64 g(1);
  std::uniform_real_distribution<double> U(-1, 1);
  for (int i = 0; i < n; ++i) {
    double s = (i & 3) == 0 ? 4.0 : (i & 3) == 1 ? 100.0 : (i & 3) == 2 ? 1e4 : 1e6;
    x[i] = s * U(g);
  }
  // exact half-way and integer multiples of pi
  for (int i = 0; i < 4096; ++i) x[i] = (i - 2048) * 0.5 * M_PI;
  double *dx, *ds, *dc, *de, *dl;
  hipMalloc(&dx, n * 8); hipMalloc(&ds, n * 8); hipMalloc(&dc, n * 8); hipMalloc(&de, n * 8); hipMalloc(&dl, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, ds, dc, de, dl, n);
  std::vector<double> s(n), c(n), e(n), l(n);
  hipMemcpy(s.data(), ds, n * 8, hipMemcpyDeviceToHost); hipMemcpy(c.data(), dc, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(e.data(), de, n * 8, hipMemcpyDeviceToHost); hipMemcpy(l.data(), dl, n * 8, hipMemcpyDeviceToHost);
  double ms = 0, mc = 0, me = 0, ml = 0, as = 0, ac = 0;
  for (int i = 0; i < n; ++i) {
    const double es = std::sin(x[i]), ec = std::cos(x[i]);
    // near the zeros of sin/cos the reduction error (pi split to ~2^-121 * n) dominates: measure
    // those absolutely against 1 ulp of the argument instead of relatively
    if (std::fabs(es) > 1e-6) ms = std::fmax(ms, ulps(s[i], es)); else as = std::fmax(as, std::fabs(s[i] - es));
    if (std::fabs(ec) > 1e-6) mc = std::fmax(mc, ulps(c[i], ec)); else ac = std::fmax(ac, std::fabs(c[i] - ec));
    me = std::fmax(me, ulps(e[i], std::exp(-std::fabs(x[i]))) * (std::exp(-std::fabs(x[i])) > 1e-300));
    const double y = std::fabs(x[i]) - std::floor(std::fabs(x[i]));
    ml = std::fmax(ml, ulps(l[i], std::log1p(y)));
  }
  printf("max ulp: sin %.2f cos %.2f exp %.2f log1p %.2f ; abs err near zeros: sin %.3e cos %.3e\n", ms, mc, me, ml, as, ac);
  return (ms < 4 && mc < 4 && me < 4 && ml < 4 && as < 1e-15 && ac < 1e-15) ? 0 : 1;
}
