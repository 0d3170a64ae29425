#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double* x, double* e, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
  double v = x[i], t = 1.0 / v;
  double r0 = __builtin_amdgcn_rcp(v);
  double e1 = fma(-v, r0, 1.0), r1 = fma(r0, e1, r0);
  double e2 = fma(-v, r1, 1.0), r2 = fma(r1, e2, r1);
  double u = fabs(t) * 2.220446049250313e-16;
  e[3 * i] = fabs(r0 - t) / u; e[3 * i + 1] = fabs(r1 - t) / u; e[3 * i + 2] = fabs(r2 - t) / u;
}
int main() {
  const int n = 1 << 20; double* hx = new double[n]; double* he = new double[3 * n];
  srand(1); for (int i = 0; i < n; ++i) hx[i] = ldexp(1.0 + (double)rand() / RAND_MAX, (rand() % 40) - 20) * ((rand() & 1) ? 1 : -1);
  double *dx, *de; hipMalloc(&dx, n * 8); hipMalloc(&de, 3 * n * 8); hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, de, n); hipMemcpy(he, de, 3 * n * 8, hipMemcpyDeviceToHost);
  double m[3] = {0, 0, 0}; for (int i = 0; i < n; ++i) for (int j = 0; j < 3; ++j) m[j] = fmax(m[j], he[3 * i + j]);
  printf("max error in ulps: rcp %.3g, +1 Newton %.3g, +2 Newton %.3g\n", m[0], m[1], m[2]);
}
