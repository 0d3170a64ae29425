// What one DEPENDENT rollout step of the cart-pole + wall model costs on one wave (the C4 line search: 199 of them in a row per
// candidate lane), by form of the math: Horner / Estrin polynomials (-DMI_POLY_ESTRIN=0|1), with and without the wave-uniform
// short cut of the contact force (softplus(z) = exp(z) bitwise once exp(z) < 2^-53: -DMI_SOFTPLUS_SKIP=0|1, dual.hpp).
//   for e in 0 1; do for k in 0 1; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -ffp-contract=fast -DMI_POLY_ESTRIN=$e -DMI_SOFTPLUS_SKIP=$k \
//     -I drake_ddp_amd/csrc tools/ubench/chain_step.hip -o tools/ubench/chain_step_$e$k; done; done
#include <hip/hip_runtime.h>
#include <cstdio>
#include "models.hpp"
using namespace mi;
template <class M, class = void> struct HasPool : std::false_type {};
template <class M> struct HasPool<M, std::void_t<decltype(M::kHasStepPool)>> : std::bool_constant<M::kHasStepPool> {};
struct NoPool_ {};
template <class M, bool = HasPool<M>::value> struct PoolOf_ { using type = NoPool_; __device__ static NoPool_ make() { return {}; } };
template <class M> struct PoolOf_<M, true> { using type = typename M::StepPool; __device__ static type make() { return M::StepPool::in_vgprs(); } };
template <class M, class P>
__device__ __forceinline__ void step_of(const double* x, const double* u, double* xn, const double* p, double dt, const P& pool) {
  if constexpr (HasPool<M>::value) M::step_pooled(x, u, xn, p, dt, pool);     // (the polynomial constants in VGPRs for the loop, as in rollout())
  else M::template step<double>(x, u, xn, p, dt);
}
template <class M, int NP>
__global__ void __launch_bounds__(64) k(double* out, long long* cyc, const double* prm, double dt, int steps, double x_wall) {
  extern __shared__ double lds[];
  double* G = lds;                                   // records: xb[4] K[4] ub kap  (10 doubles, padded to 12)
  for (int i = threadIdx.x; i < 12 * 202; i += 64) G[i] = 1e-3 * (i % 7);
  __syncthreads();
  double p[NP];
  for (int i = 0; i < NP; ++i) p[i] = prm[i];
  double x[4] = {x_wall + 1e-3 * threadIdx.x, 3.6 + 1e-3 * threadIdx.x, 0.0, 0.0};
  const typename PoolOf_<M>::type pool = PoolOf_<M>::make();
  const long long t0 = clock64();
  for (int rep = 0; rep < steps / 200; ++rep) {
    const double* g = G;
#pragma unroll 2
    for (int t = 0; t < 200; ++t) {
      double u[1], xn[4];
      u[0] = (g[8] - g[9]) - (g[4] * (x[0] - g[0]) + g[5] * (x[1] - g[1]) + g[6] * (x[2] - g[2]) + g[7] * (x[3] - g[3]));
      step_of<M>(x, u, xn, p, dt, pool);
      for (int i = 0; i < 4; ++i) x[i] = xn[i];
      x[0] = fmin(fmax(x[0], -2.0), 2.0); x[2] = fmin(fmax(x[2], -5.0), 5.0); x[3] = fmin(fmax(x[3], -5.0), 5.0);   // (keep the loop bounded; two cheap ops)
      g += 12;
    }
  }
  const long long t1 = clock64();
  out[threadIdx.x] = x[0] + x[1] + x[2] + x[3];
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  double *out, *prm; long long *cyc, h;
  hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8); hipMalloc(&prm, 16 * 8);
  const int steps = 4000;
  {
    const double p[8] = {1.0, 0.1, 0.5, 9.81, -0.5, 0.05, 200.0, 0.0032};   // mc, mp, l, g, wall face, ball radius, k, sigma
    hipMemcpy(prm, p, sizeof p, hipMemcpyHostToDevice);
    for (double xw : {0.0, -0.21}) {                  // far from the wall (z << -37 on every lane) / at the wall (contact on every lane)
      k<CartPoleWall, 8><<<1, 64, 12 * 202 * 8>>>(out, cyc, prm, 0.01, steps, xw); hipDeviceSynchronize();
      hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      printf("cart-pole + wall, cart at %5.2f: %.1f cycles per dependent step (MI_POLY_ESTRIN=%d MI_SOFTPLUS_SKIP=%d)\n", xw, (double)h / steps, MI_POLY_ESTRIN, MI_SOFTPLUS_SKIP);
    }
  }
  {
    const double p[10] = {1.0, 1.0, 1.0, 0.5, 1.0, 0.083, 0.33, 0.1, 0.1, 9.81};
    hipMemcpy(prm, p, sizeof p, hipMemcpyHostToDevice);
    k<Acrobot, 10><<<1, 64, 12 * 202 * 8>>>(out, cyc, prm, 0.004, steps, 0.0); hipDeviceSynchronize();
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("acrobot: %.1f cycles per dependent step (MI_POLY_ESTRIN=%d)\n", (double)h / steps, MI_POLY_ESTRIN);
  }
  return 0;
}
