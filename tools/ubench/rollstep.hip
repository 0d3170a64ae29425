// Where do the cycles of one pendulum rollout step go?  Variants of the step loop, one wave.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -ffp-contract=fast -I drake_ddp_amd/csrc tools/ubench/rollstep.hip -o tools/ubench/rollstep
#include <hip/hip_runtime.h>
#include <cstdio>
#include "fastmath.hpp"
using namespace mi;
#define STEPS 4000
template <int V>
__global__ void __launch_bounds__(64) k(double* out, long long* cyc, double dt, double b, double mgl, double iml2, const double* __restrict__ Gg) {
  extern __shared__ double lds[];
  // G records: xb0 xb1 K0 K1 ub kap dv pad  (8 doubles); T records: x0 x1 u pad (4 doubles)
  double* G = lds; double* T = lds + 8 * 202;
  for (int i = threadIdx.x; i < 8 * 202; i += 64) G[i] = 1e-3 * i;
  __syncthreads();
  double x0 = 0.1, x1 = 0.2;
  double* tw = (threadIdx.x == 0 || V >= 5) ? T : (T + 4 * 202 + 2 * threadIdx.x);
  const int tstep = (threadIdx.x == 0 || V >= 5) ? 4 : 0;
  long long t0 = clock64();
  if (V != 5 || threadIdx.x == 0)
  for (int rep = 0; rep < STEPS / 200; ++rep) {
    const double* g = (V >= 8) ? (Gg + (size_t)blockIdx.x * 8 * 202) : G;
    double* w = tw;
#pragma unroll 2
    for (int t = 0; t < 200; ++t) {
      double xb0, xb1, K0, K1, ub, kap;
      if (V != 2) { xb0 = g[0]; xb1 = g[1]; K0 = g[2]; K1 = g[3]; ub = g[4]; kap = g[5]; }
      else { xb0 = dt; xb1 = b; K0 = mgl; K1 = iml2; ub = dt; kap = b; }
      double u = (ub - kap) - (K0 * (x0 - xb0) + K1 * (x1 - xb1));
      double s = (V == 3) ? x0 : fast_sin(x0);
      double acc = (u - b * x1 - mgl * s) * iml2;
      if (V == 4) { x1 = s; } else {
      x1 = fma(dt, acc, x1);
      x0 = fma(dt, x1, x0); }
      if (V == 9) {} else if (V == 6) { if (threadIdx.x == 0) { w[2] = u; w[4] = x0; w[5] = x1; } }
      else if (V != 1 && V != 4) { w[2] = u; w[4] = x0; w[5] = x1; }
      w += tstep; g += 8;
    }
  }
  long long t1 = clock64();
  out[blockIdx.x * 64 + threadIdx.x] = x0 + x1;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
typedef int i16v __attribute__((ext_vector_type(16)));
typedef double d8v __attribute__((ext_vector_type(8)));
#define SLOAD16(dst, ptr, off) asm volatile("s_load_dwordx16 %0, %1, " #off : "=s"(dst) : "s"(ptr))
#define SWAIT2(a, b) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b))
template <int ST>
__device__ __forceinline__ void step10(const i16v& rec, double& x0, double& x1, double*& w, double dt, double b, double mgl, double iml2) {
  const d8v r = __builtin_bit_cast(d8v, rec);
  double u = (r[4] - r[5]) - (r[2] * (x0 - r[0]) + r[3] * (x1 - r[1]));
  double s = fast_sin(x0);
  double acc = (u - b * x1 - mgl * s) * iml2;
  x1 = fma(dt, acc, x1);
  x0 = fma(dt, x1, x0);
  if (ST) { w[2] = u; w[4] = x0; w[5] = x1; }
  w += 4;
}
template <int ST>
__global__ void __launch_bounds__(64) k10(double* out, long long* cyc, double dt, double b, double mgl, double iml2, const double* Gg, double* Tg) {
  extern __shared__ double lds[];
  double* T = lds + 8 * 202;
  double x0 = 0.1, x1 = 0.2;
  long long t0 = clock64();
  for (int rep = 0; rep < STEPS / 200; ++rep) {
    const double* g = Gg + (size_t)blockIdx.x * 8 * 202;
    double* w = ST == 2 ? Tg + (size_t)blockIdx.x * 4 * 202 : T;
    i16v a0, a1, b0, b1;
    SLOAD16(a0, g, 0x0); SLOAD16(a1, g, 0x40);
    for (int t = 0; t < 200; t += 4) {
      SWAIT2(a0, a1);
      SLOAD16(b0, g, 0x80); SLOAD16(b1, g, 0xc0); __builtin_amdgcn_sched_barrier(0);
      step10<ST>(a0, x0, x1, w, dt, b, mgl, iml2);
      step10<ST>(a1, x0, x1, w, dt, b, mgl, iml2);
      SWAIT2(b0, b1);
      SLOAD16(a0, g, 0x100); SLOAD16(a1, g, 0x140); __builtin_amdgcn_sched_barrier(0);
      step10<ST>(b0, x0, x1, w, dt, b, mgl, iml2);
      step10<ST>(b1, x0, x1, w, dt, b, mgl, iml2);
      g += 32;
    }
    SWAIT2(a0, a1);
  }
  long long t1 = clock64();
  out[blockIdx.x * 64 + threadIdx.x] = x0 + x1;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  double* out; long long* cyc, h; hipMalloc(&out, 1024 * 64 * 8); hipMalloc(&cyc, 8); double* Tg; hipMalloc(&Tg, 1024 * 4 * 202 * 8 + 4096); double* Gg; hipMalloc(&Gg, 1024 * 8 * 202 * 8); hipMemset(Gg, 0, 1024 * 8 * 202 * 8);
  for (int blocks : {1, 1024}) {
#define RUN(V, name) k<V><<<blocks, 64, 8 * 202 * 8 + 4 * 202 * 8 + 64 * 16 + 64>>>(out, cyc, 0.01, 0.1, 4.905, 4.0, Gg); hipDeviceSynchronize(); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("blocks %4d %-22s %.1f cycles/step\n", blocks, name, (double)h / STEPS);
    RUN(0, "full") RUN(1, "no ds_write") RUN(2, "no ds_read") RUN(3, "no sin") RUN(5, "whole loop lane 0 only") RUN(6, "stores if lane==0") RUN(7, "stores same address") RUN(8, "G via s_load") RUN(9, "G via s_load, no store")
#define RUN10(ST, name) k10<ST><<<blocks, 64, 8 * 202 * 8 + 4 * 202 * 8 + 64 * 16 + 64>>>(out, cyc, 0.01, 0.1, 4.905, 4.0, Gg, Tg); hipDeviceSynchronize(); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("blocks %4d %-22s %.1f cycles/step\n", blocks, name, (double)h / STEPS);
    RUN10(0, "asm s_load, no store") RUN10(1, "asm s_load, LDS store") RUN10(2, "asm s_load, global store")
  }
  return 0;
}
