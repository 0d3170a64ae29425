// Scalar abstraction for the device dynamics: the same templated model code runs
// on plain fp64 (rollouts, finite differences) and on Dual1 (one-directional
// forward-mode derivative, the device analogue of Drake's AutoDiffXd used at
// /root/reference/ilqr.py:253-270, one seeded column per lane).
#pragma once
#include <hip/hip_runtime.h>

#include "fastmath.hpp"

namespace mi {

struct Dual1 {
  double v, d;
  __host__ __device__ Dual1() {}
  __host__ __device__ Dual1(double v_) : v(v_), d(0.0) {}
  __host__ __device__ Dual1(double v_, double d_) : v(v_), d(d_) {}
};

__host__ __device__ inline Dual1 operator+(Dual1 a, Dual1 b) { return {a.v + b.v, a.d + b.d}; }
__host__ __device__ inline Dual1 operator-(Dual1 a, Dual1 b) { return {a.v - b.v, a.d - b.d}; }
__host__ __device__ inline Dual1 operator-(Dual1 a) { return {-a.v, -a.d}; }
__host__ __device__ inline Dual1 operator*(Dual1 a, Dual1 b) { return {a.v * b.v, a.d * b.v + b.d * a.v}; }
__host__ __device__ inline Dual1 operator/(Dual1 a, Dual1 b) {
  const double q = a.v / b.v;
  return {q, (a.d - q * b.d) / b.v};
}
__host__ __device__ inline Dual1 operator+(Dual1 a, double b) { return {a.v + b, a.d}; }
__host__ __device__ inline Dual1 operator+(double a, Dual1 b) { return {a + b.v, b.d}; }
__host__ __device__ inline Dual1 operator-(Dual1 a, double b) { return {a.v - b, a.d}; }
__host__ __device__ inline Dual1 operator-(double a, Dual1 b) { return {a - b.v, -b.d}; }
__host__ __device__ inline Dual1 operator*(Dual1 a, double b) { return {a.v * b, a.d * b}; }
__host__ __device__ inline Dual1 operator*(double a, Dual1 b) { return {a * b.v, a * b.d}; }
__host__ __device__ inline Dual1 operator/(Dual1 a, double b) { return {a.v / b, a.d / b}; }
__host__ __device__ inline Dual1 operator/(double a, Dual1 b) {
  const double q = a / b.v;
  return {q, (-q * b.d) / b.v};
}

__host__ __device__ inline double value_of(double a) { return a; }
__host__ __device__ inline double value_of(Dual1 a) { return a.v; }

__device__ inline double mi_sin(double a) { return fast_sin(a); }
__device__ inline double mi_cos(double a) { return fast_cos(a); }
__device__ inline double mi_rcp(double a) { return fast_rcp(a); }
__device__ inline Dual1 mi_rcp(Dual1 a) { const double r = fast_rcp(a.v); return {r, -(r * r) * a.d}; }
__device__ inline double mi_sqrt(double a) { return sqrt(a); }
__device__ inline Dual1 mi_sqrt(Dual1 a) { const double r = sqrt(a.v); return {r, a.d * fast_rcp(2.0 * r)}; }
__device__ inline double mi_exp(double a) { return exp(a); }
__device__ inline double mi_log1p(double a) { return log1p(a); }
__device__ inline Dual1 mi_sin(Dual1 a) { return {fast_sin(a.v), fast_cos(a.v) * a.d}; }
__device__ inline Dual1 mi_cos(Dual1 a) { return {fast_cos(a.v), -fast_sin(a.v) * a.d}; }
__device__ inline Dual1 mi_exp(Dual1 a) { const double e = exp(a.v); return {e, e * a.d}; }
__device__ inline Dual1 mi_log1p(Dual1 a) { return {log1p(a.v), a.d / (1.0 + a.v)}; }

// Two-directional forward-mode dual (value + derivatives along two seeds): the state Jacobian of a
// two-state closed-loop step in one evaluation (time-parallel Newton rollout, ilqr_small.hpp).
struct Dual2 {
  double v, d0, d1;
  __host__ __device__ Dual2() {}
  __host__ __device__ Dual2(double v_) : v(v_), d0(0.0), d1(0.0) {}
  __host__ __device__ Dual2(double v_, double a_, double b_) : v(v_), d0(a_), d1(b_) {}
};
__host__ __device__ inline Dual2 operator+(Dual2 a, Dual2 b) { return {a.v + b.v, a.d0 + b.d0, a.d1 + b.d1}; }
__host__ __device__ inline Dual2 operator-(Dual2 a, Dual2 b) { return {a.v - b.v, a.d0 - b.d0, a.d1 - b.d1}; }
__host__ __device__ inline Dual2 operator-(Dual2 a) { return {-a.v, -a.d0, -a.d1}; }
__host__ __device__ inline Dual2 operator*(Dual2 a, Dual2 b) { return {a.v * b.v, a.d0 * b.v + b.d0 * a.v, a.d1 * b.v + b.d1 * a.v}; }
__host__ __device__ inline Dual2 operator+(Dual2 a, double b) { return {a.v + b, a.d0, a.d1}; }
__host__ __device__ inline Dual2 operator+(double a, Dual2 b) { return {a + b.v, b.d0, b.d1}; }
__host__ __device__ inline Dual2 operator-(Dual2 a, double b) { return {a.v - b, a.d0, a.d1}; }
__host__ __device__ inline Dual2 operator-(double a, Dual2 b) { return {a - b.v, -b.d0, -b.d1}; }
__host__ __device__ inline Dual2 operator*(Dual2 a, double b) { return {a.v * b, a.d0 * b, a.d1 * b}; }
__host__ __device__ inline Dual2 operator*(double a, Dual2 b) { return {a * b.v, a * b.d0, a * b.d1}; }
__host__ __device__ inline double value_of(Dual2 a) { return a.v; }
// The time-parallel rollout's Newton sweeps need cos only as the SLOPE of sin: fast_sin_slope (fastmath.hpp) - the value bit for
// bit fast_sin, the slope to 1e-10 from the same reduction.  Same-box A/B, round 6 (tools/diag/build_variant.py cc1
// -DMI_DUAL2_CHEAP_COS=1 against =0, three runs each): C2 43.5 -> 44.4 M it/s (kernel 0.1405 -> 0.1376 ms), the same 6193
// iterations per step.
#ifndef MI_DUAL2_CHEAP_COS
#define MI_DUAL2_CHEAP_COS 1
#endif
#if MI_DUAL2_CHEAP_COS
__device__ inline Dual2 mi_sin(Dual2 a) { double c_; const double s_ = fast_sin_slope(a.v, c_); return {s_, c_ * a.d0, c_ * a.d1}; }
#else
__device__ inline Dual2 mi_sin(Dual2 a) { const double c_ = fast_cos(a.v); return {fast_sin(a.v), c_ * a.d0, c_ * a.d1}; }
#endif
__device__ inline Dual2 mi_cos(Dual2 a) { const double s_ = -fast_sin(a.v); return {fast_cos(a.v), s_ * a.d0, s_ * a.d1}; }
__device__ inline Dual2 mi_rcp(Dual2 a) { const double r = fast_rcp(a.v), q = -(r * r); return {r, q * a.d0, q * a.d1}; }

// log(1+exp(z)) = max(z,0) + log1p(exp(-|z|)), overflow-safe and branch-free; same value as the
// two-branch form of oracle/dual.py:softplus.  d/dz = logistic(z).
// MI_SOFTPLUS_SKIP (round 6): once t = exp(-|z|) < 2^-53, fast_log1p01(t) returns t itself, bit for bit (f = t, 2 + f rounds to 2,
// its reciprocal is exactly 0.5, s = t / 2, and the series term s w R ~ t^3 / 12 is below half an ulp of 2 s = t), so a wave whose
// every lane is that far from the contact leaves the logarithm - a third of the step's dependency chain - out: one compare
// and a scalar branch; the same bits either way.  (Branch per wave, not per lane: lanes in contact keep everybody on the long path.)
// MEASURED, and OFF by default: in the one-wave microbenchmark the step gets 7 - 11 % shorter (tools/ubench/chain_step.hip: 715 -> 666 /
// 635 cycles), inside the fused solve kernel it gets 12 % LONGER (same-box A/B, profiles/r06_c4_ab.txt: C4's line search 178.6 k ->
// 201.1 k cycles per iteration) - the branch splits the rollout loop's body, both arms stay resident, and the loop that held its
// values in 293 instructions without a register move now carries 36 v_accvgpr moves in 352 (tools/isa_mix.py).
#ifndef MI_SOFTPLUS_SKIP
#define MI_SOFTPLUS_SKIP 0
#endif
__device__ inline double mi_softplus(double z, const SoftplusPool& c) {
  const double t = fast_exp_nonpos(-fabs(z), c);
#if MI_SOFTPLUS_SKIP
  if (__builtin_amdgcn_ballot_w64(!(t < 0x1p-53)) == 0ull) return fmax(z, 0.0) + t;
#endif
  return fmax(z, 0.0) + fast_log1p01(t, c);
}
__device__ inline double mi_softplus(double z) { return mi_softplus(z, SoftplusPool::literals()); }
__device__ inline Dual1 mi_softplus(Dual1 z) {
  const double t = fast_exp_nonpos(-fabs(z.v));
  const double r = fast_rcp(1.0 + t);
  const double sig = z.v > 0.0 ? r : t * r;
  return {fmax(z.v, 0.0) + fast_log1p01(t), sig * z.d};
}

}  // namespace mi
