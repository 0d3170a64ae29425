// Short-dependency-chain fp64 primitives for the latency-critical loops.
//
// The wave-per-problem kernels issue ONE instruction stream per problem, so the
// time of a rollout/backward step is (instructions per step) x (issue interval).
// ocml's sin() is ~70 instructions (Payne-Hanek branch, sin AND cos kernels,
// selects); a full IEEE fp64 division is ~10.  These replacements are ~21 and ~5
// instructions, accurate to <= 2 ulp on the ranges the models use, which is far
// inside the parity tolerances stated in tests/test_gpu_parity.py.
//
// Coefficients: tools/gen_trig_poly.py (Chebyshev-node interpolation in 60-digit
// arithmetic, rounded to double, max error 2 ulp for |r| <= pi/2).
#pragma once
#include <hip/hip_runtime.h>

namespace mi {

namespace fm {
// sin(r) = r + r^3 * S(r^2), |r| <= pi/2
constexpr double kS[11] = {
    -0x1.5555555555555p-3, 0x1.1111111111111p-7,  -0x1.a01a01a01a01ap-13, 0x1.71de3a556c734p-19,
    -0x1.ae64567f544ddp-26, 0x1.6124613a8672bp-33, -0x1.ae7f3e72ccee8p-41, 0x1.952c76af6edb8p-49,
    -0x1.2f498bc847bdep-57, 0x1.71a067b399258p-66, -0x1.6db87fd9a51a5p-75};
// Cody-Waite split of pi: n*kPi1 and n*kPi2 are exact for |n| < 2^20
constexpr double kPi1 = 0x1.921fb54400000p+1;
constexpr double kPi2 = 0x1.0b4611a600000p-33;
constexpr double kPi3 = 0x1.3198a2e037073p-68;
constexpr double kInvPi = 0x1.45f306dc9c883p-2;
}  // namespace fm

// sin of a reduced argument |r| <= pi/2 (+ small margin)
__device__ __forceinline__ double sin_reduced(double r) {
  const double s = r * r;
  double p = fm::kS[10];
#pragma unroll
  for (int k = 9; k >= 0; --k) p = fma(p, s, fm::kS[k]);
  return fma(r * s, p, r);
}

__device__ __forceinline__ double flip_sign_if_odd(double v, double n) {
  // n is integer-valued; (-1)^n via the parity bit of (int)n xor-ed into the sign
  const int ni = (int)n;
  union { double d; unsigned long long u; } w;
  w.d = v;
  w.u ^= ((unsigned long long)(unsigned)(ni & 1)) << 63;
  return w.d;
}

// sin(x).  Exact-ish reduction for |x| < ~3e6 (beyond that the result is merely
// bounded: such states only occur in diverging line-search trials, whose cost is
// rejected anyway — /root/reference/ilqr.py:315-335).  NaN/inf propagate to NaN.
__device__ __forceinline__ double fast_sin(double x) {
  const double n = rint(x * fm::kInvPi);
  double r = fma(-n, fm::kPi1, x);
  r = fma(-n, fm::kPi2, r);
  r = fma(-n, fm::kPi3, r);
  return flip_sign_if_odd(sin_reduced(r), n);
}

// cos(x) = sin(x + pi/2): reduce by odd multiples of pi/2 so the sin kernel keeps
// full RELATIVE accuracy near the zeros of cos.
__device__ __forceinline__ double fast_cos(double x) {
  const double n = rint(fma(x, fm::kInvPi, 0.5));        // x + pi/2 = r + n*pi
  const double k = fma(2.0, n, -1.0);                    // r = x - (2n-1)*pi/2
  double r = fma(-k, 0.5 * fm::kPi1, x);
  r = fma(-k, 0.5 * fm::kPi2, r);
  r = fma(-k, 0.5 * fm::kPi3, r);
  return flip_sign_if_odd(sin_reduced(r), n);
}

// 1/x: hardware seed + two Newton steps (~1 ulp); no denormal/overflow rescaling.
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  return r;
}

}  // namespace mi
