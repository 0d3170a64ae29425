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
// exp(r) = 1 + r + r^2 P(r), |r| <= ln2/2 (1 ulp) ; log1p(f) = 2s + s w R(w), s = f/(2+f), w = s^2 <= 0.03 (2 ulp)
// (tools/gen_explog_poly.py)
constexpr double kE[11] = {0x1.0000000000000p-1, 0x1.5555555555557p-3, 0x1.5555555555556p-5, 0x1.111111110ff87p-7,
                           0x1.6c16c16c16212p-10, 0x1.a01a01aca0134p-13, 0x1.a01a01a741b3cp-16, 0x1.71ddffef9e7b1p-19,
                           0x1.27e4da1a3bf04p-22, 0x1.af52906239de4p-26, 0x1.1f75aba0b1e2ep-29};
constexpr double kL[8] = {0x1.5555555555555p-1, 0x1.9999999999a4ep-2, 0x1.24924924736ddp-2, 0x1.c71c720742c1ep-3,
                          0x1.745cf692c6200p-3, 0x1.3b1cb90bb738fp-3, 0x1.0fb0f07f1be07p-3, 0x1.0c90f788b03b1p-3};
constexpr double kLn2Hi = 0x1.62e42fee00000p-1, kLn2Lo = 0x1.a39ef35793c76p-33, kLog2e = 0x1.71547652b82fep+0,
                 kLn2 = 0x1.62e42fefa39efp-1, kSqrt2m1 = 0x1.a827999fcef32p-2;
}  // namespace fm

// Polynomial evaluation.  These kernels run ONE instruction stream per problem at one wave per SIMD: if a dependent fp64 operation
// waited ~8 cycles for its operand while an independent one issues every 4, what a rollout step costs would be the DEPTH of its
// dependency chain - so round 6 assumed, and built Estrin's scheme (MI_POLY_ESTRIN=1: the same polynomial in
// ceil(log2(degree + 1)) + 1 levels of independent multiply-adds for two or three more multiplications) to shorten it.  MEASURED
// (tools/ubench/chain_step.hip, one wave, dependent steps; profiles/r06_chain_step.txt): SLOWER - cart-pole + wall 715 -> 725
// cycles per step, acrobot 670 -> 694 - and less accurate (3 ulp against Horner's 2, tools/ubench/trig_acc.hip).  The reason
// (tools/ubench/issue_interval.hip, profiles/r06_issue_interval.txt): ONE wave issues a v_fma_f64 every 4.04 cycles from
// independent chains and every 4.17 from a single DEPENDENT chain - a dependent fp64 operation issues back to back, there is no
// latency for Estrin's independent multiply-adds to hide, only more instructions to issue.  What a step costs at one wave per
// SIMD is its instruction COUNT.  Horner stays.
#ifndef MI_POLY_ESTRIN
#define MI_POLY_ESTRIN 0
#endif
// c[0] + c[1] x + ... + c[10] x^10
template <class C>
__device__ __forceinline__ double poly10(const C& c, double x) {
#if MI_POLY_ESTRIN
  const double x2 = x * x, x4 = x2 * x2, x8 = x4 * x4;
  const double a0 = fma(c[1], x, c[0]), a1 = fma(c[3], x, c[2]), a2 = fma(c[5], x, c[4]), a3 = fma(c[7], x, c[6]), a4 = fma(c[9], x, c[8]);
  const double b0 = fma(a1, x2, a0), b1 = fma(a3, x2, a2), b2 = fma(c[10], x2, a4);
  return fma(b2, x8, fma(b1, x4, b0));
#else
  double p = c[10];
#pragma unroll
  for (int k = 9; k >= 0; --k) p = fma(p, x, c[k]);
  return p;
#endif
}
// c[0] + c[1] x + ... + c[7] x^7
template <class C>
__device__ __forceinline__ double poly7(const C& c, double x) {
#if MI_POLY_ESTRIN
  const double x2 = x * x, x4 = x2 * x2;
  const double a0 = fma(c[1], x, c[0]), a1 = fma(c[3], x, c[2]), a2 = fma(c[5], x, c[4]), a3 = fma(c[7], x, c[6]);
  return fma(fma(a3, x2, a2), x4, fma(a1, x2, a0));
#else
  double p = c[7];
#pragma unroll
  for (int k = 6; k >= 0; --k) p = fma(p, x, c[k]);
  return p;
#endif
}

// sin of a reduced argument |r| <= pi/2 (+ small margin)
__device__ __forceinline__ double sin_reduced(double r) {
  const double s = r * r;
  return fma(r * s, poly10(fm::kS, s), r);
}

// Round-to-nearest-even integer by the 1.5*2^52 trick: t = y + magic has ulp(t) = 1, so the
// addition's rounding IS the rint; one subtraction recovers the integer as a double, and the low
// mantissa word of t is the integer in two's complement.  v_rndne_f64 + v_cvt_i32_f64 are
// quarter-rate (16-cycle) instructions; these are full rate.  Valid for |y| < 2^51 (beyond that
// the callers' results are garbage-but-rejected, see fast_sin).
constexpr double kRoundMagic = 0x1.8p52;
struct Rounded { double n; int lo; };
__device__ __forceinline__ Rounded round_magic(double t) {   // t = y + kRoundMagic, already rounded
  Rounded r;
  r.n = t - kRoundMagic;
  r.lo = __double2loint(t);
  return r;
}
// rint(a*b): the product is rounded ONCE, inside the fma
__device__ __forceinline__ Rounded round_mul(double a, double b) { return round_magic(fma(a, b, kRoundMagic)); }

__device__ __forceinline__ double flip_sign_if_odd(double v, int n_lo) {
  // (-1)^n: the parity bit of n xor-ed into the sign
  union { double d; unsigned long long u; } w;
  w.d = v;
  w.u ^= ((unsigned long long)(unsigned)n_lo) << 63;
  return w.d;
}

// sin(x).  Exact-ish reduction for |x| < ~3e6 (beyond that the result is merely
// bounded: such states only occur in diverging line-search trials, whose cost is
// rejected anyway — /root/reference/ilqr.py:315-335).  NaN/inf propagate to NaN.
__device__ __forceinline__ double fast_sin(double x) {
  const Rounded k = round_mul(x, fm::kInvPi);
  const double n = k.n;
  double r = fma(-n, fm::kPi1, x);
  r = fma(-n, fm::kPi2, r);
  r = fma(-n, fm::kPi3, r);
  return flip_sign_if_odd(sin_reduced(r), k.lo);
}

// sin(x) - bit for bit fast_sin(x) - and cos(x) to ~1e-10 ABSOLUTE from the SAME reduction (Taylor through r^14 on |r| <= pi/2,
// remainder (pi/2)^16 / 16! = 7e-11): 8 multiply-adds instead of fast_cos's 20 instructions.  For callers that need the cosine
// only as a SLOPE - the time-parallel rollout's Newton sweeps (dual.hpp: mi_sin(Dual2)), whose fixed point is defined by the
// values alone: an error dG in the Jacobian leaves dG x (the last update < 1e-7) in the trajectory, far below its 1e-11 guard.
__device__ __forceinline__ double fast_sin_slope(double x, double& slope) {
  const Rounded k = round_mul(x, fm::kInvPi);
  const double n = k.n;
  double r = fma(-n, fm::kPi1, x);
  r = fma(-n, fm::kPi2, r);
  r = fma(-n, fm::kPi3, r);
  const double s = r * r;
  double q = -0x1.93974a8c07c9dp-37;      // -1/14!
  q = fma(q, s, 0x1.1eed8eff8d898p-29);   //  1/12!
  q = fma(q, s, -0x1.27e4fb7789f5cp-22);  // -1/10!
  q = fma(q, s, 0x1.a01a01a01a01ap-16);   //  1/8!
  q = fma(q, s, -0x1.6c16c16c16c17p-10);  // -1/6!
  q = fma(q, s, 0x1.5555555555555p-5);    //  1/4!
  q = fma(q, s, -0.5);
  slope = flip_sign_if_odd(fma(q, s, 1.0), k.lo);
  return flip_sign_if_odd(sin_reduced(r), k.lo);
}

// cos(x) = sin(x + pi/2): reduce by odd multiples of pi/2 so the sin kernel keeps
// full RELATIVE accuracy near the zeros of cos.
__device__ __forceinline__ double fast_cos(double x) {
  const Rounded kn = round_magic(fma(x, fm::kInvPi, 0.5) + kRoundMagic);      // x + pi/2 = r + n*pi
  const double n = kn.n;
  const double k = fma(2.0, n, -1.0);                    // r = x - (2n-1)*pi/2
  double r = fma(-k, 0.5 * fm::kPi1, x);
  r = fma(-k, 0.5 * fm::kPi2, r);
  r = fma(-k, 0.5 * fm::kPi3, r);
  return flip_sign_if_odd(sin_reduced(r), kn.lo);
}

// sin(x) or cos(x), chosen PER LANE, in one instruction stream: bitwise fast_sin(x) / fast_cos(x).  Both reduce x by
// k * pi/2 (k = 2n for the sine: 2n * (pi1/2) = n * pi1 exactly; k = 2n - 1 for the cosine) and run the same kernel, so a
// wave can evaluate sines on some lanes and cosines on others for the price of one of them.
__device__ __forceinline__ double fast_sin_or_cos(double x, bool is_cos) {
  const double ts = fma(x, fm::kInvPi, kRoundMagic);                 // fast_sin: the product rounded once, inside the fma
  const double tc = fma(x, fm::kInvPi, 0.5) + kRoundMagic;           // fast_cos
  const Rounded kn = round_magic(is_cos ? tc : ts);
  const double k = fma(2.0, kn.n, is_cos ? -1.0 : 0.0);
  double r = fma(-k, 0.5 * fm::kPi1, x);
  r = fma(-k, 0.5 * fm::kPi2, r);
  r = fma(-k, 0.5 * fm::kPi3, r);
  return flip_sign_if_odd(sin_reduced(r), kn.lo);
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

// The constants of exp / log1p as an argument.  fp64 VALU instructions cannot encode 64-bit literals, so each
// constant of a polynomial costs two s_mov_b32 - and the contact model's softplus brings 25 of them on top of the
// trig ones, more than the scalar file holds next to the kernel arguments: inside the rollout loop the compiler
// re-materializes them every step (26 constants = 52 scalar moves per two steps, 13 % of the loop's issue slots
// at one wave per SIMD).  A caller with a long loop loads a pool ONCE into vector registers (opaque to the
// optimizer, see SoftplusPool::in_vgprs) and passes it down; everybody else passes the literals.
struct SoftplusPool {
  double E[11], L[8], ln2;
  __device__ __forceinline__ static SoftplusPool literals() {
    SoftplusPool p;
#pragma unroll
    for (int k = 0; k < 11; ++k) p.E[k] = fm::kE[k];
#pragma unroll
    for (int k = 0; k < 8; ++k) p.L[k] = fm::kL[k];
    p.ln2 = fm::kLn2;
    return p;
  }
  // the same values pinned in VGPRs at this point of the program (volatile: not hoisted to the kernel entry,
  // where they would be live across everything and end up in AGPRs)
  __device__ __forceinline__ static SoftplusPool in_vgprs() {
    SoftplusPool p = literals();
#pragma unroll
    for (int k = 0; k < 11; ++k) asm volatile("" : "+v"(p.E[k]));
#pragma unroll
    for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(p.L[k]));
    asm volatile("" : "+v"(p.ln2));
    return p;
  }
};

// exp(x) for x <= 0 (the only range the contact model needs): Cody-Waite reduction by ln2,
// degree-10 polynomial, scale by 2^n (underflows cleanly to 0 for very negative x).
__device__ __forceinline__ double fast_exp_nonpos(double x, const SoftplusPool& c) {
  x = fmax(x, -745.0);
  const Rounded kn = round_mul(x, fm::kLog2e);
  const double nd = kn.n;
  double r = fma(-nd, fm::kLn2Hi, x);
  r = fma(-nd, fm::kLn2Lo, r);
  const double p = poly10(c.E, r);
  const double e = 1.0 + fma(r * r, p, r);
  return ldexp(e, kn.lo);
}
__device__ __forceinline__ double fast_exp_nonpos(double x) { return fast_exp_nonpos(x, SoftplusPool::literals()); }

// log1p(y) for 0 <= y <= 1: fold 1+y into [sqrt(1/2), sqrt 2] exactly, then 2 atanh(s).
__device__ __forceinline__ double fast_log1p01(double y, const SoftplusPool& c) {
  const bool hi = y > fm::kSqrt2m1;
  const double f = hi ? 0.5 * (y - 1.0) : y;           // 1+y = 2(1+f) resp. 1+f, both exact
  const double s = f * fast_rcp(2.0 + f);
  const double w = s * s;
  const double R = poly7(c.L, w);
  const double l = fma(s * w, R, 2.0 * s);
  return hi ? l + c.ln2 : l;
}
__device__ __forceinline__ double fast_log1p01(double y) { return fast_log1p01(y, SoftplusPool::literals()); }

}  // namespace mi
