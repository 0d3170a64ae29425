// "Wide" continuation kernel for n = 2 models: FOUR wavefronts per problem, one time step per lane.
//
// Why it exists: a launch of the wave-per-problem kernel lasts as long as its slowest problem -
// on BASELINE's C2 12 iterations against a mean of 6 - and while that problem iterates, three quarters of the chip
// idle.  Splitting a problem over more waves only pays once the SIMDs are free, and whether they are must not depend
// on timing (results are held bitwise between runs, batch positions, shards and batch sizes).  So the split is a rule
// on the ITERATION INDEX alone: iterations 1..phase_cap of every solve run in ilqr_small_kernel (one wave, four steps
// per lane) exactly as before; problems that have not converged by then are listed by that kernel (KArgs::cont_*) and
// continued here - same algorithm (ilqr.py:692-708), same stages, the two passes over time cut over 256 lanes:
//   * rollout (ilqr.py:306-327) = Newton on the trajectory like rollout_newton_impl, one step per lane: one model
//     evaluation per lane per sweep instead of four; the affine-map scan runs inside every wave (DPP) and the waves'
//     totals cross through LDS (one barrier), likewise the sweep's convergence test and the next lane's guess;
//   * backward pass (:623-667) = the associative Riccati scan of backward_scan with one element per lane and the waves'
//     totals folded through LDS; the reference recursion then runs for ONE step per lane.
// A trial that is not accepted (ilqr.py:330-335) falls back to the single-wave line search of ilqr_small.hpp on wave
// 0 (candidate pass, sequential re-roll) while the other waves wait - rare (9 of 6193 iterations on C2).
#pragma once
#include "ilqr_small.hpp"

namespace mi {

constexpr int kWideWaves = 4;
constexpr int kWideThreads = 64 * kWideWaves;

// LDS exchange area (inside WS::dump, which only the single-wave fallback uses otherwise): doubles
//   [0, 64)    A: per wave 16 - scan totals                      (barrier "A")
//   [64, 96)   B: per wave 8  - maxima / sums / the wave's first guess   (barrier "B")
//   [96, 104)  C: fallback results of wave 0
struct WideX {
  double *A, *B, *C;
};
__device__ __forceinline__ WideX wide_xch(const WS& w) { return {w.dump, w.dump + 64, w.dump + 96}; }

__device__ __forceinline__ void aff2_identity_dev(Aff2& t) {
  t.G[0][0] = 0.0; t.G[0][1] = 0.0; t.G[1][0] = 0.0; t.G[1][1] = 0.0; t.c[0] = 0.0; t.c[1] = 0.0;
}
// Totals of the waves before `wave` (deviation form), composed in time order; every wave runs the same instruction
// stream (the branch is wave-uniform).
__device__ __forceinline__ void aff2_fold_totals(Aff2& T, const double* A, int wave) {
  aff2_identity_dev(T);
#pragma unroll
  for (int v = 0; v < kWideWaves - 1; ++v) {
    if (v < wave) {
      Aff2 tv, o;
      const double* s = A + 16 * v;
      tv.G[0][0] = s[0]; tv.G[0][1] = s[1]; tv.G[1][0] = s[2]; tv.G[1][1] = s[3]; tv.c[0] = s[4]; tv.c[1] = s[5];
      aff2_compose_dev(o, tv, T);          // tv is LATER in time than what T holds
      T = o;
    }
  }
}
__device__ __forceinline__ void aff2_publish_total(const Aff2& P, double* A, int wave, int lane) {
  if (lane == 63) {
    double* s = A + 16 * wave;
    s[0] = P.G[0][0]; s[1] = P.G[0][1]; s[2] = P.G[1][0]; s[3] = P.G[1][1]; s[4] = P.c[0]; s[5] = P.c[1];
  }
}
__device__ __forceinline__ double wave_max(double v) {
  v = fmax(v, row_ror_f64<8>(v));
  v = fmax(v, row_ror_f64<4>(v));
  v = fmax(v, row_ror_f64<2>(v));
  v = fmax(v, row_ror_f64<1>(v));
  return fmax(fmax(readlane_f64(v, 0), readlane_f64(v, 16)), fmax(readlane_f64(v, 32), readlane_f64(v, 48)));
}

// The eps = 1 trial (or any eps) rolled out parallel in time over 4 waves, cost / acceptance / commit / linearization
// folded into the final pass (the fuse = 2 form of rollout_newton_impl).  Returns the same code on every wave.
template <class M, int JAC>
__device__ inline int rollout_newton_wide(const WS& w, const Consts<M>& c, const KArgs& a, const double* x0r, double eps,
                                          double L_last, double& L_out) {
  constexpr int n = 2, m = 1;
  static_assert(M::n == 2 && M::m == 1, "2-state closed loop");
  using Ly = Lay<n, m>;
  const WideX xc = wide_xch(w);
  const int N = w.N, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, steps = N - 1;
  const int t = tid;
  const bool valid = t < steps;
  const double* g = w.G + (valid ? t : 0) * Ly::GS;
  const double xb[n] = {g[Ly::XB + 0], g[Ly::XB + 1]}, Kk[n] = {g[Ly::KK + 0], g[Ly::KK + 1]};
  const double dd = g[Ly::UB] - eps * g[Ly::KAP];
  double X[n];
  // ---- predictor: the linearized closed loop around the nominal trajectory (see rollout_newton_impl)
  {
    const double* jr = w.J + (valid ? t : 0) * Ly::JS;
    const double kap = eps * g[Ly::KAP];
    Aff2 loc;
#pragma unroll
    for (int i = 0; i < n; ++i) {
      const double fui = jr[Ly::FU + i];
#pragma unroll
      for (int j = 0; j < n; ++j) loc.G[i][j] = fma(-fui, Kk[j], jr[Ly::FX + i * n + j]) - ((i == j) ? 1.0 : 0.0);
      loc.c[i] = -fui * kap;
    }
    Aff2 P = loc;
    aff2_prefix_dpp(P);
    aff2_publish_total(P, xc.A, wave, lane);
    team_barrier();
    Aff2 T, Pf;
    aff2_fold_totals(T, xc.A, wave);
    aff2_compose_dev(Pf, P, T);
    const double d0[n] = {x0r[0] - w.G[Ly::XB + 0], x0r[1] - w.G[Ly::XB + 1]};
    double ds[n];
#pragma unroll
    for (int i = 0; i < n; ++i) {
      const double ye = fma(Pf.G[i][0], d0[0], fma(Pf.G[i][1], d0[1], Pf.c[i] + d0[i]));
      const double yT = fma(T.G[i][0], d0[0], fma(T.G[i][1], d0[1], T.c[i] + d0[i]));     // end state of the previous wave
      const double yp = dpp_f64_or_zero<0x138, 0xF>(ye);                                   // wave_shr:1
      ds[i] = (lane == 0) ? (wave == 0 ? d0[i] : yT) : yp;
    }
    X[0] = xb[0] + ds[0]; X[1] = xb[1] + ds[1];
    if (t == 0) { X[0] = x0r[0]; X[1] = x0r[1]; }
  }
  // the next wave's first guess, for this wave's lane 63
  auto publish_first = [&](double wmax) __attribute__((always_inline)) {
    if (lane == 0) { double* s = xc.B + 8 * wave; s[0] = X[0]; s[1] = X[1]; s[2] = wmax; }
  };
  publish_first(0.0);
  team_barrier();
  constexpr int kMaxSweeps = MI_NEWTON_MAX_SWEEPS;
  constexpr double kTol = 1e-7, kFrozenTol = 5e-4;
  double Gs[n][n] = {{1.0, 0.0}, {0.0, 1.0}};
  double prev_upd = __builtin_inf();
  bool have_g = false, last_frozen = false, converged = false;
  double nxw[n] = {0.0, 0.0};
  for (int sweep = 0; sweep < kMaxSweeps && !converged; ++sweep) {
    const bool frozen = have_g && !last_frozen && prev_upd < kFrozenTol;            // uniform over the workgroup
    {
      const double* s = xc.B + 8 * (wave + 1 < kWideWaves ? wave + 1 : wave);
      nxw[0] = s[0]; nxw[1] = s[1];
    }
    const double sx0 = dpp_f64_or_zero<0x130, 0xF>(X[0]), sx1 = dpp_f64_or_zero<0x130, 0xF>(X[1]);   // wave_shl:1
    const double nx0 = (lane == 63) ? nxw[0] : sx0, nx1 = (lane == 63) ? nxw[1] : sx1;
    Aff2 loc;
    if (!frozen) {
      Dual2 xd[n] = {Dual2(X[0], 1.0, 0.0), Dual2(X[1], 0.0, 1.0)};
      Dual2 ud[m] = {dd - (Kk[0] * (xd[0] - xb[0]) + Kk[1] * (xd[1] - xb[1]))};
      Dual2 xn[n];
      M::template step<Dual2>(xd, ud, xn, a.params, a.dt);
#pragma unroll
      for (int i = 0; i < n; ++i) { Gs[i][0] = xn[i].d0; Gs[i][1] = xn[i].d1; loc.c[i] = xn[i].v - (i == 0 ? nx0 : nx1); }
    } else {
      double u[m] = {dd - (Kk[0] * (X[0] - xb[0]) + Kk[1] * (X[1] - xb[1]))};
      double xn[n];
      M::template step<double>(X, u, xn, a.params, a.dt);
#pragma unroll
      for (int i = 0; i < n; ++i) loc.c[i] = xn[i] - (i == 0 ? nx0 : nx1);
    }
    Aff2 P;
#pragma unroll
    for (int i = 0; i < n; ++i) { P.c[i] = loc.c[i]; P.G[i][0] = Gs[i][0] - (i == 0 ? 1.0 : 0.0); P.G[i][1] = Gs[i][1] - (i == 1 ? 1.0 : 0.0); }
    aff2_prefix_dpp(P);
    aff2_publish_total(P, xc.A, wave, lane);
    team_barrier();                                                               // "A"
    Aff2 T;
    aff2_fold_totals(T, xc.A, wave);
    // this lane's correction = the offset of the prefix that ends at the previous lane (applied to d_0 = 0)
    double ds[n];
#pragma unroll
    for (int i = 0; i < n; ++i) {
      const double pc = fma(P.G[i][0], T.c[0], fma(P.G[i][1], T.c[1], P.c[i] + T.c[i]));
      const double yp = dpp_f64_or_zero<0x138, 0xF>(pc);
      ds[i] = (lane == 0) ? (wave == 0 ? 0.0 : T.c[i]) : yp;
    }
    double upd = 0.0;
    if (valid) upd = (ds[0] == ds[0] && ds[1] == ds[1]) ? fmax(fabs(ds[0]), fabs(ds[1])) : __builtin_inf();
    X[0] += ds[0]; X[1] += ds[1];
    upd = wave_max(upd);
    publish_first(upd);
    team_barrier();                                                               // "B"
    upd = fmax(fmax(xc.B[2], xc.B[8 + 2]), fmax(xc.B[16 + 2], xc.B[24 + 2]));
    have_g = true; last_frozen = frozen; prev_upd = upd;
    converged = upd < kTol;
  }
  if (!converged) return NEWTON_FAILED;
  // ---- final pass: the plain fp64 step from the converged guess, with cost, acceptance, commit and linearization
  double u[m] = {0.0}, xn[n] = {0.0, 0.0};
  double cost = 0.0, dvs = 0.0, open_ = 0.0;
  if (valid) {
    u[0] = dd - (Kk[0] * (X[0] - xb[0]) + Kk[1] * (X[1] - xb[1]));
    M::template step<double>(X, u, xn, a.params, a.dt);
    cost = stage_cost<M>(c, X, u);                              // ilqr.py:325
    dvs = g[Ly::DV];                                            // :326
    if (t == steps - 1) cost += terminal_cost<M>(c, xn);        // :327
  }
  {
    // a-posteriori guard: this lane's end state against the next lane's converged start (see rollout_newton_impl)
    constexpr double kEdgeTol = 1e-11;
    const double* s = xc.B + 8 * (wave + 1 < kWideWaves ? wave + 1 : wave);
    const double sx0 = dpp_f64_or_zero<0x130, 0xF>(X[0]), sx1 = dpp_f64_or_zero<0x130, 0xF>(X[1]);
    const double nx0 = (lane == 63) ? s[0] : sx0, nx1 = (lane == 63) ? s[1] : sx1;
    if (t + 1 < steps) {
      const double d = fmax(fabs(xn[0] - nx0), fabs(xn[1] - nx1));
      const double sc = fmax(1.0, fmax(fabs(xn[0]), fabs(xn[1])));
      open_ = (d <= kEdgeTol * sc) ? 0.0 : 1.0;
    }
  }
  const double wc = wave_sum(cost), wd = wave_sum(dvs), wo = wave_max(open_);
  team_barrier();                                               // (everybody has read B's first guesses)
  if (lane == 0) { double* s = xc.B + 8 * wave; s[3] = wc; s[4] = wd; s[5] = wo; }
  team_barrier();
  const double L = (xc.B[3] + xc.B[8 + 3]) + (xc.B[16 + 3] + xc.B[24 + 3]);
  const double dv_all = (xc.B[4] + xc.B[8 + 4]) + (xc.B[16 + 4] + xc.B[24 + 4]);
  const double any_open = fmax(fmax(xc.B[5], xc.B[8 + 5]), fmax(xc.B[16 + 5], xc.B[24 + 5]));
  if (any_open != 0.0) return NEWTON_FAILED;
  const double ex = -eps * (1.0 - eps / 2.0) * dv_all;
  L_out = L;
  if (!((L_last - L) > a.gamma * ex)) return NEWTON_REJECTED;    // ilqr.py:330-331
  if (valid) {
    double* gw = w.G + t * Ly::GS;
    gw[Ly::XB + 0] = X[0]; gw[Ly::XB + 1] = X[1];
    gw[Ly::UB] = u[0];
    if (t == steps - 1) { gw[Ly::GS + Ly::XB + 0] = xn[0]; gw[Ly::GS + Ly::XB + 1] = xn[1]; }
    double* j = w.J + t * Ly::JS;
#pragma unroll
    for (int col = 0; col < n + m; ++col) {
      double d[n];
      jac_column<M, JAC>(X, u, col, a, d);
#pragma unroll
      for (int i = 0; i < n; ++i) {
        if (col < n) j[Ly::FX + i * n + col] = d[i];
        else j[Ly::FU + i * m + (col - n)] = d[i];
      }
    }
  }
  return NEWTON_ACCEPTED;
}

// Backward Riccati pass (ilqr.py:623-667, cost expansion :161-206 fused) as the associative scan of backward_scan
// with ONE element per lane over 4 waves; lane g of the workgroup owns element 255 - g (reverse time order, so that the
// suffix scan over time is a prefix scan over the lanes).
template <class M>
__device__ inline void backward_scan_wide(const WS& w, const Consts<M>& c) {
  constexpr int n = 2, m = 1;
  static_assert(M::n == 2 && M::m == 1, "n = 2");
  using Ly = Lay<n, m>;
  const WideX xc = wide_xch(w);
  const int N = w.N, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int e0 = kWideThreads - 1 - tid;                        // elements 0..N-2 steps, N-1 terminal, beyond: identity
  double min_pivot = __builtin_inf();
  double Q2[n][n], R2[m][m];
#pragma unroll
  for (int i = 0; i < n; ++i)
#pragma unroll
    for (int j = 0; j < n; ++j) Q2[i][j] = 2.0 * c.Q[i][j];
  R2[0][0] = 2.0 * c.R[0][0];
  const double R2i = fast_rcp(R2[0][0]);
  const int tg = e0 < N ? e0 : N - 1, tj = e0 < N - 1 ? e0 : N - 2;
  const double* g = w.G + tg * Ly::GS;
  const double* jr = w.J + tj * Ly::JS;
  double qx[n], qfx[n][n], qfu[n];
#pragma unroll
  for (int i = 0; i < n; ++i) {
    qx[i] = g[Ly::XB + i];
    qfu[i] = jr[Ly::FU + i * m];
#pragma unroll
    for (int j = 0; j < n; ++j) qfx[i][j] = jr[Ly::FX + i * n + j];
  }
  const double qu = g[Ly::UB];
  const bool is_step = e0 < N - 1, is_term = e0 == N - 1;
  RicElem<n> S, T, U;
#pragma unroll
  for (int i = 0; i < n; ++i) {
    double s = -c.qn[i], sf = -c.qfn[i];
#pragma unroll
    for (int j = 0; j < n; ++j) {
      s += Q2[i][j] * qx[j];
      sf += (2.0 * c.Qf[i][j]) * qx[j];
      const double idm = (i == j) ? 1.0 : 0.0;
      S.A[i][j] = is_step ? qfx[i][j] : (is_term ? 0.0 : idm);
      S.J[i][j] = is_step ? Q2[i][j] : (is_term ? 2.0 * c.Qf[i][j] : 0.0);
      S.C[i][j] = is_step ? (qfu[i] * R2i) * qfu[j] : 0.0;
    }
    S.e[i] = is_step ? -s : (is_term ? -sf : 0.0);
    S.b[i] = is_step ? -qfu[i] * qu : 0.0;
  }
  auto level = [&](auto ctrl, auto rows) __attribute__((always_inline)) {
    ric_fetch_dpp<decltype(ctrl)::value, decltype(rows)::value, n>(T, S);
    ric_combine<n>(U, S, T, min_pivot);
    S = U;
  };
  using std::integral_constant;
  level(integral_constant<int, 0x111>{}, integral_constant<int, 0xF>{});
  level(integral_constant<int, 0x112>{}, integral_constant<int, 0xF>{});
  level(integral_constant<int, 0x114>{}, integral_constant<int, 0xF>{});
  level(integral_constant<int, 0x118>{}, integral_constant<int, 0xF>{});
  level(integral_constant<int, 0x142>{}, integral_constant<int, 0xA>{});
  level(integral_constant<int, 0x143>{}, integral_constant<int, 0xC>{});   // (full: lane 63's total is folded by the later waves)
  if (lane == 63) {
    double* s = xc.A + 16 * wave;
    s[0] = S.A[0][0]; s[1] = S.A[0][1]; s[2] = S.A[1][0]; s[3] = S.A[1][1]; s[4] = S.b[0]; s[5] = S.b[1];
    s[6] = S.C[0][0]; s[7] = S.C[0][1]; s[8] = S.C[1][1]; s[9] = S.e[0]; s[10] = S.e[1];
    s[11] = S.J[0][0]; s[12] = S.J[0][1]; s[13] = S.J[1][1];
  }
  team_barrier();
  // R = total of the waves before this one (they hold LATER times): Tot_{wave-1} (x) ... (x) Tot_0
  RicElem<n> R;
  ric_identity(R);
#pragma unroll
  for (int v = 0; v < kWideWaves - 1; ++v) {
    if (v < wave) {
      const double* s = xc.A + 16 * v;
      RicElem<n> tv;
      tv.A[0][0] = s[0]; tv.A[0][1] = s[1]; tv.A[1][0] = s[2]; tv.A[1][1] = s[3]; tv.b[0] = s[4]; tv.b[1] = s[5];
      tv.C[0][0] = s[6]; tv.C[0][1] = s[7]; tv.C[1][0] = s[7]; tv.C[1][1] = s[8]; tv.e[0] = s[9]; tv.e[1] = s[10];
      tv.J[0][0] = s[11]; tv.J[0][1] = s[12]; tv.J[1][0] = s[12]; tv.J[1][1] = s[13];
      if (v == 0) R = tv;
      else { ric_combine<n>(U, tv, R, min_pivot); R = U; }      // tv is EARLIER in time than what R holds
    }
  }
  ric_combine<n, true>(U, S, R, min_pivot);                     // only (J, eta) are read from here on
  // value function at the right edge of this lane's element = (J, -eta) of the scan value one lane down
  double Vx[n], Vxx[n][n];
#pragma unroll
  for (int i = 0; i < n; ++i) {
    const double pe = dpp_f64_or_zero<0x138, 0xF>(U.e[i]);      // wave_shr:1
    Vx[i] = -((lane == 0) ? R.e[i] : pe);
#pragma unroll
    for (int j = 0; j < n; ++j) {
      const double pj = dpp_f64_or_zero<0x138, 0xF>(U.J[i][j]);
      Vxx[i][j] = (lane == 0) ? R.J[i][j] : pj;
    }
  }
  // the reference recursion for this lane's own step (ilqr.py:651-667)
  if (is_step) {
    BRegs<M> r;
#pragma unroll
    for (int i = 0; i < n; ++i) {
      double s = -c.qn[i];
#pragma unroll
      for (int j = 0; j < n; ++j) { s += Q2[i][j] * qx[j]; r.fx[i][j] = qfx[i][j]; }
      r.lx[i] = s;
      r.fu[i][0] = qfu[i];
    }
    r.lu[0] = R2[0][0] * qu;
    backward_step<M>(r, c, Q2, R2, Vx, Vxx, w.G + e0 * Ly::GS);
  }
}

template <class M, int JAC>
__global__ void __launch_bounds__(kWideThreads) ilqr_wide_kernel(const KArgs a) {
  constexpr int n = M::n, m = M::m;
  static_assert(n == 2 && m == 1, "n = 2 models");
  using Ly = Lay<n, m>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int count = __hip_atomic_load(a.cont_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // one workgroup per compute unit (the launch asks for the whole LDS), each takes the listed problems blockIdx.x,
  // blockIdx.x + gridDim.x, ...: with at most as many unfinished problems as compute units every problem has a CU -
  // four SIMDs - to itself
  for (int entry = blockIdx.x; entry < count; entry += gridDim.x) {
  const int b = a.cont_list[entry];
  const int N = a.N;
  WS w = carve<n, m>(smem, N, a.n_store);
  const WideX xc = wide_xch(w);
  const size_t oX = (size_t)b * n * N, oU = (size_t)b * m * (N - 1), oK = (size_t)b * m * n * (N - 1);
  const size_t oFx = (size_t)b * n * n * (N - 1), oFu = (size_t)b * n * m * (N - 1), oT = (size_t)b * (N - 1);
  Consts<M> c;
  c.load(a.costmat);
  double x0r[n];
#pragma unroll
  for (int i = 0; i < n; ++i) x0r[i] = a.x0[(size_t)b * n + i];
  // stage the state the first phase wrote back: one time step per thread, whole records
  for (int t = tid; t < N; t += kWideThreads) {
    double* g = w.G + t * Ly::GS;
    double* jr = w.J + t * Ly::JS;
    const bool st = t < N - 1;
    g[Ly::XB + 0] = a.x_bar[oX + t]; g[Ly::XB + 1] = a.x_bar[oX + (size_t)N + t];
    g[Ly::UB] = st ? a.u_bar[oU + t] : 0.0;
    g[Ly::KK + 0] = st ? a.K[oK + t] : 0.0; g[Ly::KK + 1] = st ? a.K[oK + (size_t)(N - 1) + t] : 0.0;
    g[Ly::KAP] = st ? a.kappa[oU + t] : 0.0;
    g[Ly::DV] = st ? a.dV[oT + t] : 0.0;
#pragma unroll
    for (int k = 0; k < n * n; ++k) jr[Ly::FX + k] = st ? a.fx[oFx + (size_t)k * (N - 1) + t] : 0.0;
#pragma unroll
    for (int k = 0; k < n * m; ++k) jr[Ly::FU + k] = st ? a.fu[oFu + (size_t)k * (N - 1) + t] : 0.0;
  }
  double L = a.cost[b];
  int it_this = a.iters[b], ls_total = a.ls_trials[b], status = MI_STATUS_CONVERGED;
  double* hist = a.hist + (size_t)b * a.hist_cap * 4;
  long long c_ls = 0, c_bp = 0;
  const long long c_begin = clock64();
  team_barrier();
  double improvement = __builtin_inf();
  long long c_prev = c_begin;
  while (improvement > a.delta) {                                // ilqr.py:692 (the first phase left improvement > delta)
    if (it_this >= a.max_iters) { status = MI_STATUS_MAX_ITERS; break; }
    double L_new = 0.0, eps = 1.0;
    int trials = 1;
    const long long c0 = c_prev;
    const int nr = rollout_newton_wide<M, JAC>(w, c, a, x0r, 1.0, L, L_new);
    if (nr != NEWTON_ACCEPTED) {
      // rejected (or not converged): the reference's sequential line search, on wave 0 alone
      team_barrier();
      if (wave == 0) {
        int slot = 0, fused = 0;
        const bool ok = linesearch<M, JAC>(w, c, a, x0r, L, false, 0, L_new, eps, trials, slot, fused, false, true);
        wave_sync();
        if (ok) commit_trial<n, m>(w, slot);                     // ilqr.py:375-376
        wave_sync();
        if (lane == 0) { xc.C[0] = ok ? 1.0 : 0.0; xc.C[1] = L_new; xc.C[2] = eps; xc.C[3] = (double)trials; }
      }
      team_barrier();
      const bool ok = xc.C[0] != 0.0;
      L_new = xc.C[1]; eps = xc.C[2]; trials = (int)xc.C[3];
      ls_total += trials;
      if (!ok) { status = MI_STATUS_LINESEARCH_FAILED; break; }
      jac_rounds<M, JAC>(w, a, wave, kWideWaves, lane);          // ilqr.py:370 at the accepted trajectory
    } else {
      ls_total += 1;
    }
    team_barrier();
    const long long c2 = clock64();
    backward_scan_wide<M>(w, c);                                 // ilqr.py:697
    team_barrier();
    const long long c3 = clock64();
    c_prev = c3;
    c_ls += c2 - c0; c_bp += c3 - c2;
    if (tid == 0 && it_this < a.hist_cap) {
      hist[4 * it_this + 0] = L_new; hist[4 * it_this + 1] = eps;
      hist[4 * it_this + 2] = (double)trials; hist[4 * it_this + 3] = 100.0;
      double* ic = a.iter_cyc + ((size_t)b * a.hist_cap + it_this) * 4;
      ic[0] = (double)(c2 - c0); ic[1] = 0.0; ic[2] = (double)(c3 - c2); ic[3] = (double)(c3 - c0);
    }
    improvement = L - L_new;                                     // ilqr.py:706
    L = L_new;
    it_this += 1;
  }
  team_barrier();
  // write-back, one time step per thread (like the n <= 2 path of ilqr_small_kernel)
  for (int t = tid; t < N; t += kWideThreads) {
    const double* g = w.G + t * Ly::GS;
    const double* jr = w.J + t * Ly::JS;
    const double xb0 = g[Ly::XB + 0], xb1 = g[Ly::XB + 1], ub = g[Ly::UB], k0 = g[Ly::KK + 0], k1 = g[Ly::KK + 1], kap = g[Ly::KAP], dv = g[Ly::DV];
    double fxr[n * n], fur[n * m];
#pragma unroll
    for (int k = 0; k < n * n; ++k) fxr[k] = jr[Ly::FX + k];
#pragma unroll
    for (int k = 0; k < n * m; ++k) fur[k] = jr[Ly::FU + k];
    wt_store(&a.x_bar[oX + t], xb0); wt_store(&a.x_bar[oX + (size_t)N + t], xb1);
    if (a.sink_x != nullptr) { a.sink_x[oX + t] = xb0; a.sink_x[oX + (size_t)N + t] = xb1; if (t < N - 1) a.sink_u[oU + t] = ub; }
    if (t < N - 1) {
      wt_store(&a.u_bar[oU + t], ub);
#pragma unroll
      for (int k = 0; k < n * n; ++k) wt_store(&a.fx[oFx + (size_t)k * (N - 1) + t], fxr[k]);
#pragma unroll
      for (int k = 0; k < n * m; ++k) wt_store(&a.fu[oFu + (size_t)k * (N - 1) + t], fur[k]);
      wt_store(&a.K[oK + t], k0); wt_store(&a.K[oK + (size_t)(N - 1) + t], k1);
      wt_store(&a.kappa[oU + t], kap);
      wt_store(&a.dV[oT + t], dv);
    }
  }
  if (tid == 0) {
    a.cost[b] = L; a.iters[b] = it_this; a.status[b] = status; a.ls_trials[b] = ls_total;
    if (a.sink_cost != nullptr) a.sink_cost[b] = L;
    a.prof[4 * b + 0] += c_ls; a.prof[4 * b + 2] += c_bp; a.prof[4 * b + 3] += clock64() - c_begin;
  }
  team_barrier();                                              // (the next problem reuses the LDS arrays)
  }
}

}  // namespace mi
