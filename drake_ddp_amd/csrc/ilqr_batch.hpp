// Lane-per-problem iLQR kernel ("throughput mode") for small state dimension, gfx950.
//
// The wave-per-problem kernels (ilqr_small.hpp) minimize the latency of ONE problem and keep
// its state in LDS: the right shape up to ~2048 problems per GPU (LDS admits 4-6 per CU).
// For tens of thousands of problems the batch axis itself fills the machine: here every LANE
// owns one problem and the solver state streams through HBM in a BATCH-MINOR layout
//     x_bar[t][i][b], u_bar[t][k][b], K[t][k][j][b], kappa[t][k][b], dV[t][b],
//     fx[t][i][j][b], fu[t][i][k][b]            (b fastest)
// so a wavefront's 64 lanes touch 64 consecutive doubles: every access of the rollout
// (ilqr.py:306-327) and of the backward pass (:623-667) is one fully coalesced 512-byte
// transaction, prefetched 1-2 time steps ahead (addresses never depend on the state).  The
// kernel moves ~the algorithmic bytes of SURVEY.md §8d per iteration (no re-reads), i.e. at
// large B it is HBM-bound by construction, which is what the roofline accounting assumes.
//
// Control flow is per lane: line-search trials are sequential like the reference's
// (:300-337), lanes that have accepted / converged are masked while the wave finishes the
// slowest of its 64 problems.  x_bar/u_bar are double-buffered with a per-lane parity, so
// accepting a trial is a flip, not a copy.  Linearization (central FD, :233-272 replaced)
// is fused into the backward sweep when every step is a key-point ('setInterval' with minN = 1: KP = false).
// The other key-point configurations (ilqr.py:417-621) run in the KP = true instantiation: every lane builds ITS OWN
// key-point list (batch-minor integer scratch in HBM), evaluates the Jacobians at its key-points, interpolates in a
// lock-step pass over time (coalesced stores; a lane fetches a segment's end matrix when it crosses a key-point) and the
// backward sweep reads fx / fu back from HBM.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mi_ilqr.h"
#include "fastmath.hpp"
#include "ilqr_small.hpp"   // KArgs, Consts, stage_cost, terminal_cost, invert_small
#include "models.hpp"

namespace mi {

// batch-minor addressing helpers: element (t, r) of an array with `rows` rows per time step
__device__ __forceinline__ size_t bm(int t, int r, int rows, int B) { return ((size_t)t * rows + r) * B; }

template <class M, int JAC, bool KP = false>
__global__ void __launch_bounds__(64) ilqr_batch_kernel(const KArgs a) {
  constexpr int n = M::n, m = M::m, nc = n + m;
  const int b = blockIdx.x * 64 + threadIdx.x;
  const int B = a.B, N = a.N;
  const bool live = b < B;
  const int bb = live ? b : B - 1;             // out-of-range lanes shadow the last problem, stores parked
  // x_bar / u_bar live in two buffers: a.x_bar|a.x_trial and a.u_bar|a.u_trial; `cur` selects
  double* const X0 = a.x_bar + bb;
  double* const X1 = a.x_trial + bb;
  double* const U0 = a.u_bar + bb;
  double* const U1 = a.u_trial + bb;
  double* Kp = a.K + bb;
  double* kapp = a.kappa + bb;
  double* dVp = a.dV + bb;
  double* Fxp = a.fx + bb;
  double* Fup = a.fu + bb;
  // per-lane sink for masked stores: the last column of a scratch row (a.trial_cost, (B,2))
  double* sink = a.trial_cost + 2 * (size_t)bb;

  Consts<M> c;
  c.load(a.costmat);
  double x0r[n];
#pragma unroll
  for (int i = 0; i < n; ++i) x0r[i] = a.x0[(size_t)bb * n + i];
  double Q2[n][n], R2[m][m];
#pragma unroll
  for (int i = 0; i < n; ++i)
#pragma unroll
    for (int j = 0; j < n; ++j) Q2[i][j] = 2.0 * c.Q[i][j];
#pragma unroll
  for (int i = 0; i < m; ++i)
#pragma unroll
    for (int j = 0; j < m; ++j) R2[i][j] = 2.0 * c.R[i][j];

  // cold start / pending initial guess
  if (a.cold) {
    for (int t = 0; t < N; ++t) {
#pragma unroll
      for (int i = 0; i < n; ++i) X0[bm(t, i, n, B)] = 0.0;
    }
    for (int t = 0; t < N - 1; ++t) {
#pragma unroll
      for (int k = 0; k < m; ++k) {
        kapp[bm(t, k, m, B)] = 0.0;
#pragma unroll
        for (int j = 0; j < n; ++j) Kp[bm(t, k * n + j, m * n, B)] = 0.0;
      }
      dVp[(size_t)t * B] = 0.0;
    }
  }
  if (a.u_pending) {
    const double* ug = a.u_guess + bb;
    for (int t = 0; t < N - 1; ++t) {
#pragma unroll
      for (int k = 0; k < m; ++k) U0[bm(t, k, m, B)] = ug[bm(t, k, m, B)];
    }
  }

  int cur = 0;                                  // which buffer holds x_bar/u_bar for this lane
  double L = __builtin_inf(), improvement = __builtin_inf();
  int iters = 0, ls_total = 0, status = MI_STATUS_CONVERGED;
  int nk_lane = N - 1;                          // key-points of the lane's last linearization
  bool active = live;
  double* hist = a.hist + (size_t)bb * a.hist_cap * 4;

  while (__any(active && improvement > a.delta)) {
    const bool it_active = active && improvement > a.delta;
    if (it_active && iters >= a.max_iters) { status = MI_STATUS_MAX_ITERS; active = false; }
    const bool go = it_active && active;
    // ---------------- line search: sequential trials per lane (ilqr.py:300-337)
    double eps = 1.0, L_new = 0.0, eps_acc = 1.0;
    int trials = 0;
    bool accepted = !go;
    while (__any(!accepted && eps >= 1e-8)) {
      const bool run = !accepted && eps >= 1e-8;
      const double* xb = cur ? X1 : X0;
      const double* ub = cur ? U1 : U0;
      double* xw = run ? (cur ? X0 : X1) : sink; // masked lanes park their stores
      double* uw = run ? (cur ? U0 : U1) : sink;
      const size_t wstep_x = run ? (size_t)B : 0, wstep_u = run ? (size_t)B : 0;
      double x[n];
#pragma unroll
      for (int i = 0; i < n; ++i) { x[i] = x0r[i]; xw[(size_t)i * wstep_x] = x[i]; }
      double Lc = 0.0, ex = 0.0;
      const double ce = -eps * (1.0 - eps / 2.0);
      struct Regs { double xbv[n], ubv[m], kv[m], Kv[m][n], dv; };
      auto load = [&](Regs& r, int t) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < n; ++i) r.xbv[i] = xb[bm(t, i, n, B)];
#pragma unroll
        for (int k = 0; k < m; ++k) {
          r.ubv[k] = ub[bm(t, k, m, B)];
          r.kv[k] = kapp[bm(t, k, m, B)];
#pragma unroll
          for (int j = 0; j < n; ++j) r.Kv[k][j] = Kp[bm(t, k * n + j, m * n, B)];
        }
        r.dv = dVp[(size_t)t * B];
      };
      auto step = [&](const Regs& r, int t) __attribute__((always_inline)) {
        double u[m];
#pragma unroll
        for (int k = 0; k < m; ++k) {
          double acc = 0.0;
#pragma unroll
          for (int j = 0; j < n; ++j) acc += r.Kv[k][j] * (x[j] - r.xbv[j]);
          u[k] = (r.ubv[k] - eps * r.kv[k]) - acc;                     // :313
        }
        double xn[n];
        M::template step<double>(x, u, xn, a.params, a.dt);             // :316
        Lc += stage_cost<M>(c, x, u);                                   // :325
        ex += ce * r.dv;                                                // :326
#pragma unroll
        for (int k = 0; k < m; ++k) uw[((size_t)t * m + k) * wstep_u] = u[k];
#pragma unroll
        for (int i = 0; i < n; ++i) { xw[((size_t)(t + 1) * n + i) * wstep_x] = xn[i]; x[i] = xn[i]; }
      };
      Regs A, Bq;
      load(A, 0);
      int t = 0;
      const int tlast = N - 2;                   // last valid control index
      for (; t + 1 < N - 1; t += 2) {
        load(Bq, t + 1);
        __builtin_amdgcn_sched_barrier(0);
        step(A, t);
        load(A, (t + 2 <= tlast) ? t + 2 : tlast);
        __builtin_amdgcn_sched_barrier(0);
        step(Bq, t + 1);
      }
      if (t < N - 1) step(A, t);
      Lc += terminal_cost<M>(c, x);                                     // :327
      if (run) {
        trials += 1;
        if ((L - Lc) > a.gamma * ex) { accepted = true; L_new = Lc; eps_acc = eps; }   // :330-331
        else eps *= a.beta;                                             // :335
      }
    }
    bool ok = go && accepted;
    if (go) {
      ls_total += trials;
      if (!ok) { status = MI_STATUS_LINESEARCH_FAILED; active = false; }
    }
    if (ok) cur ^= 1;                            // u_bar <- u, x_bar <- x (:375-376): a flip
    // ---------------- linearization fused into the backward sweep (:380-415 with setInterval/1, :623-667)
    if (__any(ok)) {
      const double* xb = cur ? X1 : X0;
      const double* ub = cur ? U1 : U0;
      double* Kw = ok ? Kp : sink;
      double* kw = ok ? kapp : sink;
      double* dw = ok ? dVp : sink;
      double* fxw = ok ? Fxp : sink;
      double* fuw = ok ? Fup : sink;
      const size_t ws = ok ? (size_t)B : 0;
      double Vx[n], Vxx[n][n];
      {
        double xT[n];
#pragma unroll
        for (int i = 0; i < n; ++i) xT[i] = xb[bm(N - 1, i, n, B)];
#pragma unroll
        for (int i = 0; i < n; ++i) {
          double s = -c.qfn[i];
#pragma unroll
          for (int j = 0; j < n; ++j) { s += (2.0 * c.Qf[i][j]) * xT[j]; Vxx[i][j] = 2.0 * c.Qf[i][j]; }
          Vx[i] = s;                                                   // :203-204
        }
      }
      const double h = a.fd_h, inv2h = 1.0 / (2.0 * h);
      // one column of [fx | fu] after the other at (xv, uv): central differences or forward-mode duals (ilqr.py:233-272)
      auto jac_columns = [&](const double (&xv)[n], const double (&uv)[m], auto emit) __attribute__((always_inline)) {
#pragma unroll
        for (int col = 0; col < nc; ++col) {
          double d[n];
          if (JAC == MI_JAC_FD_CENTRAL) {
            double xp[n], up[m], fp[n], fmv[n];
#pragma unroll
            for (int i = 0; i < n; ++i) xp[i] = (col == i) ? xv[i] + h : xv[i];
#pragma unroll
            for (int k = 0; k < m; ++k) up[k] = (col == n + k) ? uv[k] + h : uv[k];
            M::template step<double>(xp, up, fp, a.params, a.dt);
#pragma unroll
            for (int i = 0; i < n; ++i) xp[i] = (col == i) ? xv[i] - h : xv[i];
#pragma unroll
            for (int k = 0; k < m; ++k) up[k] = (col == n + k) ? uv[k] - h : uv[k];
            M::template step<double>(xp, up, fmv, a.params, a.dt);
#pragma unroll
            for (int i = 0; i < n; ++i) d[i] = (fp[i] - fmv[i]) * inv2h;
          } else {
            Dual1 xd[n], ud[m], fd[n];
#pragma unroll
            for (int i = 0; i < n; ++i) xd[i] = Dual1(xv[i], (col == i) ? 1.0 : 0.0);
#pragma unroll
            for (int k = 0; k < m; ++k) ud[k] = Dual1(uv[k], (col == n + k) ? 1.0 : 0.0);
            M::template step<Dual1>(xd, ud, fd, a.params, a.dt);
#pragma unroll
            for (int i = 0; i < n; ++i) d[i] = fd[i].d;
          }
          emit(col, d);
        }
      };
      if constexpr (KP) {
        // ---- _get_derivatives (ilqr.py:380-415) per LANE: key-point list -> kpl[i] (batch-minor ints), Jacobians at the
        //      key-points -> fx / fu, then interpolation.  Lanes walk in lock step and mask what is not theirs.
        int* const kpl = a.bm_scratch + bb;                          // [i][b], i < N-1: the lane's key-points, ascending
        int* const done = a.bm_scratch + (size_t)(N - 1) * B + bb;     // [t][b]: derivative evaluated at t (iterativeError)
        int* binA = a.bm_scratch + (size_t)2 * (N - 1) * B + bb;       // [2 i + {0,1}][b]: the level's bins (s, e)
        int* binB = a.bm_scratch + (size_t)4 * (N - 1) * B + bb;
        const size_t Bz = (size_t)B;
        // Jacobians at the lane's own time step tt (act: this lane takes part) -> fx / fu
        auto eval_store = [&](int tt, bool act) __attribute__((always_inline)) {
          if (!__any(act)) return;
          const int tq = act ? tt : 0;
          double xv[n], uv[m];
#pragma unroll
          for (int i = 0; i < n; ++i) xv[i] = xb[bm(tq, i, n, B)];
#pragma unroll
          for (int k = 0; k < m; ++k) uv[k] = ub[bm(tq, k, m, B)];
          double* fo = act ? Fxp : sink;
          double* go = act ? Fup : sink;
          const size_t wz = act ? Bz : 0;
          jac_columns(xv, uv, [&](int col, const double (&d)[n]) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < n; ++i) {
              if (col < n) fo[((size_t)tq * n * n + i * n + col) * wz] = d[i];
              else go[((size_t)tq * n * m + i * m + (col - n)) * wz] = d[i];
            }
          });
        };
        // Key-point lists that are known BEFORE any Jacobian is needed (setInterval, adaptiveJerk): ONE forward pass over time
        // evaluates and interpolates (round 6).  Every lane walks the same t; where t is the NEXT key-point of any lane of the wave
        // the wave reads row t of x_bar / u_bar - a coalesced row, the same for all lanes - and differentiates the step; a lane
        // whose key-point it is keeps the result (fE), stores it, and fills the interior of the segment it closes, rows t-1 .. s+1 -
        // again the same rows for every lane that takes part - from fS (its previous key-point, still in registers) and fE: the
        // expression of ilqr.py:607-621, the same bits.  Until round 6 the lanes GATHERED x, u at their own key-points (a row per
        // distinct time step among the 64 lanes), scattered the Jacobians, and a second pass gathered both end matrices of every
        // segment back: 1.23 - 1.29 x the algorithmic bytes and twice the reads (profiles/r05_pmc_throughput.json); now no
        // Jacobian is read back before the backward sweep and nothing is gathered but the lane's own list (4 bytes a key-point).
        auto eval_and_interpolate = [&](int nk_) __attribute__((always_inline)) {
          constexpr int cnt = n * n + n * m;
          double fS[cnt];
#pragma unroll
          for (int r = 0; r < cnt; ++r) fS[r] = 0.0;
          int ptr = 0, s_ = 0;
          int next = (ok && nk_ > 0) ? kpl[0] : -1;
          for (int t = 0; t < N - 1; ++t) {
            const bool act = ok && t == next;
            if (!__any(act)) continue;
            double xv[n], uv[m], fE[cnt];
#pragma unroll
            for (int i = 0; i < n; ++i) xv[i] = xb[bm(t, i, n, B)];
#pragma unroll
            for (int k = 0; k < m; ++k) uv[k] = ub[bm(t, k, m, B)];
            jac_columns(xv, uv, [&](int col, const double (&d)[n]) __attribute__((always_inline)) {
#pragma unroll
              for (int i = 0; i < n; ++i) {
                if (col < n) fE[i * n + col] = d[i];
                else fE[n * n + i * m + (col - n)] = d[i];
              }
            });
            {
              double* fo = act ? Fxp : sink;
              double* go = act ? Fup : sink;
              const size_t wz = act ? Bz : 0;
#pragma unroll
              for (int r = 0; r < n * n; ++r) fo[((size_t)t * n * n + r) * wz] = fE[r];
#pragma unroll
              for (int r = 0; r < n * m; ++r) go[((size_t)t * n * m + r) * wz] = fE[n * n + r];
            }
            const int gap = (act && ptr > 0) ? t - s_ - 1 : 0;                  // interior points of the segment (s_, t)
            int gap_max = gap;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const int w_ = __shfl_xor(gap_max, o); gap_max = w_ > gap_max ? w_ : gap_max; }
            const double len = (double)(t - s_);
            for (int w = 1; w <= gap_max; ++w) {
              const int row = t - w;
              const bool in = w <= gap;
              const double w_ = (double)(row - s_);
              double* fo = in ? Fxp : sink;
              double* go = in ? Fup : sink;
              const size_t wz = in ? Bz : 0;
#pragma unroll
              for (int r = 0; r < n * n; ++r) fo[((size_t)row * n * n + r) * wz] = fS[r] + (fE[r] - fS[r]) * w_ / len;
#pragma unroll
              for (int r = 0; r < n * m; ++r) go[((size_t)row * n * m + r) * wz] = fS[n * n + r] + (fE[n * n + r] - fS[n * n + r]) * w_ / len;
            }
            if (act) {
#pragma unroll
              for (int r = 0; r < cnt; ++r) fS[r] = fE[r];
              s_ = t; ptr += 1;
              next = ptr < nk_ ? kpl[(size_t)ptr * Bz] : -1;
            }
          }
        };
        int nk = 0;
        bool fused_interp = false;
        if (a.kp_method == MI_KP_SET_INTERVAL) {
          // ilqr.py:417-432: arange(0, N-1, minN), the LAST entry overwritten with N-2 - the same list in every lane
          const int count = (N - 2) / a.minN + 1;
          for (int i = 0; i < count; ++i) {
            int tv = i * a.minN;
            if (i == count - 1 && tv != N - 2) tv = N - 2;
            if (ok) kpl[(size_t)i * Bz] = tv;
          }
          nk = count;
          eval_and_interpolate(nk);
          fused_interp = true;
        } else if (a.kp_method == MI_KP_ADAPTIVE_JERK) {
          // ilqr.py:434-486: signed second difference of the "velocity rows" x[dof + i], dof = int(n / 2); a key-point when the
          // counter has reached minN and a jerk exceeds the threshold, or when it reaches maxN; the last one overwritten with N-2
          constexpr int dof = n / 2;
          int since = 0, last = 0;
          if (ok) kpl[0] = 0;
          nk = 1;
          double v0[dof > 0 ? dof : 1], v1[dof > 0 ? dof : 1];
#pragma unroll
          for (int i = 0; i < dof; ++i) { v0[i] = xb[bm(0, i + dof, n, B)]; v1[i] = xb[bm(N > 1 ? 1 : 0, i + dof, n, B)]; }
          for (int t = 0; t < N - 3; ++t) {
            bool trig = false;
#pragma unroll
            for (int i = 0; i < dof; ++i) {
              const double v2 = xb[bm(t + 2, i + dof, n, B)];
              const double jerk = (v2 - v1[i]) - (v1[i] - v0[i]);
              trig = trig || (jerk > a.jerk_thr);
              v0[i] = v1[i]; v1[i] = v2;
            }
            since += 1;
            if (since >= a.minN && trig) { if (ok) kpl[(size_t)nk * Bz] = t; last = t; nk += 1; since = 0; }
            if (since >= a.maxN) { if (ok) kpl[(size_t)nk * Bz] = t; last = t; nk += 1; since = 0; }
          }
          if (last != N - 2 && ok) kpl[(size_t)(nk - 1) * Bz] = N - 2;                       // :465-466
          eval_and_interpolate(nk);                                                          // (lanes differ in their lists: see there)
          fused_interp = true;
        } else {
          // ilqr.py:488-593: level-synchronous bisection of [0, N-2]; a bin wider than minN is tested at its midpoint against the
          // mean of its end matrices (fx only, divisor 2n) and split when the error exceeds the threshold; every index the test
          // touches gets its exact Jacobians (memoized); the key-points are the indices that have them
          for (int t = 0; t < N - 1; ++t) { if (ok) done[(size_t)t * Bz] = 0; }
          int nb = 1;
          if (ok) { binA[0] = 0; binA[Bz] = N - 2; }
          bool level_active = ok;
          while (__any(level_active)) {
            int nb_max = level_active ? nb : 0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const int w_ = __shfl_xor(nb_max, o); nb_max = w_ > nb_max ? w_ : nb_max; }
            int nn = 0;
            for (int i = 0; i < nb_max; ++i) {
              const bool have = level_active && i < nb;
              const int s_ = have ? binA[(size_t)(2 * i) * Bz] : 0, e_ = have ? binA[(size_t)(2 * i + 1) * Bz] : 0;
              const bool big = have && (e_ - s_ > a.minN);
              const int mid = (s_ + e_) / 2;
              for (int q = 0; q < 3; ++q) {
                const int idx = q == 0 ? s_ : (q == 1 ? mid : e_);
                const bool need = big && done[(size_t)idx * Bz] == 0;
                eval_store(idx, need);
                if (need) done[(size_t)idx * Bz] = 1;
              }
              bool bad = false;
              if (__any(big)) {
                double sum = 0.0;
#pragma unroll
                for (int r = 0; r < n * n; ++r) {
                  const double fe = Fxp[bm(e_, r, n * n, B)], fs = Fxp[bm(s_, r, n * n, B)], fmid = Fxp[bm(mid, r, n * n, B)];
                  const double lin = (fe + fs) / 2.0;
                  const double df = lin - fmid;
                  sum += df * df;
                }
                bad = big && (sum / (2.0 * n)) > a.err_thr;
              }
              if (bad) {
                binB[(size_t)(4 * nn) * Bz] = s_; binB[(size_t)(4 * nn + 1) * Bz] = mid;
                binB[(size_t)(4 * nn + 2) * Bz] = mid; binB[(size_t)(4 * nn + 3) * Bz] = e_;
                nn += 1;
              }
            }
            if (level_active) {
              if (nn == 0) level_active = false;
              else { nb = 2 * nn; int* tmp = binA; binA = binB; binB = tmp; }
            }
          }
          for (int t = 0; t < N - 1; ++t) {
            if (ok && done[(size_t)t * Bz] != 0) { kpl[(size_t)nk * Bz] = t; nk += 1; }
          }
        }
        // ---- interpolate_derivatives (ilqr.py:596-621): lock step over time, interior points only; a lane fetches its segment's
        //      end matrices when it crosses a key-point
        //      (iterativeError only: its list is known only once the Jacobians it tests have been evaluated; the other two methods
        //       interpolate in the pass that evaluates, above)
        if (!fused_interp) {
          constexpr int cnt = n * n + n * m;
          double fS[cnt], fE[cnt];
          int seg = 0, s_ = 0, e_ = 0;
          auto fetch = [&](double (&f)[cnt], int tt, bool act) __attribute__((always_inline)) {
            const int tq = act ? tt : 0;
#pragma unroll
            for (int r = 0; r < n * n; ++r) f[r] = Fxp[bm(tq, r, n * n, B)];
#pragma unroll
            for (int r = 0; r < n * m; ++r) f[n * n + r] = Fup[bm(tq, r, n * m, B)];
          };
          const bool have_seg = ok && nk >= 2;
          s_ = have_seg ? kpl[0] : 0;
          e_ = have_seg ? kpl[Bz] : 0;
          fetch(fS, s_, have_seg);
          fetch(fE, e_, have_seg);
          for (int t = 0; t < N - 1; ++t) {
            bool adv = have_seg && t == e_ && seg + 2 < nk;                  // crossing into the next segment
            if (__any(adv)) {
              const int e2 = adv ? kpl[(size_t)(seg + 2) * Bz] : 0;
              double fN[cnt];
              fetch(fN, e2, adv);
              if (adv) {
#pragma unroll
                for (int r = 0; r < cnt; ++r) { fS[r] = fE[r]; fE[r] = fN[r]; }
                s_ = e_; e_ = e2; seg += 1;
              }
            }
            const bool inner = have_seg && t > s_ && t < e_;
            if (__any(inner)) {
              const double len = (double)(e_ - s_), w_ = (double)(t - s_);
              double* fo = inner ? Fxp : sink;
              double* go = inner ? Fup : sink;
              const size_t wz = inner ? Bz : 0;
#pragma unroll
              for (int r = 0; r < n * n; ++r) fo[((size_t)t * n * n + r) * wz] = fS[r] + (fE[r] - fS[r]) * w_ / len;
#pragma unroll
              for (int r = 0; r < n * m; ++r) go[((size_t)t * n * m + r) * wz] = fS[n * n + r] + (fE[n * n + r] - fS[n * n + r]) * w_ / len;
            }
          }
        }
        if (ok) { nk_lane = nk; }
      }
      struct XU { double x[n], u[m]; };
      auto loadxu = [&](XU& r, int t) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < n; ++i) r.x[i] = xb[bm(t, i, n, B)];
#pragma unroll
        for (int k = 0; k < m; ++k) r.u[k] = ub[bm(t, k, m, B)];
      };
      XU cu, nx;
      loadxu(cu, N - 2);
      for (int t = N - 2; t >= 0; --t) {
        loadxu(nx, t > 0 ? t - 1 : 0);
        // dynamics partials at (x_bar_t, u_bar_t): evaluated here (every step a key-point), or read back (key-point variant)
        double fx[n][n], fu[n][m];
        if constexpr (KP) {
#pragma unroll
          for (int i = 0; i < n; ++i) {
#pragma unroll
            for (int j = 0; j < n; ++j) fx[i][j] = Fxp[bm(t, i * n + j, n * n, B)];
#pragma unroll
            for (int k = 0; k < m; ++k) fu[i][k] = Fup[bm(t, i * m + k, n * m, B)];
          }
        } else {
          jac_columns(cu.x, cu.u, [&](int col, const double (&d)[n]) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < n; ++i) {
              if (col < n) { fx[i][col] = d[i]; fxw[((size_t)t * n * n + i * n + col) * ws] = d[i]; }
              else { fu[i][col - n] = d[i]; fuw[((size_t)t * n * m + i * m + (col - n)) * ws] = d[i]; }
            }
          });
        }
        // cost expansion + Riccati step (same arithmetic as backward_scalar)
        double Qx[n], Qu[m], Qxx[n][n], Quu[m][m], Qux[m][n], Am[n][n], Bm[m][n];
#pragma unroll
        for (int i = 0; i < n; ++i) {
          double s = -c.qn[i];
#pragma unroll
          for (int j = 0; j < n; ++j) s += Q2[i][j] * cu.x[j];
#pragma unroll
          for (int k = 0; k < n; ++k) s += fx[k][i] * Vx[k];
          Qx[i] = s;
        }
#pragma unroll
        for (int a_ = 0; a_ < m; ++a_) {
          double s = 0.0;
#pragma unroll
          for (int j = 0; j < m; ++j) s += R2[a_][j] * cu.u[j];
#pragma unroll
          for (int k = 0; k < n; ++k) s += fu[k][a_] * Vx[k];
          Qu[a_] = s;
        }
#pragma unroll
        for (int i = 0; i < n; ++i)
#pragma unroll
          for (int j = 0; j < n; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < n; ++k) s += fx[k][i] * Vxx[k][j];
            Am[i][j] = s;
          }
#pragma unroll
        for (int a_ = 0; a_ < m; ++a_)
#pragma unroll
          for (int j = 0; j < n; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < n; ++k) s += fu[k][a_] * Vxx[k][j];
            Bm[a_][j] = s;
          }
#pragma unroll
        for (int i = 0; i < n; ++i)
#pragma unroll
          for (int j = 0; j < n; ++j) {
            double s = Q2[i][j];
#pragma unroll
            for (int k = 0; k < n; ++k) s += Am[i][k] * fx[k][j];
            Qxx[i][j] = s;
          }
#pragma unroll
        for (int a_ = 0; a_ < m; ++a_) {
#pragma unroll
          for (int b_ = 0; b_ < m; ++b_) {
            double s = R2[a_][b_];
#pragma unroll
            for (int k = 0; k < n; ++k) s += Bm[a_][k] * fu[k][b_];
            Quu[a_][b_] = s;
          }
#pragma unroll
          for (int j = 0; j < n; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < n; ++k) s += Bm[a_][k] * fx[k][j];
            Qux[a_][j] = s;
          }
        }
        double Qi[m][m];
        invert_small<m>(Quu, Qi);
        double kap[m], Kg[m][n], QuQi[m];
#pragma unroll
        for (int a_ = 0; a_ < m; ++a_) {
          double s = 0.0, q = 0.0;
#pragma unroll
          for (int b_ = 0; b_ < m; ++b_) { s += Qi[a_][b_] * Qu[b_]; q += Qu[b_] * Qi[b_][a_]; }
          kap[a_] = s; QuQi[a_] = q;
#pragma unroll
          for (int j = 0; j < n; ++j) {
            double g = 0.0;
#pragma unroll
            for (int b_ = 0; b_ < m; ++b_) g += Qi[a_][b_] * Qux[b_][j];
            Kg[a_][j] = g;
          }
        }
        double dv = 0.0;
#pragma unroll
        for (int a_ = 0; a_ < m; ++a_) dv += QuQi[a_] * Qu[a_];
#pragma unroll
        for (int a_ = 0; a_ < m; ++a_) {
          kw[((size_t)t * m + a_) * ws] = kap[a_];
#pragma unroll
          for (int j = 0; j < n; ++j) Kw[((size_t)t * m * n + a_ * n + j) * ws] = Kg[a_][j];
        }
        dw[(size_t)t * ws] = dv;
#pragma unroll
        for (int j = 0; j < n; ++j) {
          double s = Qx[j];
#pragma unroll
          for (int a_ = 0; a_ < m; ++a_) s -= QuQi[a_] * Qux[a_][j];
          Vx[j] = s;
        }
        double QuxTQi[n][m];
#pragma unroll
        for (int i = 0; i < n; ++i)
#pragma unroll
          for (int b_ = 0; b_ < m; ++b_) {
            double s = 0.0;
#pragma unroll
            for (int a_ = 0; a_ < m; ++a_) s += Qux[a_][i] * Qi[a_][b_];
            QuxTQi[i][b_] = s;
          }
#pragma unroll
        for (int i = 0; i < n; ++i)
#pragma unroll
          for (int j = 0; j < n; ++j) {
            double s = Qxx[i][j];
#pragma unroll
            for (int b_ = 0; b_ < m; ++b_) s -= QuxTQi[i][b_] * Qux[b_][j];
            Vxx[i][j] = s;
          }
        cu = nx;
      }
    }
    if (ok) {
      if (iters < a.hist_cap) {
        hist[4 * iters + 0] = L_new; hist[4 * iters + 1] = eps_acc;
        hist[4 * iters + 2] = (double)trials; hist[4 * iters + 3] = KP ? (double)nk_lane / (double)(N - 1) * 100.0 : 100.0;
      }
      improvement = L - L_new;                                          // :706
      L = L_new;
      iters += 1;
    }
  }
  // make buffer 0 the canonical x_bar/u_bar
  if (live && cur == 1) {
    for (int t = 0; t < N; ++t) {
#pragma unroll
      for (int i = 0; i < n; ++i) X0[bm(t, i, n, B)] = X1[bm(t, i, n, B)];
    }
    for (int t = 0; t < N - 1; ++t) {
#pragma unroll
      for (int k = 0; k < m; ++k) U0[bm(t, k, m, B)] = U1[bm(t, k, m, B)];
    }
  }
  if (live) {
    a.cost[b] = L; a.iters[b] = iters; a.status[b] = status; a.ls_trials[b] = ls_total; a.kp_count[b] = KP ? nk_lane : N - 1;
    if constexpr (KP) {
      if (iters > 0) { for (int i = 0; i < nk_lane; ++i) a.kp_list[(size_t)b * (N - 1) + i] = a.bm_scratch[(size_t)i * B + b]; }
    }
  }
}

}  // namespace mi
