// Wave-per-problem iLQR kernels for small state dimension (n <= 4..6), gfx950.
//
// One 64-lane wavefront (= one workgroup) owns one problem of the batch.  The
// whole per-problem solver state of the reference (x_bar,u_bar,K,kappa,fx,fu,
// dV_coeff — /root/reference/ilqr.py:70-83) is staged ONCE from HBM into LDS,
// the entire Solve() loop (ilqr.py:692-708) runs out of LDS/registers, and the
// results are written back ONCE.  Lanes are used for what is parallel in the
// algorithm:
//   * line search (ilqr.py:300-337): lane j rolls out candidate eps = beta^(base+j)
//     concurrently; the first accepted candidate in lane order is exactly the one
//     the reference's sequential loop accepts.  Lane 0 also stores
//     its trajectory, so the common case (eps=1 accepted) costs ONE rollout.
//   * linearization (ilqr.py:380-415, 233-272): (key-point, column) pairs are
//     spread over the lanes; central finite differences (or one-directional
//     forward-mode duals) replace Drake AutoDiff.
//   * key-point selection / interpolation (ilqr.py:417-621): ballots + per-lane
//     segments.
//   * backward Riccati pass (ilqr.py:623-667 with :161-206 fused in): strictly
//     sequential in t, evaluated wave-uniformly out of registers.
// Time is the fastest LDS axis (the reference's own layout), so
// HBM<->LDS staging is a linear, fully coalesced copy per array.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mi_ilqr.h"
#include "models.hpp"

namespace mi {

enum KernelMode { MODE_SOLVE = 0, MODE_ROLLOUT = 1, MODE_FORWARD = 2, MODE_LINEARIZE = 3, MODE_BACKWARD = 4 };

struct KArgs {
  // persistent per-problem solver state, reference layout with a leading batch axis
  double *x_bar, *u_bar, *K, *kappa, *dV, *fx, *fu;
  const double* x0;        // (B,n)
  const double* u_guess;   // (B,m,N-1) pending SetInitialGuess input (used when u_pending)
  double* cost;            // (B,)
  double* hist;            // (B,hist_cap,4)
  double *x_trial, *u_trial, *trial_cost;   // stage outputs
  const double* stage_in;  // (B,) eps (ROLLOUT) or L_last (FORWARD)
  const double* costmat;   // Q[n*n] R[m*m] Qf[n*n] x_nom[n]
  int32_t *iters, *status, *ls_trials, *kp_count, *kp_list;
  double params[MI_ILQR_MAX_PARAMS];
  double dt, delta, beta, gamma, jerk_thr, err_thr, fd_h;
  int32_t N, B, kp_method, minN, maxN, max_iters, hist_cap;
  int32_t cold;       // 1: persistent state is all-zero, do not read it
  int32_t u_pending;  // 1: take u_bar from u_guess
};

__device__ __forceinline__ double bcast_lane0(double v) {
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readfirstlane(u.i[0]);
  u.i[1] = __builtin_amdgcn_readfirstlane(u.i[1]);
  return u.d;
}

__device__ __forceinline__ void wave_sync() { __syncthreads(); }

// Per-problem workspace carved out of dynamic LDS.  Strides: SN = N for state
// trajectories, SM = N-1 for everything indexed by control step.
struct WS {
  double *xb, *xn, *ub, *un, *K, *kap, *fx, *fu, *dV;
  int *kp, *aux, *need, *binA, *binB;
  int N, SN, SM;
};

template <int n, int m>
__host__ __device__ constexpr size_t ws_doubles(int N) {
  return (size_t)2 * n * N + (size_t)(2 * m + m * n + m + n * n + n * m + 1) * (N - 1);
}
template <int n, int m>
__host__ __device__ constexpr size_t ws_bytes(int N) {
  // doubles + kp[N] + aux[N] + need[N] + binA[2N] + binB[2N] ints
  return ws_doubles<n, m>(N) * 8 + (size_t)7 * N * 4 + 16;
}

template <int n, int m>
__device__ inline WS carve(char* base, int N) {
  WS w;
  w.N = N; w.SN = N; w.SM = N - 1;
  double* p = reinterpret_cast<double*>(base);
  w.xb = p; p += n * N;
  w.xn = p; p += n * N;
  w.ub = p; p += m * (N - 1);
  w.un = p; p += m * (N - 1);
  w.K = p; p += m * n * (N - 1);
  w.kap = p; p += m * (N - 1);
  w.fx = p; p += n * n * (N - 1);
  w.fu = p; p += n * m * (N - 1);
  w.dV = p; p += (N - 1);
  int* q = reinterpret_cast<int*>(p);
  w.kp = q; q += N;
  w.aux = q; q += N;
  w.need = q; q += N;
  w.binA = q; q += 2 * N;
  w.binB = q;
  return w;
}

__device__ inline void copy_in(double* dst, const double* src, int count, bool zero) {
  if (zero) { for (int i = threadIdx.x; i < count; i += 64) dst[i] = 0.0; }
  else { for (int i = threadIdx.x; i < count; i += 64) dst[i] = src[i]; }
}
__device__ inline void copy_out(double* dst, const double* src, int count) {
  for (int i = threadIdx.x; i < count; i += 64) dst[i] = src[i];
}

template <class M>
struct Consts {
  static constexpr int n = M::n, m = M::m;
  double Q[n][n], R[m][m], Qf[n][n], xnom[n];
  double qn[n];    // 2*x_nom^T Q    (ilqr.py:180)
  double qfn[n];   // 2*x_nom^T Qf   (ilqr.py:203)
  __device__ inline void load(const double* cm) {
#pragma unroll
    for (int i = 0; i < n; ++i)
#pragma unroll
      for (int j = 0; j < n; ++j) { Q[i][j] = cm[i * n + j]; Qf[i][j] = cm[n * n + m * m + i * n + j]; }
#pragma unroll
    for (int i = 0; i < m; ++i)
#pragma unroll
      for (int j = 0; j < m; ++j) R[i][j] = cm[n * n + i * m + j];
#pragma unroll
    for (int i = 0; i < n; ++i) xnom[i] = cm[2 * n * n + m * m + i];
#pragma unroll
    for (int j = 0; j < n; ++j) {
      double s = 0.0, sf = 0.0;
#pragma unroll
      for (int i = 0; i < n; ++i) { s += (2.0 * xnom[i]) * Q[i][j]; sf += (2.0 * xnom[i]) * Qf[i][j]; }
      qn[j] = s; qfn[j] = sf;
    }
  }
};

// ---------------------------------------------------------------------------
// One line-search trial (ilqr.py:306-327) for this lane's eps.  Reads the
// nominal trajectory and gains from LDS (wave-uniform addresses -> broadcast
// reads, software-prefetched one step ahead so the LDS latency is off the x
// dependency chain); lane 0 optionally stores the trajectory into xn/un.
// ---------------------------------------------------------------------------
template <class M>
__device__ inline void rollout(const WS& w, const Consts<M>& c, const KArgs& a, const double* x0r,
                               double eps, bool store, double& L_out, double& exp_out) {
  constexpr int n = M::n, m = M::m;
  const int N = w.N, SN = w.SN, SM = w.SM;
  double x[n];
#pragma unroll
  for (int i = 0; i < n; ++i) x[i] = x0r[i];
  if (store) {
#pragma unroll
    for (int i = 0; i < n; ++i) w.xn[i * SN] = x[i];
  }
  double L = 0.0, expd = 0.0;
  const double ce = -eps * (1.0 - eps / 2.0);

  double ub[m], kp[m], Kt[m][n], xb[n], dv;
  // prefetch t = 0
#pragma unroll
  for (int k = 0; k < m; ++k) {
    ub[k] = w.ub[k * SM]; kp[k] = w.kap[k * SM];
#pragma unroll
    for (int j = 0; j < n; ++j) Kt[k][j] = w.K[(k * n + j) * SM];
  }
#pragma unroll
  for (int i = 0; i < n; ++i) xb[i] = w.xb[i * SN];
  dv = w.dV[0];

  for (int t = 0; t < N - 1; ++t) {
    double ub_c[m], kp_c[m], K_c[m][n], xb_c[n];
    const double dv_c = dv;
#pragma unroll
    for (int k = 0; k < m; ++k) {
      ub_c[k] = ub[k]; kp_c[k] = kp[k];
#pragma unroll
      for (int j = 0; j < n; ++j) K_c[k][j] = Kt[k][j];
    }
#pragma unroll
    for (int i = 0; i < n; ++i) xb_c[i] = xb[i];
    const int tn = (t + 1 < N - 1) ? t + 1 : t;   // clamp: last prefetch re-reads t
#pragma unroll
    for (int k = 0; k < m; ++k) {
      ub[k] = w.ub[k * SM + tn]; kp[k] = w.kap[k * SM + tn];
#pragma unroll
      for (int j = 0; j < n; ++j) Kt[k][j] = w.K[(k * n + j) * SM + tn];
    }
#pragma unroll
    for (int i = 0; i < n; ++i) xb[i] = w.xb[i * SN + tn];
    dv = w.dV[tn];

    // u_t = u_bar_t - eps*kappa_t - K_t (x_t - x_bar_t)          (ilqr.py:313)
    double u[m];
#pragma unroll
    for (int k = 0; k < m; ++k) {
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < n; ++j) acc += K_c[k][j] * (x[j] - xb_c[j]);
      u[k] = (ub_c[k] - eps * kp_c[k]) - acc;
    }
    double xnext[n];
    M::template step<double>(x, u, xnext, a.params, a.dt);        // ilqr.py:316

    // stage cost (no 1/2 factor, ilqr.py:325) and expected improvement (:326)
    double dx[n];
#pragma unroll
    for (int i = 0; i < n; ++i) dx[i] = x[i] - c.xnom[i];
    double q = 0.0;
#pragma unroll
    for (int i = 0; i < n; ++i) {
      double r = 0.0;
#pragma unroll
      for (int j = 0; j < n; ++j) r += c.Q[i][j] * dx[j];
      q += dx[i] * r;
    }
    double ru = 0.0;
#pragma unroll
    for (int i = 0; i < m; ++i) {
      double r = 0.0;
#pragma unroll
      for (int j = 0; j < m; ++j) r += c.R[i][j] * u[j];
      ru += u[i] * r;
    }
    L += q + ru;
    expd += ce * dv_c;

    if (store) {
#pragma unroll
      for (int k = 0; k < m; ++k) w.un[k * SM + t] = u[k];
#pragma unroll
      for (int i = 0; i < n; ++i) w.xn[i * SN + t + 1] = xnext[i];
    }
#pragma unroll
    for (int i = 0; i < n; ++i) x[i] = xnext[i];
  }
  // terminal cost (ilqr.py:327)
  {
    double dx[n];
#pragma unroll
    for (int i = 0; i < n; ++i) dx[i] = x[i] - c.xnom[i];
    double q = 0.0;
#pragma unroll
    for (int i = 0; i < n; ++i) {
      double r = 0.0;
#pragma unroll
      for (int j = 0; j < n; ++j) r += c.Qf[i][j] * dx[j];
      q += dx[i] * r;
    }
    L += q;
  }
  L_out = L;
  exp_out = expd;
}

// ---------------------------------------------------------------------------
// Speculative parallel line search (ilqr.py:300-337).  Returns true on accept;
// xn/un then hold the accepted trajectory.  `trials` is the reference-equivalent
// sequential trial count (accepted candidate index + 1).
// ---------------------------------------------------------------------------
template <class M>
__device__ inline bool linesearch(const WS& w, const Consts<M>& c, const KArgs& a, const double* x0r,
                                  double L_last, double& L_out, double& eps_out, int& trials) {
  const int lane = threadIdx.x;
  int base = 0;
  double eps_base = 1.0;
  for (;;) {
    double eps = eps_base;
    for (int i = 0; i < lane; ++i) eps *= a.beta;   // eps *= beta, repeated (ilqr.py:335): bit-identical sequence
    const bool valid = eps >= 1e-8;                 // while eps >= 1e-8 (ilqr.py:302)
    double L, ex;
    rollout<M>(w, c, a, x0r, eps, lane == 0, L, ex);
    const bool acc = valid && ((L_last - L) > a.gamma * ex);   // ilqr.py:330-331
    const unsigned long long mask = __ballot(acc);
    if (mask != 0ull) {
      const int k = __ffsll((long long)mask) - 1;
      if (k == 0) {
        L_out = bcast_lane0(L);
        eps_out = eps_base;
        trials = base + 1;
        return true;
      }
      // candidate base+k wins: re-run with it in lane 0 so its trajectory is stored
      for (int i = 0; i < k; ++i) eps_base *= a.beta;
      base += k;
      wave_sync();
      continue;
    }
    const unsigned long long vmask = __ballot(valid);
    if (vmask != ~0ull) {   // ran out of eps >= 1e-8 without acceptance (ilqr.py:337)
      trials = base + __popcll(vmask);
      return false;
    }
    for (int i = 0; i < 64; ++i) eps_base *= a.beta;
    base += 64;
    wave_sync();
  }
}

// ---------------------------------------------------------------------------
// Dynamics partials at the listed time steps (replaces _calc_dynamics_partials,
// ilqr.py:233-272): items (list entry, column) over lanes.
// ---------------------------------------------------------------------------
template <class M, int JAC>
__device__ inline void jac_at(const WS& w, const KArgs& a, const double* xs, const double* us,
                              const int* list, int count) {
  constexpr int n = M::n, m = M::m, nc = n + m;
  const int SN = w.SN, SM = w.SM;
  const double h = a.fd_h, inv2h = 1.0 / (2.0 * h);
  for (int it = threadIdx.x; it < count * nc; it += 64) {
    const int ki = it / nc, col = it - ki * nc;
    const int t = list[ki];
    double x[n], u[m], d[n];
#pragma unroll
    for (int i = 0; i < n; ++i) x[i] = xs[i * SN + t];
#pragma unroll
    for (int k = 0; k < m; ++k) u[k] = us[k * SM + t];
    if (JAC == MI_JAC_FD_CENTRAL) {
      double xp[n], up[m], xm[n], um[m], fp[n], fm[n];
#pragma unroll
      for (int i = 0; i < n; ++i) { xp[i] = (col == i) ? x[i] + h : x[i]; xm[i] = (col == i) ? x[i] - h : x[i]; }
#pragma unroll
      for (int k = 0; k < m; ++k) { up[k] = (col == n + k) ? u[k] + h : u[k]; um[k] = (col == n + k) ? u[k] - h : u[k]; }
      M::template step<double>(xp, up, fp, a.params, a.dt);
      M::template step<double>(xm, um, fm, a.params, a.dt);
#pragma unroll
      for (int i = 0; i < n; ++i) d[i] = (fp[i] - fm[i]) * inv2h;
    } else {
      Dual1 xd[n], ud[m], fd[n];
#pragma unroll
      for (int i = 0; i < n; ++i) xd[i] = Dual1(x[i], (col == i) ? 1.0 : 0.0);
#pragma unroll
      for (int k = 0; k < m; ++k) ud[k] = Dual1(u[k], (col == n + k) ? 1.0 : 0.0);
      M::template step<Dual1>(xd, ud, fd, a.params, a.dt);
#pragma unroll
      for (int i = 0; i < n; ++i) d[i] = fd[i].d;
    }
    if (col < n) {
#pragma unroll
      for (int i = 0; i < n; ++i) w.fx[(i * n + col) * SM + t] = d[i];
    } else {
#pragma unroll
      for (int i = 0; i < n; ++i) w.fu[(i * m + (col - n)) * SM + t] = d[i];
    }
  }
}

// Ordered stream compaction of {t in [0,count) : pred(t)} into list; returns size.
template <class Pred>
__device__ inline int compact(int count, int* list, Pred pred) {
  const int lane = threadIdx.x;
  int total = 0;
  for (int t0 = 0; t0 < count; t0 += 64) {
    const int t = t0 + lane;
    const bool p = (t < count) && pred(t);
    const unsigned long long mask = __ballot(p);
    const int pos = total + __popcll(mask & ((1ull << lane) - 1ull));
    if (p) list[pos] = t;
    total += __popcll(mask);
  }
  return total;
}

// get_keypoints_set_interval (ilqr.py:417-432)
__device__ inline int keypoints_set_interval(const WS& w, int minN) {
  const int N = w.N;
  const int count = (N - 2) / minN + 1;            // len(arange(0, N-1, minN))
  for (int i = threadIdx.x; i < count; i += 64) {
    int v = i * minN;
    if (i == count - 1 && v != N - 2) v = N - 2;   // overwrite, not append (:428-430)
    w.kp[i] = v;
  }
  return count;
}

// get_keypoints_adaptive_jerk + calc_jerk_profile (ilqr.py:434-486).  The jerk
// test is evaluated for 64 time steps at once; the counter automaton then walks
// the ballot mask with scalar code.
template <int n>
__device__ inline int keypoints_adaptive_jerk(const WS& w, const KArgs& a, const double* xs) {
  const int N = w.N, SN = w.SN, lane = threadIdx.x;
  constexpr int dof = n / 2;
  int nk = 0, since = 0, last = 0;
  if (lane == 0) w.kp[0] = 0;
  nk = 1;
  for (int t0 = 0; t0 < N - 3; t0 += 64) {
    const int t = t0 + lane;
    bool trig = false;
    if (t < N - 3) {
#pragma unroll
      for (int i = 0; i < dof; ++i) {
        const double* v = xs + (i + dof) * SN + t;
        const double jerk = (v[2] - v[1]) - (v[1] - v[0]);   // signed, no abs (:481-484)
        trig = trig || (jerk > a.jerk_thr);
      }
    }
    const unsigned long long mask = __ballot(trig);
    const int lim = (N - 3 - t0) < 64 ? (N - 3 - t0) : 64;
    for (int j = 0; j < lim; ++j) {
      since += 1;
      if (since >= a.minN && ((mask >> j) & 1ull)) {
        if (lane == 0) w.kp[nk] = t0 + j;
        last = t0 + j; nk += 1; since = 0;
      }
      if (since >= a.maxN) {
        if (lane == 0) w.kp[nk] = t0 + j;
        last = t0 + j; nk += 1; since = 0;
      }
    }
  }
  if (last != N - 2 && lane == 0) w.kp[nk - 1] = N - 2;      // :465-466
  return nk;
}

// get_keypoints_iterative_error + check_one_matrix_error (ilqr.py:488-593):
// level-synchronous bisection, one lane per bin; Jacobians are evaluated (and
// written into fx/fu) only where the reference would evaluate them.
template <class M, int JAC>
__device__ inline int keypoints_iterative_error(const WS& w, const KArgs& a, const double* xs, const double* us) {
  constexpr int n = M::n;
  const int N = w.N, SM = w.SM, lane = threadIdx.x;
  int* done = w.aux;             // 0/1 per time step: derivative evaluated (deriv_calculated_at_index)
  int* need = w.need;            // scratch flags: indices a level wants evaluated
  for (int t = lane; t < N; t += 64) done[t] = 0;
  int* bins = w.binA;            // (s,e) pairs
  int* next = w.binB;
  int nb = 1;
  if (lane == 0) { bins[0] = 0; bins[1] = N - 2; }
  wave_sync();
  // A level's bins are disjoint sub-intervals of [0,N-2] of width >= 1: at most N-1
  // pairs = 2(N-1) ints per buffer.
  for (;;) {
    for (int t = lane; t < N; t += 64) need[t] = 0;
    wave_sync();
    for (int i = lane; i < nb; i += 64) {
      const int s = bins[2 * i], e = bins[2 * i + 1];
      if (e - s > a.minN) { const int mid = (s + e) / 2; need[s] = 1; need[mid] = 1; need[e] = 1; }
    }
    wave_sync();
    const int cnt = compact(N, w.kp, [&](int t) { return need[t] && !done[t]; });
    wave_sync();
    jac_at<M, JAC>(w, a, xs, us, w.kp, cnt);
    for (int i = lane; i < cnt; i += 64) done[w.kp[i]] = 1;
    wave_sync();
    // evaluate bins; bad ones are split (order within a level is irrelevant to the result)
    int nn = 0;
    for (int i0 = 0; i0 < nb; i0 += 64) {
      const int i = i0 + lane;
      bool bad = false;
      int s = 0, e = 0, mid = 0;
      if (i < nb) {
        s = bins[2 * i]; e = bins[2 * i + 1]; mid = (s + e) / 2;
        if (e - s > a.minN) {
          double sum = 0.0;
          for (int r = 0; r < n * n; ++r) {
            const double lin = (w.fx[r * SM + e] + w.fx[r * SM + s]) / 2.0;
            const double df = lin - w.fx[r * SM + mid];
            sum += df * df;
          }
          bad = (sum / (2.0 * n)) > a.err_thr;       // divisor 2n, fx only (:583-591)
        }
      }
      const unsigned long long mask = __ballot(bad);
      const int pos = nn + __popcll(mask & ((1ull << lane) - 1ull));
      if (bad) { next[4 * pos] = s; next[4 * pos + 1] = mid; next[4 * pos + 2] = mid; next[4 * pos + 3] = e; }
      nn += __popcll(mask);
    }
    wave_sync();
    if (nn == 0) break;
    nb = 2 * nn;
    int* tmp = bins; bins = next; next = tmp;
  }
  const int nk = compact(N - 1, w.kp, [&](int t) { return done[t] != 0; });
  wave_sync();
  return nk;
}

// interpolate_derivatives (ilqr.py:596-621): one lane per key-point segment,
// interior points only (the end points are reproduced exactly by the formula).
template <int n, int m>
__device__ inline void interpolate(const WS& w, int nk) {
  const int SM = w.SM;
  for (int i = threadIdx.x; i < nk - 1; i += 64) {
    const int s = w.kp[i], e = w.kp[i + 1];
    if (e - s < 2) continue;
    const double len = (double)(e - s);
    for (int r = 0; r < n * n; ++r) {
      const double fs = w.fx[r * SM + s], fe = w.fx[r * SM + e];
      for (int j = s + 1; j < e; ++j) w.fx[r * SM + j] = fs + (fe - fs) * (double)(j - s) / len;
    }
    for (int r = 0; r < n * m; ++r) {
      const double fs = w.fu[r * SM + s], fe = w.fu[r * SM + e];
      for (int j = s + 1; j < e; ++j) w.fu[r * SM + j] = fs + (fe - fs) * (double)(j - s) / len;
    }
  }
}

// _get_derivatives (ilqr.py:380-415) at trajectory (xs,us).  Returns key-point count.
template <class M, int JAC>
__device__ inline int linearize(const WS& w, const KArgs& a, const double* xs, const double* us) {
  constexpr int n = M::n, m = M::m;
  int nk;
  if (a.kp_method == MI_KP_SET_INTERVAL) {
    nk = keypoints_set_interval(w, a.minN);
    wave_sync();
    jac_at<M, JAC>(w, a, xs, us, w.kp, nk);
  } else if (a.kp_method == MI_KP_ADAPTIVE_JERK) {
    nk = keypoints_adaptive_jerk<n>(w, a, xs);
    wave_sync();
    jac_at<M, JAC>(w, a, xs, us, w.kp, nk);
  } else {
    nk = keypoints_iterative_error<M, JAC>(w, a, xs, us);
  }
  wave_sync();
  if (!(a.kp_method == MI_KP_SET_INTERVAL && a.minN == 1)) {   // ilqr.py:414
    interpolate<n, m>(w, nk);
    wave_sync();
  }
  return nk;
}

template <int m>
__device__ inline void invert_small(const double (&A)[m][m], double (&Ai)[m][m]) {
  static_assert(m >= 1 && m <= 2, "wave-per-problem path covers m <= 2");
  if constexpr (m == 1) {
    Ai[0][0] = 1.0 / A[0][0];
  } else {
    const double det = A[0][0] * A[1][1] - A[0][1] * A[1][0];
    const double id = 1.0 / det;
    Ai[0][0] = A[1][1] * id; Ai[0][1] = -A[0][1] * id;
    Ai[1][0] = -A[1][0] * id; Ai[1][1] = A[0][0] * id;
  }
}

// ---------------------------------------------------------------------------
// Backward Riccati pass (ilqr.py:623-667) with the quadratic cost expansion
// (:161-206) fused in.  Wave-uniform: every lane carries the same Vx/Vxx in
// registers; LDS reads are broadcasts, prefetched one step ahead.
// ---------------------------------------------------------------------------
template <class M>
__device__ inline void backward(const WS& w, const Consts<M>& c) {
  constexpr int n = M::n, m = M::m;
  const int N = w.N, SN = w.SN, SM = w.SM;
  const bool writer = threadIdx.x == 0;
  double Vx[n], Vxx[n][n];
  {
    double xT[n];
#pragma unroll
    for (int i = 0; i < n; ++i) xT[i] = w.xb[i * SN + N - 1];
#pragma unroll
    for (int i = 0; i < n; ++i) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < n; ++j) { s += (2.0 * c.Qf[i][j]) * xT[j]; Vxx[i][j] = 2.0 * c.Qf[i][j]; }
      Vx[i] = s - c.qfn[i];                                   // ilqr.py:203-204
    }
  }
  double x[n], u[m], fx[n][n], fu[n][m];
  auto fetch = [&](int t) {
#pragma unroll
    for (int i = 0; i < n; ++i) {
      x[i] = w.xb[i * SN + t];
#pragma unroll
      for (int j = 0; j < n; ++j) fx[i][j] = w.fx[(i * n + j) * SM + t];
#pragma unroll
      for (int k = 0; k < m; ++k) fu[i][k] = w.fu[(i * m + k) * SM + t];
    }
#pragma unroll
    for (int k = 0; k < m; ++k) u[k] = w.ub[k * SM + t];
  };
  fetch(N - 2);
  for (int t = N - 2; t >= 0; --t) {
    double xc[n], uc[m], fxc[n][n], fuc[n][m];
#pragma unroll
    for (int i = 0; i < n; ++i) {
      xc[i] = x[i];
#pragma unroll
      for (int j = 0; j < n; ++j) fxc[i][j] = fx[i][j];
#pragma unroll
      for (int k = 0; k < m; ++k) fuc[i][k] = fu[i][k];
    }
#pragma unroll
    for (int k = 0; k < m; ++k) uc[k] = u[k];
    fetch(t > 0 ? t - 1 : 0);

    // cost partials (ilqr.py:180-184): lx = 2Qx - 2x_nom^T Q, lu = 2Ru, lxx = 2Q, luu = 2R, lux = 0
    double Qx[n], Qu[m], Qxx[n][n], Quu[m][m], Qux[m][n];
#pragma unroll
    for (int i = 0; i < n; ++i) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < n; ++j) s += (2.0 * c.Q[i][j]) * xc[j];
      double g = 0.0;
#pragma unroll
      for (int k = 0; k < n; ++k) g += fxc[k][i] * Vx[k];
      Qx[i] = (s - c.qn[i]) + g;                              // :651
    }
#pragma unroll
    for (int a_ = 0; a_ < m; ++a_) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < m; ++j) s += (2.0 * c.R[a_][j]) * uc[j];
      double g = 0.0;
#pragma unroll
      for (int k = 0; k < n; ++k) g += fuc[k][a_] * Vx[k];
      Qu[a_] = s + g;                                         // :652
    }
    // A = fx^T Vxx (n x n), Bm = fu^T Vxx (m x n)  — the reference's association (fx.T@Vxx)@fx
    double A[n][n], Bm[m][n];
#pragma unroll
    for (int i = 0; i < n; ++i)
#pragma unroll
      for (int j = 0; j < n; ++j) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < n; ++k) s += fxc[k][i] * Vxx[k][j];
        A[i][j] = s;
      }
#pragma unroll
    for (int a_ = 0; a_ < m; ++a_)
#pragma unroll
      for (int j = 0; j < n; ++j) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < n; ++k) s += fuc[k][a_] * Vxx[k][j];
        Bm[a_][j] = s;
      }
#pragma unroll
    for (int i = 0; i < n; ++i)
#pragma unroll
      for (int j = 0; j < n; ++j) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < n; ++k) s += A[i][k] * fxc[k][j];
        Qxx[i][j] = 2.0 * c.Q[i][j] + s;                      // :653
      }
#pragma unroll
    for (int a_ = 0; a_ < m; ++a_) {
#pragma unroll
      for (int b_ = 0; b_ < m; ++b_) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < n; ++k) s += Bm[a_][k] * fuc[k][b_];
        Quu[a_][b_] = 2.0 * c.R[a_][b_] + s;                  // :654
      }
#pragma unroll
      for (int j = 0; j < n; ++j) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < n; ++k) s += Bm[a_][k] * fxc[k][j];
        Qux[a_][j] = s;                                       // :656 (lux = 0)
      }
    }
    double Qi[m][m];
    invert_small<m>(Quu, Qi);                                 // :655 explicit inverse
    double kap[m], Kg[m][n], QuQi[m];
#pragma unroll
    for (int a_ = 0; a_ < m; ++a_) {
      double s = 0.0, r = 0.0;
#pragma unroll
      for (int b_ = 0; b_ < m; ++b_) { s += Qi[a_][b_] * Qu[b_]; r += Qu[b_] * Qi[b_][a_]; }
      kap[a_] = s;                                            // :659
      QuQi[a_] = r;                                           // Qu^T Quu_inv
#pragma unroll
      for (int j = 0; j < n; ++j) {
        double g = 0.0;
#pragma unroll
        for (int b_ = 0; b_ < m; ++b_) g += Qi[a_][b_] * Qux[b_][j];
        Kg[a_][j] = g;                                        // :660
      }
    }
    double dv = 0.0;
#pragma unroll
    for (int a_ = 0; a_ < m; ++a_) dv += QuQi[a_] * Qu[a_];   // :663
    if (writer) {
#pragma unroll
      for (int a_ = 0; a_ < m; ++a_) {
        w.kap[a_ * SM + t] = kap[a_];
#pragma unroll
        for (int j = 0; j < n; ++j) w.K[(a_ * n + j) * SM + t] = Kg[a_][j];
      }
      w.dV[t] = dv;
    }
    // Vx = Qx - Qu^T Quu_inv Qux ; Vxx = Qxx - Qux^T Quu_inv Qux   (:666-667; no symmetrization)
#pragma unroll
    for (int j = 0; j < n; ++j) {
      double s = 0.0;
#pragma unroll
      for (int a_ = 0; a_ < m; ++a_) s += QuQi[a_] * Qux[a_][j];
      Vx[j] = Qx[j] - s;
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
        double s = 0.0;
#pragma unroll
        for (int b_ = 0; b_ < m; ++b_) s += QuxTQi[i][b_] * Qux[b_][j];
        Vxx[i][j] = Qxx[i][j] - s;
      }
  }
}

// ---------------------------------------------------------------------------
// The kernel: stage, run MODE, write back.
// ---------------------------------------------------------------------------
template <class M, int JAC, int MODE>
__global__ void __launch_bounds__(64) ilqr_small_kernel(const KArgs a) {
  constexpr int n = M::n, m = M::m;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  const int N = a.N;
  WS w = carve<n, m>(smem, N);
  const size_t oX = (size_t)b * n * N, oU = (size_t)b * m * (N - 1), oK = (size_t)b * m * n * (N - 1);
  const size_t oFx = (size_t)b * n * n * (N - 1), oFu = (size_t)b * n * m * (N - 1), oT = (size_t)b * (N - 1);

  const bool cold = a.cold != 0;
  copy_in(w.xb, a.x_bar + oX, n * N, cold);
  copy_in(w.ub, (a.u_pending ? a.u_guess : a.u_bar) + oU, m * (N - 1), false);
  copy_in(w.K, a.K + oK, m * n * (N - 1), cold);
  copy_in(w.kap, a.kappa + oU, m * (N - 1), cold);
  copy_in(w.dV, a.dV + oT, N - 1, cold);
  copy_in(w.fx, a.fx + oFx, n * n * (N - 1), cold);
  copy_in(w.fu, a.fu + oFu, n * m * (N - 1), cold);

  Consts<M> c;
  c.load(a.costmat);
  double x0r[n];
#pragma unroll
  for (int i = 0; i < n; ++i) x0r[i] = a.x0[(size_t)b * n + i];
  wave_sync();

  if (MODE == MODE_ROLLOUT) {
    double L, ex;
    rollout<M>(w, c, a, x0r, a.stage_in[b], lane == 0, L, ex);
    wave_sync();
    copy_out(a.x_trial + oX, w.xn, n * N);
    copy_out(a.u_trial + oU, w.un, m * (N - 1));
    if (lane == 0) { a.trial_cost[2 * b] = L; a.trial_cost[2 * b + 1] = ex; }
    return;
  }
  if (MODE == MODE_LINEARIZE) {
    const int nk = linearize<M, JAC>(w, a, w.xb, w.ub);
    copy_out(a.fx + oFx, w.fx, n * n * (N - 1));
    copy_out(a.fu + oFu, w.fu, n * m * (N - 1));
    for (int i = lane; i < nk; i += 64) a.kp_list[(size_t)b * (N - 1) + i] = w.kp[i];
    if (lane == 0) a.kp_count[b] = nk;
    return;
  }
  if (MODE == MODE_BACKWARD) {
    backward<M>(w, c);
    wave_sync();
    copy_out(a.K + oK, w.K, m * n * (N - 1));
    copy_out(a.kappa + oU, w.kap, m * (N - 1));
    copy_out(a.dV + oT, w.dV, N - 1);
    return;
  }

  // MODE_SOLVE / MODE_FORWARD: the Solve loop (ilqr.py:680-708)
  double L = (MODE == MODE_FORWARD) ? a.stage_in[b] : __builtin_inf();
  double improvement = __builtin_inf();
  int iters = 0, ls_total = 0, nk = 0;
  int status = MI_STATUS_CONVERGED;
  double* hist = a.hist + (size_t)b * a.hist_cap * 4;
  while (improvement > a.delta) {
    if (iters >= a.max_iters) { status = MI_STATUS_MAX_ITERS; break; }
    double L_new, eps; int trials;
    const bool ok = linesearch<M>(w, c, a, x0r, L, L_new, eps, trials);
    ls_total += trials;
    if (!ok) { status = MI_STATUS_LINESEARCH_FAILED; break; }
    wave_sync();
    nk = linearize<M, JAC>(w, a, w.xn, w.un);                  // at the ACCEPTED trajectory (:370)
    { double* t_; t_ = w.xb; w.xb = w.xn; w.xn = t_; t_ = w.ub; w.ub = w.un; w.un = t_; }   // :375-376
    if (MODE == MODE_SOLVE) { backward<M>(w, c); wave_sync(); }   // :697
    if (lane == 0 && iters < a.hist_cap) {
      hist[4 * iters + 0] = L_new; hist[4 * iters + 1] = eps;
      hist[4 * iters + 2] = (double)trials; hist[4 * iters + 3] = (double)nk / (double)(N - 1) * 100.0;   // :406
    }
    improvement = L - L_new;                                    // :706
    L = L_new;
    iters += 1;
    if (MODE == MODE_FORWARD) break;
  }
  wave_sync();
  copy_out(a.x_bar + oX, w.xb, n * N);
  copy_out(a.u_bar + oU, w.ub, m * (N - 1));
  copy_out(a.fx + oFx, w.fx, n * n * (N - 1));
  copy_out(a.fu + oFu, w.fu, n * m * (N - 1));
  if (MODE == MODE_SOLVE) {
    copy_out(a.K + oK, w.K, m * n * (N - 1));
    copy_out(a.kappa + oU, w.kap, m * (N - 1));
    copy_out(a.dV + oT, w.dV, N - 1);
  }
  for (int i = lane; i < nk; i += 64) a.kp_list[(size_t)b * (N - 1) + i] = w.kp[i];
  if (lane == 0) {
    a.cost[b] = L; a.iters[b] = iters; a.status[b] = status; a.ls_trials[b] = ls_total; a.kp_count[b] = nk;
  }
}

}  // namespace mi
