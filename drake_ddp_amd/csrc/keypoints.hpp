// Derivative key-pointing shared by the small-state and large-state kernels
// (/root/reference/ilqr.py:380-621): key-point selection by the three methods of
// utils_derivs_interpolation, and linear interpolation of fx/fu between key-points.
//
// Written against an accessor `Acc` so the same code runs on the LDS records of the
// wave-per-problem kernels and on the time-major HBM arrays of the workgroup-per-
// problem kernels:
//   int N;  int *kp,*aux,*need,*binA,*binB;          integer scratch (LDS), N / 2N ints
//   double x(t,i)                                     nominal state
//   double fx(t,r), fu(t,r); void set_fx/set_fu       partials, r = row-major element index
//   static constexpr int n, m
// Every wave of the workgroup executes these functions redundantly with identical
// results (lane = threadIdx.x & 63), so control flow is workgroup-uniform and the
// __syncthreads() inside are legal for any block size that is a multiple of 64.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/mi_ilqr.h"

namespace mi {

// Ordered stream compaction of {t in [0,count) : pred(t)} into list; returns size.
template <class Pred>
__device__ inline int compact(int count, int* list, Pred pred) {
  const int lane = threadIdx.x & 63;
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
template <class Acc>
__device__ inline int keypoints_set_interval(const Acc& acc, int minN) {
  const int N = acc.N;
  const int count = (N - 2) / minN + 1;            // len(arange(0, N-1, minN))
  for (int i = threadIdx.x & 63; i < count; i += 64) {
    int v = i * minN;
    if (i == count - 1 && v != N - 2) v = N - 2;   // overwrite, not append (:428-430)
    acc.kp[i] = v;
  }
  return count;
}

// get_keypoints_adaptive_jerk + calc_jerk_profile (ilqr.py:434-486).  The jerk test
// is evaluated for 64 time steps at once; the counter automaton then walks the ballot
// mask with scalar code.
template <class Acc>
__device__ inline int keypoints_adaptive_jerk(const Acc& acc, int minN, int maxN, double jerk_thr) {
  const int N = acc.N, lane = threadIdx.x & 63;
  constexpr int dof = Acc::n / 2;
  int nk = 0, since = 0, last = 0;
  if (lane == 0) acc.kp[0] = 0;
  nk = 1;
  for (int t0 = 0; t0 < N - 3; t0 += 64) {
    const int t = t0 + lane;
    bool trig = false;
    if (t < N - 3) {
      for (int i = 0; i < dof; ++i) {
        const double v0 = acc.x(t, i + dof), v1 = acc.x(t + 1, i + dof), v2 = acc.x(t + 2, i + dof);
        const double jerk = (v2 - v1) - (v1 - v0);           // signed, no abs (:481-484)
        trig = trig || (jerk > jerk_thr);
      }
    }
    const unsigned long long mask = __ballot(trig);
    const int lim = (N - 3 - t0) < 64 ? (N - 3 - t0) : 64;
    for (int j = 0; j < lim; ++j) {
      since += 1;
      if (since >= minN && ((mask >> j) & 1ull)) {
        if (lane == 0) acc.kp[nk] = t0 + j;
        last = t0 + j; nk += 1; since = 0;
      }
      if (since >= maxN) {
        if (lane == 0) acc.kp[nk] = t0 + j;
        last = t0 + j; nk += 1; since = 0;
      }
    }
  }
  if (last != N - 2 && lane == 0) acc.kp[nk - 1] = N - 2;    // :465-466
  return nk;
}

// get_keypoints_iterative_error + check_one_matrix_error (ilqr.py:488-593):
// level-synchronous bisection, one lane per bin; `jac(list,count)` evaluates (and
// stores) the partials only where the reference would evaluate them.
template <class Acc, class JacFn>
__device__ __forceinline__ int keypoints_iterative_error(const Acc& acc, int minN, double err_thr, JacFn jac) {
  constexpr int n = Acc::n;
  const int N = acc.N, lane = threadIdx.x & 63;
  int* done = acc.aux;           // 0/1 per time step: derivative evaluated (deriv_calculated_at_index)
  int* need = acc.need;          // scratch flags: indices a level wants evaluated
  for (int t = lane; t < N; t += 64) done[t] = 0;
  int* bins = acc.binA;          // (s,e) pairs
  int* next = acc.binB;
  int nb = 1;
  if (lane == 0) { bins[0] = 0; bins[1] = N - 2; }
  __syncthreads();
  // A level's bins are disjoint sub-intervals of [0,N-2] of width >= 1: at most N-1
  // pairs = 2(N-1) ints per buffer.
  for (;;) {
    for (int t = lane; t < N; t += 64) need[t] = 0;
    __syncthreads();
    for (int i = lane; i < nb; i += 64) {
      const int s = bins[2 * i], e = bins[2 * i + 1];
      if (e - s > minN) { const int mid = (s + e) / 2; need[s] = 1; need[mid] = 1; need[e] = 1; }
    }
    __syncthreads();
    const int cnt = compact(N, acc.kp, [&](int t) { return need[t] && !done[t]; });
    __syncthreads();
    jac(acc.kp, cnt);
    __syncthreads();
    for (int i = lane; i < cnt; i += 64) done[acc.kp[i]] = 1;
    __syncthreads();
    // evaluate bins; bad ones are split (order within a level is irrelevant to the result)
    int nn = 0;
    for (int i0 = 0; i0 < nb; i0 += 64) {
      const int i = i0 + lane;
      bool bad = false;
      int s = 0, e = 0, mid = 0;
      if (i < nb) {
        s = bins[2 * i]; e = bins[2 * i + 1]; mid = (s + e) / 2;
        if (e - s > minN) {
          double sum = 0.0;
          for (int r = 0; r < n * n; ++r) {
            const double lin = (acc.fx(e, r) + acc.fx(s, r)) / 2.0;
            const double df = lin - acc.fx(mid, r);
            sum += df * df;
          }
          bad = (sum / (2.0 * n)) > err_thr;         // divisor 2n, fx only (:583-591)
        }
      }
      const unsigned long long mask = __ballot(bad);
      const int pos = nn + __popcll(mask & ((1ull << lane) - 1ull));
      if (bad) { next[4 * pos] = s; next[4 * pos + 1] = mid; next[4 * pos + 2] = mid; next[4 * pos + 3] = e; }
      nn += __popcll(mask);
    }
    __syncthreads();
    if (nn == 0) break;
    nb = 2 * nn;
    int* tmp = bins; bins = next; next = tmp;
  }
  const int nk = compact(N - 1, acc.kp, [&](int t) { return done[t] != 0; });
  __syncthreads();
  return nk;
}

// interpolate_derivatives (ilqr.py:596-621): (segment, element) pairs over all threads
// of the workgroup; interior points only (the end points are reproduced exactly by
// the formula fs + (fe-fs)*0/len and are only read).
template <class Acc>
__device__ inline void interpolate(const Acc& acc, int nk) {
  constexpr int n = Acc::n, m = Acc::m;
  constexpr int cnt = n * n + n * m;
  const int items = (nk - 1) * cnt;
  for (int it = threadIdx.x; it < items; it += blockDim.x) {
    const int seg = it / cnt, r = it - seg * cnt;
    const int s = acc.kp[seg], e = acc.kp[seg + 1];
    if (e - s < 2) continue;
    const double len = (double)(e - s);
    if (r < n * n) {
      const double fs = acc.fx(s, r), fe = acc.fx(e, r);
      for (int j = s + 1; j < e; ++j) acc.set_fx(j, r, fs + (fe - fs) * (double)(j - s) / len);
    } else {
      const int q = r - n * n;
      const double fs = acc.fu(s, q), fe = acc.fu(e, q);
      for (int j = s + 1; j < e; ++j) acc.set_fu(j, q, fs + (fe - fs) * (double)(j - s) / len);
    }
  }
}

// _get_derivatives (ilqr.py:380-415).  Returns the key-point count (list in acc.kp).
template <class Acc, class JacFn>
__device__ __forceinline__ int linearize_generic(const Acc& acc, int kp_method, int minN, int maxN, double jerk_thr,
                                        double err_thr, JacFn jac) {
  int nk;
  if (kp_method != MI_KP_ITERATIVE_ERROR) {
    nk = (kp_method == MI_KP_SET_INTERVAL) ? keypoints_set_interval(acc, minN)
                                           : keypoints_adaptive_jerk(acc, minN, maxN, jerk_thr);
    __syncthreads();
    jac(acc.kp, nk);
  } else {
    nk = keypoints_iterative_error(acc, minN, err_thr, jac);
  }
  __syncthreads();
  if (!(kp_method == MI_KP_SET_INTERVAL && minN == 1)) {     // ilqr.py:414
    interpolate(acc, nk);
    __syncthreads();
  }
  return nk;
}

}  // namespace mi
