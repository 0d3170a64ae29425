// Workgroup-per-problem iLQR kernels for large state dimension (n ~ 36, m ~ 12), gfx950.
//
// One 256-thread workgroup (4 wavefronts, one per SIMD of a CU) owns one problem.
// The per-problem solver state of the reference (/root/reference/ilqr.py:70-83) is
// too large for LDS at this size (fx alone is n*n*(N-1)*8 = 404 KB at n=36,N=40), so
// it stays in HBM/L2 in a TIME-MAJOR layout ([t][row][col]: one time step's matrices
// are contiguous, loads are coalesced) and each sequential step stages what it needs
// through LDS:
//   rollout  (ilqr.py:306-327): K_t(x-x_bar) as 16-lane partial dot products, one
//            lane per degree of freedom for the dynamics, per-thread cost partials
//            reduced once per trial;
//   linearize(ilqr.py:380-415): (key-point, column) items over the 256 threads,
//            central differences or forward-mode duals; shared key-point code;
//   backward (ilqr.py:623-667): per step  T1 = [Vxx F | Vx],  H = F^T T1  with
//            F = [fx | fu]  (one (n+m)x(n+m+1) product yields Qxx,Qux,Quu,Qx,Qu at
//            once), register-tiled out of LDS; Quu is factorized (LDL^T, in
//            registers, no pivot search — Quu = 2R + fu^T Vxx fu) redundantly by the
//            n+1 threads that each solve one right-hand side (columns of Qux, and Qu),
//            so the solve needs no intra-step synchronization; Vxx <- Qxx - Qux^T K.
// The (B,...) arrays of this path are time-major in HBM; mi_ilqr_get/_set transpose
// to/from the reference's time-last layout at the boundary.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mi_ilqr.h"
#include "fastmath.hpp"
#include "ilqr_small.hpp"   // KArgs, KernelMode
#include "keypoints.hpp"
#include "models.hpp"

namespace mi {

constexpr int kLargeThreads = 256;

template <int n, int m>
struct LLay {
  static constexpr int nm = n + m;
  static constexpr int TS = ((nm + 1 + 3) / 4) * 4;      // padded row stride of T1 / H (multiple of 4)
  // doubles
  static constexpr int oQ = 0, oQf = oQ + n * n, oR = oQf + n * n, oXnom = oR + m * m, oQn = oXnom + n,
                       oQfn = oQn + n, oVxx = oQfn + n, oVx = oVxx + n * n, oF = oVx + n,
                       oT1 = oF + n * nm, oH = oT1 + n * TS, oXs = oH + nm * TS, oUs = oXs + n,
                       oRed = oUs + m, oXb = oRed + kLargeThreads, oEnd = oXb + n + m;
  static constexpr size_t doubles = oEnd + 8;
};

template <int n, int m>
__host__ __device__ constexpr size_t large_lds_bytes(int N) {
  return LLay<n, m>::doubles * 8 + (size_t)7 * N * 4 + 16;
}

// Per-problem views of the time-major HBM arrays.
template <int n, int m>
struct LView {
  double *X, *U, *K, *kap, *dV, *Fx, *Fu, *Xn, *Un;
  int N;
};

template <int n_, int m_>
struct LargeAcc {
  static constexpr int n = n_, m = m_;
  double *X, *Fx, *Fu;
  int *kp, *aux, *need, *binA, *binB;
  int N;
  __device__ __forceinline__ double x(int t, int i) const { return X[t * n + i]; }
  __device__ __forceinline__ double fx(int t, int r) const { return Fx[(size_t)t * n * n + r]; }
  __device__ __forceinline__ double fu(int t, int r) const { return Fu[(size_t)t * n * m + r]; }
  __device__ __forceinline__ void set_fx(int t, int r, double v) const { Fx[(size_t)t * n * n + r] = v; }
  __device__ __forceinline__ void set_fu(int t, int r, double v) const { Fu[(size_t)t * n * m + r] = v; }
};

__device__ __forceinline__ double block_sum(double v, double* red) {
  // deterministic fixed-order tree: 64-lane butterfly, then 4 wave partials
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// One line-search trial (ilqr.py:306-327).  Returns L on every thread; trajectory -> Xn/Un.
template <class M>
__device__ inline double large_rollout(const LView<M::n, M::m>& v, double* lds, const KArgs& a,
                                       const double* x0g, double eps, double& expd_out) {
  constexpr int n = M::n, m = M::m;
  using Ly = LLay<n, m>;
  const int tid = threadIdx.x, N = v.N;
  double* xs = lds + Ly::oXs;
  double* us = lds + Ly::oUs;
  double* xb = lds + Ly::oXb;          // x_bar_t | (u_bar_t - eps*kappa_t)
  const double* Q = lds + Ly::oQ;
  const double* R = lds + Ly::oR;
  const double* Qf = lds + Ly::oQf;
  const double* xnom = lds + Ly::oXnom;
  if (tid < n) { xs[tid] = x0g[tid]; v.Xn[tid] = x0g[tid]; }
  double acc = 0.0;                    // per-thread cost partial over all time steps
  __syncthreads();
  for (int t = 0; t < N - 1; ++t) {
    // u_t = u_bar_t - eps*kappa_t - K_t (x_t - x_bar_t)   (ilqr.py:313): 16 lanes per control row
    if (tid < m * 16) {
      const int k = tid >> 4, l = tid & 15;
      const double* Kr = v.K + ((size_t)t * m + k) * n;
      const double* xbt = v.X + (size_t)t * n;
      double p = 0.0;
      for (int j = l; j < n; j += 16) p += Kr[j] * (xs[j] - xbt[j]);
      p += __shfl_xor(p, 8, 16);
      p += __shfl_xor(p, 4, 16);
      p += __shfl_xor(p, 2, 16);
      p += __shfl_xor(p, 1, 16);
      if (l == 0) us[k] = (v.U[(size_t)t * m + k] - eps * v.kap[(size_t)t * m + k]) - p;
    }
    __syncthreads();
    // dynamics: one lane per degree of freedom (ilqr.py:316); cost rows on the other waves (:325)
    double qn_ = 0.0, vn_ = 0.0;
    if (tid < M::nq) {
      M::template dof<double>(tid, xs, us, qn_, vn_, a.params, a.dt);
    } else if (tid >= 64 && tid < 64 + n) {
      const int i = tid - 64;
      double r = 0.0;
      for (int j = 0; j < n; ++j) r += Q[i * n + j] * (xs[j] - xnom[j]);
      acc += (xs[i] - xnom[i]) * r;
    } else if (tid >= 128 && tid < 128 + m) {
      const int k = tid - 128;
      double r = 0.0;
      for (int j = 0; j < m; ++j) r += R[k * m + j] * us[j];
      acc += us[k] * r;
      v.Un[(size_t)t * m + k] = us[k];
    }
    __syncthreads();
    if (tid < M::nq) {
      xs[tid] = qn_; xs[M::nq + tid] = vn_;
      v.Xn[(size_t)(t + 1) * n + tid] = qn_;
      v.Xn[(size_t)(t + 1) * n + M::nq + tid] = vn_;
    }
    __syncthreads();
  }
  if (tid >= 64 && tid < 64 + n) {     // terminal cost (ilqr.py:327)
    const int i = tid - 64;
    double r = 0.0;
    for (int j = 0; j < n; ++j) r += Qf[i * n + j] * (xs[j] - xnom[j]);
    acc += (xs[i] - xnom[i]) * r;
  }
  double dvp = 0.0;
  for (int t = tid; t < N - 1; t += kLargeThreads) dvp += v.dV[t];
  double* red = lds + Ly::oRed;
  const double L = block_sum(acc, red);
  const double dvs = block_sum(dvp, red);
  expd_out = -eps * (1.0 - eps / 2.0) * dvs;             // ilqr.py:326
  (void)xb;
  return L;
}

// Sequential line search (ilqr.py:300-337); on accept Xn/Un hold the trajectory.
template <class M>
__device__ inline bool large_linesearch(const LView<M::n, M::m>& v, double* lds, const KArgs& a, const double* x0g,
                                        double L_last, double& L_out, double& eps_out, int& trials) {
  double eps = 1.0;
  trials = 0;
  while (eps >= 1e-8) {
    trials += 1;
    double ex;
    const double L = large_rollout<M>(v, lds, a, x0g, eps, ex);
    if ((L_last - L) > a.gamma * ex) { L_out = L; eps_out = eps; return true; }
    eps *= a.beta;
    __syncthreads();
  }
  return false;
}

// Dynamics partials at the listed time steps of the nominal trajectory (X,U).
template <class M, int JAC>
__device__ __forceinline__ void large_jac_at(const LView<M::n, M::m>& v, const KArgs& a, const int* list, int count) {
  constexpr int n = M::n, m = M::m, nc = n + m;
  const double h = a.fd_h, inv2h = 1.0 / (2.0 * h);
  for (int it = threadIdx.x; it < count * nc; it += kLargeThreads) {
    const int ki = it / nc, col = it - ki * nc;
    const int t = list[ki];
    const double* xg = v.X + (size_t)t * n;
    const double* ug = v.U + (size_t)t * m;
    double d[n];
    if (JAC == MI_JAC_FD_CENTRAL) {
      double x[n], u[m], f[n];
#pragma unroll
      for (int i = 0; i < n; ++i) x[i] = (col == i) ? xg[i] + h : xg[i];
#pragma unroll
      for (int k = 0; k < m; ++k) u[k] = (col == n + k) ? ug[k] + h : ug[k];
      M::template step<double>(x, u, d, a.params, a.dt);
#pragma unroll
      for (int i = 0; i < n; ++i) x[i] = (col == i) ? xg[i] - h : xg[i];
#pragma unroll
      for (int k = 0; k < m; ++k) u[k] = (col == n + k) ? ug[k] - h : ug[k];
      M::template step<double>(x, u, f, a.params, a.dt);
#pragma unroll
      for (int i = 0; i < n; ++i) d[i] = (d[i] - f[i]) * inv2h;
    } else {
      Dual1 xd[n], ud[m], fd[n];
#pragma unroll
      for (int i = 0; i < n; ++i) xd[i] = Dual1(xg[i], (col == i) ? 1.0 : 0.0);
#pragma unroll
      for (int k = 0; k < m; ++k) ud[k] = Dual1(ug[k], (col == n + k) ? 1.0 : 0.0);
      M::template step<Dual1>(xd, ud, fd, a.params, a.dt);
#pragma unroll
      for (int i = 0; i < n; ++i) d[i] = fd[i].d;
    }
    if (col < n) {
      double* o = v.Fx + (size_t)t * n * n + col;
#pragma unroll
      for (int i = 0; i < n; ++i) o[i * n] = d[i];
    } else {
      double* o = v.Fu + (size_t)t * n * m + (col - n);
#pragma unroll
      for (int i = 0; i < n; ++i) o[i * m] = d[i];
    }
  }
}

// C[r][c] (+)= sum_k A(r,k) * B(k,c) for a TR x TC register tile; A/B/C in LDS.
template <int TR, int TC, int KD, class AFn, class BFn>
__device__ __forceinline__ void tile_mm(double (&acc)[TR][TC], AFn A, BFn B) {
#pragma unroll 4
  for (int k = 0; k < KD; ++k) {
    double av[TR], bv[TC];
#pragma unroll
    for (int r = 0; r < TR; ++r) av[r] = A(r, k);
#pragma unroll
    for (int c = 0; c < TC; ++c) bv[c] = B(k, c);
#pragma unroll
    for (int r = 0; r < TR; ++r)
#pragma unroll
      for (int c = 0; c < TC; ++c) acc[r][c] += av[r] * bv[c];
  }
}

// Backward Riccati pass (ilqr.py:623-667), cost expansion (:161-206) fused.
#ifdef MI_PROF_BACKWARD
#define BP_TICK(k) do { const long long c_ = clock64(); bp_acc[k] += c_ - bp_last; bp_last = c_; } while (0)
#else
#define BP_TICK(k) do {} while (0)
#endif

template <class M>
__device__ inline void large_backward(const LView<M::n, M::m>& v, double* lds, long long* bp_acc = nullptr) {
#ifdef MI_PROF_BACKWARD
  long long bp_last = clock64();
#endif
  constexpr int n = M::n, m = M::m, nm = n + m;
  using Ly = LLay<n, m>;
  constexpr int TS = Ly::TS;
  const int tid = threadIdx.x, N = v.N;
  const double* Q = lds + Ly::oQ;
  const double* R = lds + Ly::oR;
  const double* Qf = lds + Ly::oQf;
  const double* qn = lds + Ly::oQn;
  const double* qfn = lds + Ly::oQfn;
  double* Vxx = lds + Ly::oVxx;
  double* Vx = lds + Ly::oVx;
  double* F = lds + Ly::oF;          // [n][nm]  = [fx | fu]
  double* T1 = lds + Ly::oT1;        // [n][TS]  = [Vxx F | Vx]
  double* H = lds + Ly::oH;          // [nm][TS] = F^T T1
  double* xb = lds + Ly::oXb;        // x_bar_t (n) | u_bar_t (m)

  // terminal: Vx = 2 Qf x_T - 2 x_nom^T Qf ; Vxx = 2 Qf   (ilqr.py:203-204, :638)
  for (int e = tid; e < n * n; e += kLargeThreads) Vxx[e] = 2.0 * Qf[e];
  if (tid < n) {
    const double* xT = v.X + (size_t)(N - 1) * n;
    double s = 0.0;
    for (int j = 0; j < n; ++j) s += (2.0 * Qf[tid * n + j]) * xT[j];
    Vx[tid] = s - qfn[tid];
  }
  __syncthreads();

  for (int t = N - 2; t >= 0; --t) {
    // stage F = [fx_t | fu_t], x_bar_t, u_bar_t
    const double* fxg = v.Fx + (size_t)t * n * n;
    const double* fug = v.Fu + (size_t)t * n * m;
    for (int e = tid; e < n * n; e += kLargeThreads) { const int i = e / n, j = e - i * n; F[i * nm + j] = fxg[e]; }
    for (int e = tid; e < n * m; e += kLargeThreads) { const int i = e / m, k = e - i * m; F[i * nm + n + k] = fug[e]; }
    if (tid < n) xb[tid] = v.X[(size_t)t * n + tid];
    else if (tid < nm) xb[tid] = v.U[(size_t)t * m + (tid - n)];
    __syncthreads();
    BP_TICK(0);
    // T1 = Vxx F  (n x nm), 2x4 tiles;  T1[:, nm] = Vx
    {
      constexpr int TR = 2, TC = 4, tr = n / TR, tc = nm / TC;
      static_assert(n % TR == 0 && nm % TC == 0, "tile shape");
      for (int tile = tid; tile < tr * tc; tile += kLargeThreads) {
        const int i0 = (tile / tc) * TR, j0 = (tile % tc) * TC;
        double acc[TR][TC] = {};
        tile_mm<TR, TC, n>(acc, [&](int r, int k) { return Vxx[(i0 + r) * n + k]; },
                           [&](int k, int c) { return F[k * nm + j0 + c]; });
#pragma unroll
        for (int r = 0; r < TR; ++r)
#pragma unroll
          for (int c = 0; c < TC; ++c) T1[(i0 + r) * TS + j0 + c] = acc[r][c];
      }
      if (tid < n) T1[tid * TS + nm] = Vx[tid];
    }
    __syncthreads();
    BP_TICK(1);
    // H = F^T T1  ((nm) x (nm+1)), 3x4 tiles; the unused fx^T Vxx fu block is skipped
    {
      constexpr int TR = 3, TC = 4, tr = nm / TR, tc = TS / TC;
      static_assert(nm % TR == 0, "tile shape");
      for (int tile = tid; tile < tr * tc; tile += kLargeThreads) {
        const int p0 = (tile / tc) * TR, q0 = (tile % tc) * TC;
        if (p0 + TR <= n && q0 >= n && q0 + TC <= nm) continue;
        double acc[TR][TC] = {};
        tile_mm<TR, TC, n>(acc, [&](int r, int k) { return F[k * nm + p0 + r]; },
                           [&](int k, int c) { return T1[k * TS + q0 + c]; });
#pragma unroll
        for (int r = 0; r < TR; ++r)
#pragma unroll
          for (int c = 0; c < TC; ++c) H[(p0 + r) * TS + q0 + c] = acc[r][c];
      }
    }
    __syncthreads();
    // first-order terms into column nm of H: Qx = lx + fx^T Vx, Qu = lu + fu^T Vx   (:651-652)
    if (tid < n) {
      double s = 0.0;
      for (int j = 0; j < n; ++j) s += (2.0 * Q[tid * n + j]) * xb[j];
      H[tid * TS + nm] += s - qn[tid];
    } else if (tid < nm) {
      const int k = tid - n;
      double s = 0.0;
      for (int j = 0; j < m; ++j) s += (2.0 * R[k * m + j]) * xb[n + j];
      H[tid * TS + nm] += s;
    }
    __syncthreads();
    BP_TICK(2);
    // Quu = 2R + H[n:,n:] ; solve Quu * Y = [Qux | Qu]: one right-hand side per thread (:655-660)
    if (tid <= n) {
      double A[m][m];
#pragma unroll
      for (int i = 0; i < m; ++i)
#pragma unroll
        for (int j = 0; j < m; ++j) A[i][j] = 2.0 * R[i * m + j] + H[(n + i) * TS + n + j];
      // LDL^T (unit lower L stored below the diagonal of A, D on it)
#pragma unroll
      for (int j = 0; j < m; ++j) {
        double dj = A[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) dj -= A[j][k] * A[j][k] * A[k][k];
        A[j][j] = dj;
        const double idj = 1.0 / dj;
#pragma unroll
        for (int i = j + 1; i < m; ++i) {
          double s = A[i][j];
#pragma unroll
          for (int k = 0; k < j; ++k) s -= A[i][k] * A[j][k] * A[k][k];
          A[i][j] = s * idj;
        }
      }
      double y[m];
#pragma unroll
      for (int i = 0; i < m; ++i) y[i] = H[(n + i) * TS + (tid < n ? tid : nm)];
#pragma unroll
      for (int i = 0; i < m; ++i)
#pragma unroll
        for (int k = 0; k < i; ++k) y[i] -= A[i][k] * y[k];
#pragma unroll
      for (int i = 0; i < m; ++i) y[i] /= A[i][i];
#pragma unroll
      for (int i = m - 1; i >= 0; --i)
#pragma unroll
        for (int k = i + 1; k < m; ++k) y[i] -= A[k][i] * y[k];
      if (tid < n) {
        // K_t[:, tid]  (:660) -> HBM and into rows n.. of T1 (scratch) for the Vxx update
#pragma unroll
        for (int i = 0; i < m; ++i) {
          v.K[((size_t)t * m + i) * n + tid] = y[i];
          T1[i * TS + tid] = y[i];
        }
      } else {
        double dv = 0.0;
#pragma unroll
        for (int i = 0; i < m; ++i) {
          v.kap[(size_t)t * m + i] = y[i];                       // kappa_t (:659)
          T1[i * TS + nm] = y[i];
          dv += H[(n + i) * TS + nm] * y[i];                      // Qu^T Quu^{-1} Qu (:663)
        }
        v.dV[t] = dv;
      }
    }
    __syncthreads();
    BP_TICK(3);
    // Vxx = Qxx - Qux^T K ; Vx = Qx - Qux^T kappa  (:666-667), K/kappa staged in T1 rows 0..m-1
    for (int e = tid; e < n * (n + 1); e += kLargeThreads) {
      const int i = e / (n + 1), j = e - i * (n + 1);
      const int col = (j < n) ? j : nm;
      double s = H[i * TS + col];
      if (j < n) s += 2.0 * Q[i * n + j];
#pragma unroll
      for (int a_ = 0; a_ < m; ++a_) s -= H[(n + a_) * TS + i] * T1[a_ * TS + col];
      if (j < n) Vxx[i * n + j] = s; else Vx[i] = s;
    }
    __syncthreads();
    BP_TICK(4);
  }
}

template <class M, int JAC, int MODE>
__global__ void __launch_bounds__(kLargeThreads) ilqr_large_kernel(const KArgs a) {
  constexpr int n = M::n, m = M::m;
  using Ly = LLay<n, m>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* lds = reinterpret_cast<double*>(smem);
  int* ilds = reinterpret_cast<int*>(lds + Ly::doubles);
  const int b = blockIdx.x, tid = threadIdx.x, N = a.N;
  LView<n, m> v;
  v.N = N;
  v.X = a.x_bar + (size_t)b * n * N;
  v.U = a.u_bar + (size_t)b * m * (N - 1);
  v.K = a.K + (size_t)b * m * n * (N - 1);
  v.kap = a.kappa + (size_t)b * m * (N - 1);
  v.dV = a.dV + (size_t)b * (N - 1);
  v.Fx = a.fx + (size_t)b * n * n * (N - 1);
  v.Fu = a.fu + (size_t)b * n * m * (N - 1);
  v.Xn = a.x_trial + (size_t)b * n * N;
  v.Un = a.u_trial + (size_t)b * m * (N - 1);
  LargeAcc<n, m> acc;
  acc.X = v.X; acc.Fx = v.Fx; acc.Fu = v.Fu; acc.N = N;
  acc.kp = ilds; acc.aux = ilds + N; acc.need = ilds + 2 * N; acc.binA = ilds + 3 * N; acc.binB = ilds + 5 * N;
  const double* x0g = a.x0 + (size_t)b * n;

  // cost constants -> LDS ; 2 x_nom^T Q and 2 x_nom^T Qf (ilqr.py:180,203)
  {
    const double* cm = a.costmat;
    for (int e = tid; e < n * n; e += kLargeThreads) { lds[Ly::oQ + e] = cm[e]; lds[Ly::oQf + e] = cm[n * n + m * m + e]; }
    for (int e = tid; e < m * m; e += kLargeThreads) lds[Ly::oR + e] = cm[n * n + e];
    if (tid < n) lds[Ly::oXnom + tid] = cm[2 * n * n + m * m + tid];
    __syncthreads();
    if (tid < n) {
      double s = 0.0, sf = 0.0;
      for (int i = 0; i < n; ++i) {
        s += (2.0 * lds[Ly::oXnom + i]) * lds[Ly::oQ + i * n + tid];
        sf += (2.0 * lds[Ly::oXnom + i]) * lds[Ly::oQf + i * n + tid];
      }
      lds[Ly::oQn + tid] = s; lds[Ly::oQfn + tid] = sf;
    }
  }
  // lazily-zero persistent state / pending initial guess
  if (a.cold) {
    for (int e = tid; e < n * N; e += kLargeThreads) v.X[e] = 0.0;
    for (int e = tid; e < m * n * (N - 1); e += kLargeThreads) v.K[e] = 0.0;
    for (int e = tid; e < m * (N - 1); e += kLargeThreads) v.kap[e] = 0.0;
    for (int e = tid; e < N - 1; e += kLargeThreads) v.dV[e] = 0.0;
    for (int e = tid; e < n * n * (N - 1); e += kLargeThreads) v.Fx[e] = 0.0;
    for (int e = tid; e < n * m * (N - 1); e += kLargeThreads) v.Fu[e] = 0.0;
  }
  if (a.u_pending) {
    const double* ug = a.u_guess + (size_t)b * m * (N - 1);
    for (int e = tid; e < m * (N - 1); e += kLargeThreads) v.U[e] = ug[e];
  }
  __syncthreads();

  auto jac = [&](const int* list, int count) __attribute__((always_inline)) { large_jac_at<M, JAC>(v, a, list, count); };
  auto do_linearize = [&]() __attribute__((always_inline)) {
    return linearize_generic(acc, a.kp_method, a.minN, a.maxN, a.jerk_thr, a.err_thr, jac);
  };

  if (MODE == MODE_ROLLOUT) {
    double ex;
    const double L = large_rollout<M>(v, lds, a, x0g, a.stage_in[b], ex);
    if (tid == 0) { a.trial_cost[2 * b] = L; a.trial_cost[2 * b + 1] = ex; }
    return;
  }
  if (MODE == MODE_LINEARIZE) {
    const int nk = do_linearize();
    for (int i = tid; i < nk; i += kLargeThreads) a.kp_list[(size_t)b * (N - 1) + i] = acc.kp[i];
    if (tid == 0) a.kp_count[b] = nk;
    return;
  }
  if (MODE == MODE_BACKWARD) {
    large_backward<M>(v, lds);
    return;
  }

  double L = (MODE == MODE_FORWARD) ? a.stage_in[b] : __builtin_inf();
  double improvement = __builtin_inf();
  int iters = 0, ls_total = 0, nk = 0;
  int status = MI_STATUS_CONVERGED;
  double* hist = a.hist + (size_t)b * a.hist_cap * 4;
  long long c_ls = 0, c_lin = 0, c_bp = 0;
  const long long c_begin = clock64();
  while (improvement > a.delta) {
    if (iters >= a.max_iters) { status = MI_STATUS_MAX_ITERS; break; }
    double L_new, eps; int trials;
    const long long c0 = clock64();
    const bool ok = large_linesearch<M>(v, lds, a, x0g, L, L_new, eps, trials);
    ls_total += trials;
    if (!ok) { status = MI_STATUS_LINESEARCH_FAILED; break; }
    __syncthreads();
    const long long c1 = clock64();
    for (int e = tid; e < n * N; e += kLargeThreads) v.X[e] = v.Xn[e];              // :375-376
    for (int e = tid; e < m * (N - 1); e += kLargeThreads) v.U[e] = v.Un[e];
    __syncthreads();
    nk = do_linearize();                                                             // :370
    __syncthreads();
    const long long c2 = clock64();
#ifdef MI_PROF_BACKWARD
    long long bpa[5] = {0, 0, 0, 0, 0};
    if (MODE == MODE_SOLVE) { large_backward<M>(v, lds, bpa); __syncthreads(); }
    if (tid == 0 && iters == 0) { for (int q_ = 0; q_ < 4; ++q_) a.prof[4 * b + q_] = bpa[q_]; a.hist[(size_t)b * a.hist_cap * 4 + 4 * (a.hist_cap - 1)] = (double)bpa[4]; }
#else
    if (MODE == MODE_SOLVE) { large_backward<M>(v, lds); __syncthreads(); }          // :697
#endif
    const long long c3 = clock64();
    c_ls += c1 - c0; c_lin += c2 - c1; c_bp += c3 - c2;
    if (tid == 0 && iters < a.hist_cap) {
      hist[4 * iters + 0] = L_new; hist[4 * iters + 1] = eps;
      hist[4 * iters + 2] = (double)trials; hist[4 * iters + 3] = (double)nk / (double)(N - 1) * 100.0;
    }
    improvement = L - L_new;
    L = L_new;
    iters += 1;
    if (MODE == MODE_FORWARD) break;
  }
  for (int i = tid; i < nk; i += kLargeThreads) a.kp_list[(size_t)b * (N - 1) + i] = acc.kp[i];
  if (tid == 0) {
    a.cost[b] = L; a.iters[b] = iters; a.status[b] = status; a.ls_trials[b] = ls_total; a.kp_count[b] = nk;
#ifndef MI_PROF_BACKWARD
    a.prof[4 * b + 0] = c_ls; a.prof[4 * b + 1] = c_lin; a.prof[4 * b + 2] = c_bp; a.prof[4 * b + 3] = clock64() - c_begin;
#endif
  }
}

}  // namespace mi
