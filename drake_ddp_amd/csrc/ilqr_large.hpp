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
  static constexpr int TS = ((n + m + 15) / 16) * 16 + 4;  // row stride of T1 / H: whole tiles + the Vx/first-order column
  // doubles
  static constexpr int QC = m * (m + 1) / 2;               // packed lower triangle of Quu
  static constexpr int T16 = 16;                           // MFMA tile edge
  static constexpr int NP = ((n + 15) / 16) * 16;          // n padded to whole tiles (rows of Vxx)
  static constexpr int NMP = ((nm + 15) / 16) * 16;
  static constexpr int VS = n + 1;                         // odd row stride of Vxx: conflict-free column-of-tile reads
  static constexpr int oQ = 0, oQf = oQ + n * n, oR = oQf + n * n, oXnom = oR + m * m, oQn = oXnom + n,
                       oQfn = oQn + n, oVxx = oQfn + n, oVx = oVxx + NP * VS, oF = oVx + n,
                       oT1 = oF + n * NMP, oH = oT1 + n * TS, oXs = oH + NMP * TS, oUs = oXs + n,
                       oRed = oUs + m, oXb = oRed + kLargeThreads, oQc = oXb + n + m + ((n + m) & 1),
                       oQT = oQc + m * m + m + (m & 1), oEnd = oQT + n * n;
  static constexpr size_t doubles = oEnd + 8;
};

template <int n, int m>
__host__ __device__ constexpr size_t large_lds_bytes(int N) {
  // fixed block + per-step cost gradients [N][n+m] + integer scratch of the key-point code
  return (LLay<n, m>::doubles + (size_t)N * (n + m)) * 8 + (size_t)7 * N * 4 + 16;
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

// Workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does NOT drain the
// vector-memory counter, so global prefetches issued before it stay in flight across it (and
// global stores are not waited for).  Use only where no thread reads global data another thread
// of the workgroup wrote since the last full __syncthreads().
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

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
// Per step: (1) 16 lanes per control row form K_t(x-x_bar) partial dots — K_t, x_bar_t,
// u_bar_t, kappa_t come from HBM/L2 and are prefetched one step ahead into registers;
// (2) one lane per degree of freedom advances the dynamics while other waves add the
// stage-cost rows (their Q/R rows live in registers); (3) the new state is published.
template <class M>
__device__ inline double large_rollout(const LView<M::n, M::m>& v, double* lds, const KArgs& a,
                                       const double* x0g, double eps, double& expd_out) {
  constexpr int n = M::n, m = M::m;
  using Ly = LLay<n, m>;
  constexpr int JR = (n + 15) / 16;    // K-row elements per lane
  const int tid = threadIdx.x, N = v.N;
  double* xs = lds + Ly::oXs;
  double* us = lds + Ly::oUs;
  const double* xnom = lds + Ly::oXnom;
  const bool urole = tid < m * 16;
  const int uk = tid >> 4, ul = tid & 15;
  const bool qrole = tid >= 64 && tid < 64 + n;      // cost row i = tid-64
  const bool rrole = tid >= 128 && tid < 128 + m;    // control-cost row k = tid-128
  // cost rows -> registers (one-off)
  double qrow[n], rrow[m];
  if (qrole) {
#pragma unroll
    for (int j = 0; j < n; ++j) qrow[j] = lds[Ly::oQ + (tid - 64) * n + j];
  }
  if (rrole) {
#pragma unroll
    for (int j = 0; j < m; ++j) rrow[j] = lds[Ly::oR + (tid - 128) * m + j];
  }
  if (tid < n) { xs[tid] = x0g[tid]; v.Xn[tid] = x0g[tid]; }
  double acc = 0.0;                    // per-thread cost partial over all time steps
  // prefetch registers for step t
  double kr[JR], xbr[JR], ubk = 0.0, kpk = 0.0;
  auto prefetch = [&](int t) __attribute__((always_inline)) {
    if (urole) {
      const double* Kr = v.K + ((size_t)t * m + uk) * n;
      const double* xbt = v.X + (size_t)t * n;
#pragma unroll
      for (int q = 0; q < JR; ++q) {
        const int j = ul + 16 * q;
        kr[q] = (j < n) ? Kr[j] : 0.0;
        xbr[q] = (j < n) ? xbt[j] : 0.0;
      }
      if (ul == 0) { ubk = v.U[(size_t)t * m + uk]; kpk = v.kap[(size_t)t * m + uk]; }
    }
  };
  prefetch(0);
  __syncthreads();
  for (int t = 0; t < N - 1; ++t) {
    // u_t = u_bar_t - eps*kappa_t - K_t (x_t - x_bar_t)   (ilqr.py:313)
    if (urole) {
      double p = 0.0;
#pragma unroll
      for (int q = 0; q < JR; ++q) {
        const int j = ul + 16 * q;
        if (j < n) p += kr[q] * (xs[j] - xbr[q]);
      }
      p += __shfl_xor(p, 8, 16);
      p += __shfl_xor(p, 4, 16);
      p += __shfl_xor(p, 2, 16);
      p += __shfl_xor(p, 1, 16);
      if (ul == 0) us[uk] = (ubk - eps * kpk) - p;
    }
    if (t + 1 < N - 1) prefetch(t + 1);          // lands while the dynamics run
    lds_barrier();
    // dynamics: one lane per degree of freedom (ilqr.py:316); cost rows on the other waves (:325)
    double qn_ = 0.0, vn_ = 0.0;
    if (tid < M::nq) {
      M::template dof<double>(tid, xs, us, qn_, vn_, a.params, a.dt);
    } else if (qrole) {
      const int i = tid - 64;
      double r = 0.0;
#pragma unroll
      for (int j = 0; j < n; ++j) r += qrow[j] * (xs[j] - xnom[j]);
      acc += (xs[i] - xnom[i]) * r;
    } else if (rrole) {
      const int k = tid - 128;
      double r = 0.0;
#pragma unroll
      for (int j = 0; j < m; ++j) r += rrow[j] * us[j];
      acc += us[k] * r;
      v.Un[(size_t)t * m + k] = us[k];
    }
    lds_barrier();
    if (tid < M::nq) {
      xs[tid] = qn_; xs[M::nq + tid] = vn_;
      v.Xn[(size_t)(t + 1) * n + tid] = qn_;
      v.Xn[(size_t)(t + 1) * n + M::nq + tid] = vn_;
    }
    lds_barrier();
  }
  if (qrole) {                         // terminal cost (ilqr.py:327)
    const int i = tid - 64;
    const double* Qf = lds + Ly::oQf;
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

#ifdef MI_PROF_BACKWARD
#define BP_TICK(k) do { const long long c_ = clock64(); if (bp_acc) bp_acc[k] += c_ - bp_last; bp_last = c_; } while (0)
#else
#define BP_TICK(k) do {} while (0)
#endif

typedef double d4_t __attribute__((ext_vector_type(4)));

// One 16x16 output tile on the matrix core: acc += sum over KSTEPS of A(16x4) B(4x16) with
// v_mfma_f64_16x16x4_f64.  Lane l supplies A[r = l&15][k0 + (l>>4)] and B[k0 + (l>>4)][c = l&15];
// it receives D[(l>>4) + 4*reg][l&15], reg = 0..3 (layout verified by tools/ubench/mfma64.hip).
//   a_ptr: address of A[r][0]-equivalent for this lane, a_ks: stride between k-steps (4 k's)
template <int KSTEPS>
struct TileOps {
  double av[KSTEPS], bv[KSTEPS];
  __device__ __forceinline__ void load(const double* a_ptr, int a_kstride, const double* b_ptr, int b_kstride) {
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) { av[ks] = a_ptr[ks * a_kstride]; bv[ks] = b_ptr[ks * b_kstride]; }
  }
  __device__ __forceinline__ d4_t run(d4_t acc) const {
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks], bv[ks], acc, 0, 0, 0);
    return acc;
  }
};

// value of lane LANE of this lane's 16-lane row (DPP row_share), for a double
template <int LANE>
__device__ __forceinline__ double row_share(double v) {
  union { double d; int i[2]; } u, r;
  u.d = v;
  r.i[0] = __builtin_amdgcn_update_dpp(0, u.i[0], 0x150 + LANE, 0xF, 0xF, false);
  r.i[1] = __builtin_amdgcn_update_dpp(0, u.i[1], 0x150 + LANE, 0xF, 0xF, false);
  return r.d;
}

// Right-looking LDL^T of an m x m matrix held one ROW PER LANE (lane i of a 16-lane row holds
// A[i][0..m-1]); pivots and column entries travel by DPP row_share, no LDS, no barriers.
// On exit lane i holds L[i][k] in a[k] for k < i, and every lane holds 1/D[k] in dinv[k].
template <int m, int K, int J>
struct LdlInner {
  static __device__ __forceinline__ void run(double (&a)[m], double lik) {
    const double ajk = row_share<J>(a[K]);          // A[J][K] before scaling = D[K] * L[J][K]
    a[J] = fma(-lik, ajk, a[J]);
    LdlInner<m, K, J + 1>::run(a, lik);
  }
};
template <int m, int K>
struct LdlInner<m, K, m> {
  static __device__ __forceinline__ void run(double (&)[m], double) {}
};
template <int m, int K>
struct LdlOuter {
  static __device__ __forceinline__ void run(double (&a)[m], double (&dinv)[m]) {
    const double d = row_share<K>(a[K]);
    const double inv = fast_rcp(d);
    dinv[K] = inv;
    const double lik = a[K] * inv;
    LdlInner<m, K, K + 1>::run(a, lik);
    a[K] = lik;
    LdlOuter<m, K + 1>::run(a, dinv);
  }
};
template <int m>
struct LdlOuter<m, m> {
  static __device__ __forceinline__ void run(double (&)[m], double (&)[m]) {}
};

// Backward Riccati pass (ilqr.py:623-667), cost expansion (:161-206) fused.
//
// Per time step, with F = [fx | fu] (n x (n+m)):
//     T1 = Vxx F                      (n x (n+m))      9 tiles x 9 k-steps
//     H  = F^T [T1 | Vx]              ((n+m) x (n+m+1)) = [[Qxx-lxx, . ],[Qux, Quu-luu]] and F^T Vx
//     Vxx' = Qxx - Qux^T K            (n x n)          9 tiles x 3 k-steps
// run as 16x16 tiles of v_mfma_f64_16x16x4_f64, 2-3 tiles per wave.  On gfx950 the fp64
// matrix rate equals the fp64 VALU rate (65 cycles per 16x16x4 = 15.7 FMA/clk/SIMD), so
// the point of the matrix core here is OPERAND DELIVERY: two 8-byte LDS reads per lane
// feed 1024 FMAs, where a VALU formulation needs a (broadcast) LDS read per 1-2 FMAs and
// is LDS-issue-bound at one wave per SIMD (tools/ubench/t1.hip: 4.5-10.7k cycles for T1
// alone vs ~1.8k here).  Quu is factorized ONCE per step (LDL^T) by 16 lanes with DPP
// row broadcasts; n+1 threads then substitute one right-hand side each (columns of Qux, Qu).
template <class M>
__device__ inline void large_backward(const LView<M::n, M::m>& v, double* lds, long long* bp_acc = nullptr) {
  constexpr int n = M::n, m = M::m, nm = n + m;
  using Ly = LLay<n, m>;
  constexpr int TS = Ly::TS, VS = Ly::VS, FS = Ly::NMP, NP = Ly::NP;
  constexpr int RT = NP / 16, CT = Ly::NMP / 16;       // row tiles of an n-row matrix, col tiles of an nm-col one
  static_assert(n % 4 == 0 && m % 4 == 0, "k-steps of 4");
  const int tid = threadIdx.x, N = v.N, wave = tid >> 6, lane = tid & 63;
  const int lr = lane & 15, lk = lane >> 4;
  const double* Q = lds + Ly::oQ;
  const double* R = lds + Ly::oR;
  const double* Qf = lds + Ly::oQf;
  const double* qn = lds + Ly::oQn;
  const double* qfn = lds + Ly::oQfn;
  double* Vxx = lds + Ly::oVxx;      // [NP][VS], rows >= n are zero
  double* Vx = lds + Ly::oVx;
  double* F = lds + Ly::oF;          // [n][FS]  = [fx | fu | 0-pad]
  double* T1 = lds + Ly::oT1;        // [n][TS]  = [Vxx F | . | Vx at column FS]; later rows 0..m-1 hold [K | kappa]
  double* H = lds + Ly::oH;          // [NMP][TS] = F^T T1, first-order terms in column FS
  double* Qc = lds + Ly::oQc;        // L (m x m, row-major, strictly-lower part valid) then 1/D (m)
  double* QT = lds + Ly::oQT;        // Q^T
  constexpr int CV = FS;             // column index of Vx / first-order terms
#ifdef MI_PROF_BACKWARD
  long long bp_last = clock64();
#endif

  // terminal: Vx = 2 Qf x_T - 2 x_nom^T Qf ; Vxx = 2 Qf   (ilqr.py:203-204, :638); pads = 0
  for (int e = tid; e < NP * VS; e += kLargeThreads) {
    const int i = e / VS, j = e - i * VS;
    Vxx[e] = (i < n && j < n) ? 2.0 * Qf[i * n + j] : 0.0;
  }
  for (int e = tid; e < n * FS; e += kLargeThreads) F[e] = 0.0;
  for (int e = tid; e < n * n; e += kLargeThreads) { const int i = e / n, j = e - i * n; QT[j * n + i] = Q[e]; }
  for (int e = tid; e < Ly::NMP * TS; e += kLargeThreads) H[e] = 0.0;
  if (tid < n) {
    const double* xT = v.X + (size_t)(N - 1) * n;
    double s = 0.0;
    for (int j = 0; j < n; ++j) s += (2.0 * Qf[tid * n + j]) * xT[j];
    Vx[tid] = s - qfn[tid];
  }
  __syncthreads();
  // cost gradients for ALL steps, off the recursion (ilqr.py:180-181): lx_t = 2Q x_bar_t - 2 x_nom^T Q,
  // lu_t = 2R u_bar_t.  Q^T is read so consecutive lanes hit consecutive banks.
  double* Lxu = lds + Ly::doubles;
  for (int idx = tid; idx < (N - 1) * nm; idx += kLargeThreads) {
    const int tt = idx / nm, pp = idx - tt * nm;
    double s_;
    if (pp < n) {
      const double* xg = v.X + (size_t)tt * n;
      s_ = -qn[pp];
      for (int j = 0; j < n; ++j) s_ += (2.0 * QT[j * n + pp]) * xg[j];
    } else {
      const double* ug = v.U + (size_t)tt * m;
      s_ = 0.0;
      for (int j = 0; j < m; ++j) s_ += (2.0 * R[(pp - n) * m + j]) * ug[j];
    }
    Lxu[idx] = s_;
  }
  __syncthreads();
  // prefetch registers: elements tid + 256*r of the contiguous fx_t (n*n) and fu_t (n*m) blocks
  constexpr int NFX = (n * n + kLargeThreads - 1) / kLargeThreads, NFU = (n * m + kLargeThreads - 1) / kLargeThreads;
  double frx[NFX], fru[NFU];
  const int fx_i0 = tid / n, fx_j0 = tid - fx_i0 * n, fu_i0 = tid / m, fu_k0 = tid - fu_i0 * m;
  auto fetch = [&](int t) __attribute__((always_inline)) {
    const double* fxg = v.Fx + (size_t)t * n * n;
    const double* fug = v.Fu + (size_t)t * n * m;
#pragma unroll
    for (int r = 0; r < NFX; ++r) { const int e = tid + kLargeThreads * r; frx[r] = fxg[e < n * n ? e : n * n - 1]; }
#pragma unroll
    for (int r = 0; r < NFU; ++r) { const int e = tid + kLargeThreads * r; fru[r] = fug[e < n * m ? e : n * m - 1]; }
  };
  auto publish = [&]() __attribute__((always_inline)) {
    int i = fx_i0, j = fx_j0;
#pragma unroll
    for (int r = 0; r < NFX; ++r) {
      if (tid + kLargeThreads * r < n * n) F[i * FS + j] = frx[r];
      i += kLargeThreads / n; j += kLargeThreads % n;
      if (j >= n) { j -= n; i += 1; }
    }
    i = fu_i0; j = fu_k0;
#pragma unroll
    for (int r = 0; r < NFU; ++r) {
      if (tid + kLargeThreads * r < n * m) F[i * FS + n + j] = fru[r];
      i += kLargeThreads / m; j += kLargeThreads % m;
      if (j >= m) { j -= m; i += 1; }
    }
  };
  fetch(N - 2);
  publish();
  __syncthreads();

  for (int t = N - 2; t >= 0; --t) {
    if (t > 0) fetch(t - 1);                     // next step's operands: in flight during this step
    BP_TICK(0);
    // ---- T1 = Vxx F : RT x CT tiles, K = n.  Operands of ALL of this wave's tiles are loaded
    //      before the first MFMA (sched_barrier) so LDS latency is paid once, not per k-step.
    //      Wave w < CT owns column tile w and sweeps the RT row tiles, so every tile offset is a
    //      compile-time immediate on top of one per-lane base address (no address registers kept
    //      live across the time loop); the spare wave does the Vx column.
    if (wave < CT) {
      const double* a_base = Vxx + lr * VS + lk;
      const double* b_base = F + lk * FS + 16 * wave + lr;
      double* d_base = T1 + lk * TS + 16 * wave + lr;
      TileOps<n / 4> ops[RT];
#pragma unroll
      for (int q = 0; q < RT; ++q) ops[q].load(a_base + 16 * q * VS, 4, b_base, 4 * FS);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < RT; ++q) {
        d4_t acc = {0.0, 0.0, 0.0, 0.0};
        acc = ops[q].run(acc);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int ib = 16 * q + 4 * reg;                 // rows ib + lk, lk = 0..3
          if (ib + 3 < n) d_base[ib * TS] = acc[reg];
          else if (ib < n) { if (ib + lk < n) d_base[ib * TS] = acc[reg]; }
        }
      }
    } else if (lane < n) {
      T1[lane * TS + CV] = Vx[lane];
    }
    lds_barrier();
    BP_TICK(1);
    // ---- H = F^T T1 : CT x CT tiles, K = n;  H[:, CV] = F^T Vx
    if (wave < CT) {
      const double* a_base = F + lk * FS + lr;                       // A = F^T: A[p][k] = F[k][p]
      const double* b_base = T1 + lk * TS + 16 * wave + lr;
      double* d_base = H + lk * TS + 16 * wave + lr;
      TileOps<n / 4> ops[CT];
#pragma unroll
      for (int q = 0; q < CT; ++q) ops[q].load(a_base + 16 * q, 4 * FS, b_base, 4 * TS);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < CT; ++q) {
        d4_t acc = {0.0, 0.0, 0.0, 0.0};
        acc = ops[q].run(acc);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) d_base[(16 * q + 4 * reg) * TS] = acc[reg];
      }
    } else if (lane < nm) {                                // spare wave: H[:, CV] = l_{x,u} + F^T Vx  (:651-652)
      double s = Lxu[t * nm + lane];                       // lx_t / lu_t (precomputed for all t)
#pragma unroll 6
      for (int k = 0; k < n; ++k) s += F[k * FS + lane] * Vx[k];
      H[lane * TS + CV] = s;
    }
    lds_barrier();
    BP_TICK(2);
    // ---- wave 1 factorizes Quu = 2R + fu^T Vxx fu (:654) = L D L^T, one row per lane
    if (false) {
    } else if (wave == 1) {
      static_assert(m <= 16, "one Quu row per lane of a 16-lane DPP row");
      const int i = lane < m ? lane : m - 1;               // lanes >= m shadow the last row (harmless)
      double arow[m], dinv[m];
#pragma unroll
      for (int j = 0; j < m; ++j) arow[j] = 2.0 * R[i * m + j] + H[(n + i) * TS + n + j];
      LdlOuter<m, 0>::run(arow, dinv);
      if (lane < m) {
#pragma unroll
        for (int k = 0; k < m; ++k) Qc[lane * m + k] = arow[k];      // L[lane][k] for k < lane
      }
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < m; ++k) Qc[m * m + k] = dinv[k];
      }
    }
    lds_barrier();
    BP_TICK(5);
    // ---- Y = Quu^{-1} [Qux | Qu] (:655-660): one right-hand side per thread, forward/back substitution
    if (tid <= n) {
      double y[m], dinv[m], Lr[m][m];
      const int rhs = tid < n ? tid : CV;
#pragma unroll
      for (int i = 0; i < m; ++i) { y[i] = H[(n + i) * TS + rhs]; dinv[i] = Qc[m * m + i]; }
#pragma unroll
      for (int i = 1; i < m; ++i)
#pragma unroll
        for (int k = 0; k < i; ++k) Lr[i][k] = Qc[i * m + k];
      __builtin_amdgcn_sched_barrier(0);                   // all (broadcast) LDS reads in flight before the FMA chains
#pragma unroll
      for (int i = 0; i < m; ++i)
#pragma unroll
        for (int k = 0; k < i; ++k) y[i] -= Lr[i][k] * y[k];
#pragma unroll
      for (int i = 0; i < m; ++i) y[i] *= dinv[i];
#pragma unroll
      for (int i = m - 1; i >= 0; --i)
#pragma unroll
        for (int k = i + 1; k < m; ++k) y[i] -= Lr[k][i] * y[k];
      if (tid < n) {
#pragma unroll
        for (int i = 0; i < m; ++i) T1[i * TS + tid] = y[i];       // K_t[:, tid] (:660), stored to HBM below
      } else {
        double dv = 0.0;
#pragma unroll
        for (int i = 0; i < m; ++i) {
          v.kap[(size_t)t * m + i] = y[i];                        // kappa_t (:659)
          T1[i * TS + CV] = y[i];
          dv += H[(n + i) * TS + CV] * y[i];                      // Qu^T Quu^{-1} Qu (:663)
        }
        v.dV[t] = dv;
      }
    }
    lds_barrier();
    BP_TICK(3);
    // ---- Vxx = Qxx - Qux^T K (RT x RT tiles, K = m; :667) ; Vx = Qx - Qux^T kappa (:666)
    if (wave < RT) {
      const double* a_base = H + (n + lk) * TS + lr;                 // A = Qux^T: A[i][a] = H[n+a][i]
      const double* b_base = T1 + lk * TS + 16 * wave + lr;          // B = K
      const double* c_base = H + lk * TS + 16 * wave + lr;           // C = Qxx - lxx
      const double* q_base = Q + lk * n + 16 * wave + lr;
      double* d_base = Vxx + lk * VS + 16 * wave + lr;
      const bool col_ok = 16 * wave + lr < n;
      TileOps<m / 4> ops[RT];
      d4_t accs[RT];
#pragma unroll
      for (int q = 0; q < RT; ++q) {
        ops[q].load(a_base + 16 * q, 4 * TS, b_base, 4 * TS);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int ib = 16 * q + 4 * reg;
          const bool ok = col_ok && (ib + 3 < n || (ib < n && ib + lk < n));
          accs[q][reg] = ok ? c_base[ib * TS] + 2.0 * q_base[ib * n] : 0.0;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < RT; ++q) {
#pragma unroll
        for (int ks = 0; ks < m / 4; ++ks) ops[q].av[ks] = -ops[q].av[ks];
        const d4_t acc = ops[q].run(accs[q]);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int ib = 16 * q + 4 * reg;
          const bool ok = col_ok && (ib + 3 < n || (ib < n && ib + lk < n));
          if (ok) d_base[ib * VS] = acc[reg];
        }
      }
    }
    if (wave == 3) {                              // K_t (m x n, contiguous in HBM): coalesced store from its LDS copy
      double* Kg = v.K + (size_t)t * m * n;
      for (int e = lane; e < m * n; e += 64) { const int i = e / n, j = e - i * n; Kg[e] = T1[i * TS + j]; }
    }
    if (wave == 3 && lane < n) {
      static_assert(RT <= 3 && CT <= 3, "wave 3 is the spare wave");
      double s = H[lane * TS + CV];
#pragma unroll
      for (int a_ = 0; a_ < m; ++a_) s -= H[(n + a_) * TS + lane] * T1[a_ * TS + CV];
      Vx[lane] = s;
    }
    if (t > 0) publish();                        // F/xb are free after the H phase
    lds_barrier();
    BP_TICK(4);
  }
}

template <class M, int JAC, int MODE>
__global__ void __launch_bounds__(kLargeThreads) ilqr_large_kernel(const KArgs a) {
  constexpr int n = M::n, m = M::m;
  using Ly = LLay<n, m>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* lds = reinterpret_cast<double*>(smem);
  int* ilds = reinterpret_cast<int*>(lds + Ly::doubles + (size_t)a.N * (n + m));
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
    long long bpa[6] = {0, 0, 0, 0, 0, 0};
    if (MODE == MODE_SOLVE) { large_backward<M>(v, lds, bpa); __syncthreads(); }
    if (tid == 0 && iters == 0) { for (int q_ = 0; q_ < 4; ++q_) a.prof[4 * b + q_] = bpa[q_]; a.hist[(size_t)b * a.hist_cap * 4 + 4 * (a.hist_cap - 1)] = (double)bpa[4]; a.hist[(size_t)b * a.hist_cap * 4 + 4 * (a.hist_cap - 1) + 1] = (double)bpa[5]; }
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
