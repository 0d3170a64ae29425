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
//     the reference's sequential loop accepts.  The eps = 1 trial is
//     attempted on its own first, so the common case costs ONE rollout.
//   * linearization (ilqr.py:380-415, 233-272): (key-point, column) pairs are
//     spread over the lanes; central finite differences (or one-directional
//     forward-mode duals) replace Drake AutoDiff.
//   * key-point selection / interpolation (ilqr.py:417-621): ballots + per-lane
//     segments.
//   * n = 2: TIME itself.  The rollout of a trial is Newton's method on the whole
//     trajectory (rollout_newton: every lane owns four consecutive steps, the
//     linearized recurrence is a prefix scan of affine maps over the lanes) and the
//     backward Riccati pass (ilqr.py:623-667 with :161-206 fused in) is an
//     associative scan of its second-order elements (backward_scan); both scans move
//     their operands with DPP row shifts / row broadcasts.
//   * n = 3..4: the backward pass stays sequential in t, each step two fp64 MFMAs
//     (backward_mfma); the rollout is sequential, evaluated wave-uniformly.
// Time is the fastest LDS axis (the reference's own layout), so
// HBM<->LDS staging is a linear, fully coalesced copy per array.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/mi_ilqr.h"
#include "fastmath.hpp"
#include "keypoints.hpp"
#include "models.hpp"

namespace mi {

enum KernelMode { MODE_SOLVE = 0, MODE_ROLLOUT = 1, MODE_FORWARD = 2, MODE_LINEARIZE = 3, MODE_BACKWARD = 4, MODE_MPC = 5 };

constexpr int kMaxStateDim = 40;   // largest model state (Synth36: 36), for by-value kernel arguments

// Per-solve aggregate over the batch, written into pinned, device-mapped host memory (one small host
// read after a blocking solve instead of four D2H copies).
struct DevStats {
  long long total_iters, total_ls;
  int n_conv, n_max, n_fail, max_iters_seen, best_index, n_internal, n_not_pd, pad_;
  double best_cost;
};

constexpr int kSyncWords = MI_ILQR_CLUSTER_WORDS;      // (= 40) 64-bit handshake words per problem (KArgs::cluster_sync): 8 + the costs of 4 x 7 line-search candidates
struct KArgs {
  // persistent per-problem solver state, reference layout with a leading batch axis
  double *x_bar, *u_bar, *K, *kappa, *dV, *fx, *fu;
  const double* x0;        // (B,n)
  const double* u_guess;   // (B,m,N-1) pending SetInitialGuess input (used when u_pending)
  double* cost;            // (B,)
  double* hist;            // (B,hist_cap,4)
  double* iter_cyc;        // (B,hist_cap,4) per-iteration stopwatches: line search, linearization, backward pass, whole iteration (cycles)
  double *x_trial, *u_trial, *trial_cost;   // stage outputs
  const double* stage_in;  // (B,) eps (ROLLOUT) or L_last (FORWARD)
  const double* costmat;   // Q[n*n] R[m*m] Qf[n*n] x_nom[n]
  int32_t *iters, *status, *ls_trials, *kp_count, *kp_list;
  long long* prof;         // (B,4) shader-clock cycles: line search, linearization, backward pass, whole solve
  double params[MI_ILQR_MAX_PARAMS];
  double dt, delta, beta, gamma, jerk_thr, err_thr, fd_h;
  int32_t N, B, kp_method, minN, maxN, max_iters, hist_cap;
  int32_t n_store;    // line-search candidates whose trajectories are kept in LDS (>= 1)
  int32_t cold;       // 1: persistent state is all-zero, do not read it
  int32_t u_pending;  // 1: take u_bar from u_guess
  // MODE_MPC: receding-horizon loop kept on the device (acrobot.py:145-155, mini_cheetah.py:190-201)
  int32_t mpc_resolves, mpc_replan;
  double mpc_target_step[kMaxStateDim];   // added to x_nom before every re-solve (mini_cheetah.py:151-156); zeros = fixed target
  double* mpc_log;             // (B, mpc_resolves, n+2): x0 of the re-solve | cost | iterations
  int32_t helpers;             // extra wavefronts per problem that share the linearization (0, 1 or 3), see ilqr_small_kernel
  int32_t seq_backward;        // 0: fastest backward pass; 1: sequential sweep (A/B measurements); 2: the reference's scalar recursion verbatim (asymmetric / indefinite costs)
  int32_t newton_rollout;      // 1: the eps = 1 trial is rolled out parallel in time (Newton on the trajectory) when it converges
  // MODE_SOLVE / MODE_MPC of the wave-per-problem kernels: the last workgroup to finish aggregates the
  // batch statistics itself (no second kernel per solve).  Null: the host launches stats_kernel.
  DevStats* stats_out;
  int32_t* done_counter;       // zero between launches
  // workgroup-per-problem kernels, MODE_SOLVE / MODE_MPC with every step a key-point: `cluster` workgroups per
  // problem - one leader that runs the solve and cluster-1 helpers that share its linearizations
  // (ilqr_large.hpp: cluster handshake).  cluster_sync: kSyncWords 64-bit words per problem, zero at launch.
  // cluster: bits 0-7 workgroups per problem, bits 8-9 their placement (0: consecutive blocks, a cluster spans XCDs; 1, 2: all on
  // one XCD), bit 10: early linearization (the helpers linearize the line search's first trial while it is being rolled out),
  // bit 11: candidate groups (mid-size kernels: the helpers roll out line-search candidates 4 .. beside the leader's four).
  int32_t cluster;
  unsigned long long* cluster_sync;
  // wave-per-problem kernels: optional RESULT SINK (mi_ilqr_set_result_sink) - device-visible, page-locked HOST arrays
  // that receive x_bar (B,n,N), u_bar (B,m,N-1) and the costs (B,) straight from the kernel's write-back, problem by
  // problem as each one finishes: the copy-out of a batch overlaps the launch's stragglers instead of following it.
  double *sink_x, *sink_u, *sink_cost;
  // lane-per-problem kernels with key-points (ilqr_batch.hpp, KP = true): 6 (N-1) x B ints, batch-minor - the lanes' key-point
  // lists, "derivative evaluated" flags and the two bin buffers of the iterative-error bisection
  int32_t* bm_scratch;
  // mid-size workgroup-per-problem kernels (ilqr_large.hpp: mid_rollout4): trial trajectories of the line-search candidates
  // rolled out beside the first, [3][B][N][n] and [3][B][N-1][m]
  double *x_spec, *u_spec;
  int spec_policy;
  int q_diag;                     // Q has no off-diagonal entry (every script of the reference): the sequential rollouts' stage cost skips the n (n - 1) products with zeros
  // workgroup-per-problem kernels, long horizons: the cost gradients [B][N-1][n+m] in HBM instead of LDS (ilqr_large.hpp)
  double* lxu;
  int pd_continue;                // mi_ilqr_desc.on_indefinite
  int cost_asym;                  // workgroup-per-problem kernels, n <= 32: Q, R or Qf is not symmetric (mi_ilqr_set_cost)
};

// threadIdx.x behind an empty asm, for the STAGES of a solve kernel (a rollout, a linearization, a backward pass): what a stage
// derives from its lane index is loop-invariant for the solve loop around the stages, the compiler hoists it out of that loop,
// and the hoisted values - dozens of lane-dependent addresses per stage - then live across every other stage and get spilled in
// whichever inner loop is tightest.  Opaque per call, they are formed at the top of the stage and die with it (measured on the
// workgroup-per-problem kernels, round 5: backward pass of the arm 6.2 k -> 5.5 k cycles per step, of the coupled arm 7.4 k -> 5.5 k;
// on the wave-per-problem kernels of this file it changes nothing - C2 43.19 M it/s either way - and they keep threadIdx.x).
__device__ __forceinline__ int stage_lane() {
  int t = threadIdx.x;
#ifndef MI_NO_STAGE_TID
  asm volatile("" : "+v"(t));
#endif
  return t;
}

__device__ __forceinline__ double bcast_lane0(double v) {
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readfirstlane(u.i[0]);
  u.i[1] = __builtin_amdgcn_readfirstlane(u.i[1]);
  return u.d;
}

// One wavefront owns a problem's LDS outside the linearization: ordering its own LDS traffic needs
// no s_barrier (LDS executes a wave's operations in order), only that the compiler keeps the order
// and waits for completion.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Rendezvous of the main wave with its helper waves (LDS traffic only).
__device__ __forceinline__ void team_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// ---------------------------------------------------------------------------
// LDS layout: array-of-records, ONE record per time step, so that everything a
// sequential step touches is reachable from a single per-step pointer with
// compile-time (immediate) offsets — no per-access address arithmetic in the
// latency-critical loops, and adjacent fields fuse into ds_read2/ds_read_b128.
//   G_t  nominal trajectory + gains : x_bar[n] | K[m][n] | u_bar[m] | kappa[m], dV
//   T_t  trial trajectory           : x[n] | u[m]
//   J_t  dynamics partials          : fx[n][n] | fu[n][m]
// (HBM keeps the reference's time-last layout; the staging copy transposes.)
// Each array has one pad record before index 0 and after the last index so the
// one-step-ahead software prefetch never needs a clamp.
// ---------------------------------------------------------------------------
template <int n, int m>
struct Lay {
  static constexpr int even(int v) { return (v + 1) & ~1; }
  static constexpr int XB = 0;
  static constexpr int KK = even(n);
  static constexpr int UB = KK + even(m * n);
  static constexpr int KAP = UB + even(m);
  static constexpr int DV = KAP + m;
  // n = 2 (the passes over time are lane-chunked there: Riccati scan, Newton rollout): record strides
  // are ODD numbers of doubles, so lanes reading consecutive records hit 32 different bank pairs and
  // lanes owning chunks of 2..4 consecutive records conflict at most 4-way instead of 32-way.  Larger
  // n keeps 16-byte aligned records (b128 loads in the wave-uniform sweeps matter more there).
  static constexpr int pad(int v) { return n <= 2 ? (v | 1) : even(v); }
  static constexpr int GS = pad(DV + 1);
  static constexpr int XN = 0, UN = even(n), TS = pad(UN + m);
  static constexpr int FX = 0, FU = even(n * n), JS = pad(FU + n * m);
  static constexpr int DUMP_DOUBLES = 64 * 2 + (GS > JS ? GS : JS);   // 16 B per lane + one record of slack
  // copy of the cost constants (Consts<M>) for code that runs outside the kernel function (outlined passes)
  static constexpr int CST_DOUBLES = (n >= 3) ? even(2 * n * n + m * m + 3 * n) : 0;
};

struct WS {
  double *G, *T, *J;       // point at record index 0 (pad record lives at index -1)
  double* dump;            // per-lane sink for predicated-off stores (lane*16 B)
  double* cst;             // Consts<M> image (n >= 3)
  int *kp, *aux, *need, *binA, *binB;
  int N;
  int n_store, t_stride;   // T holds n_store trajectories, t_stride doubles apart
};

template <int n, int m>
__host__ __device__ constexpr size_t ws_bytes(int N, int n_store = 1) {
  using L = Lay<n, m>;
  return ((size_t)(N + 2) * L::GS + (size_t)n_store * (N + 2) * L::TS + (size_t)(N + 2) * L::JS + L::DUMP_DOUBLES + L::CST_DOUBLES) * 8 +
         (size_t)7 * N * 4 + 16;
}

template <int n, int m>
__device__ inline WS carve(char* base, int N, int n_store) {
  using L = Lay<n, m>;
  WS w;
  w.N = N;
  double* p = reinterpret_cast<double*>(base);
  w.G = p + L::GS; p += (size_t)(N + 2) * L::GS;
  w.T = p + L::TS; p += (size_t)n_store * (N + 2) * L::TS;
  w.n_store = n_store; w.t_stride = (N + 2) * L::TS;
  w.J = p + L::JS; p += (size_t)(N + 2) * L::JS;
  w.dump = p; p += L::DUMP_DOUBLES;
  w.cst = p; p += L::CST_DOUBLES;
  int* q = reinterpret_cast<int*>(p);
  w.kp = q; q += N;
  w.aux = q; q += N;
  w.need = q; q += N;
  w.binA = q; q += 2 * N;
  w.binB = q;
  return w;
}

// Result write-back store that goes THROUGH the L2 (device-scope relaxed store = sc1): most waves of a launch
// finish long before its slowest problem, and lines they leave dirty would all be written back by the
// end-of-kernel release, i.e. inside the gap before the next dispatch.
__device__ __forceinline__ void wt_store(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// HBM (rows,len) time-last  ->  LDS records rec[t*RS + off + row]   (and back)
__device__ inline void stage_in(double* recs, int RS, int off, const double* src, int rows, int len, bool zero) {
  for (int r = 0; r < rows; ++r) {
    double* d = recs + off + r;
    const double* s = src + (size_t)r * len;
    if (zero) { for (int t = threadIdx.x; t < len; t += 64) d[t * RS] = 0.0; }
    else {
#pragma unroll 4
      for (int t = threadIdx.x; t < len; t += 64) d[t * RS] = s[t];
    }
  }
}
__device__ inline void stage_out(double* dst, const double* recs, int RS, int off, int rows, int len) {
  for (int r = 0; r < rows; ++r) {
    const double* s = recs + off + r;
    double* d = dst + (size_t)r * len;
#pragma unroll 4
    for (int t = threadIdx.x; t < len; t += 64) d[t] = s[t * RS];
  }
}

// The same model with the kernels' backward pass run as the time-parallel scan.  For n = 3..4 the scan
// pays from two steps per lane on (N > 128) and its register appetite must not touch the kernels of
// short horizons, so the host picks this instantiation by horizon (mi_ilqr.hip: launch_jac).
template <class M>
struct LongHorizon : M { static constexpr bool kScanBackward = true; };
template <class M, class = void>
struct UsesScanBackward : std::false_type {};
template <class M>
struct UsesScanBackward<M, std::void_t<decltype(M::kScanBackward)>> : std::bool_constant<M::kScanBackward> {};

// The same model with the kernels' backward pass run as the reference's scalar recursion verbatim
// (cost matrices the MFMA / scan forms do not cover: asymmetric or indefinite Q, Qf, R).
template <class M>
struct ExactCost : M { static constexpr bool kExactBackward = true; };
template <class M, class = void>
struct UsesExactBackward : std::false_type {};
template <class M>
struct UsesExactBackward<M, std::void_t<decltype(M::kExactBackward)>> : std::bool_constant<M::kExactBackward> {};

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
  // LDS image: Q | Qf | R | xnom | qn | qfn
  __device__ inline void to_lds(double* d) const {
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
      for (int i = 0; i < n; ++i)
#pragma unroll
        for (int j = 0; j < n; ++j) { d[i * n + j] = Q[i][j]; d[n * n + i * n + j] = Qf[i][j]; }
#pragma unroll
      for (int i = 0; i < m; ++i)
#pragma unroll
        for (int j = 0; j < m; ++j) d[2 * n * n + i * m + j] = R[i][j];
#pragma unroll
      for (int i = 0; i < n; ++i) { d[2 * n * n + m * m + i] = xnom[i]; d[2 * n * n + m * m + n + i] = qn[i]; d[2 * n * n + m * m + 2 * n + i] = qfn[i]; }
    }
  }
  __device__ inline void from_lds(const double* d) {
#pragma unroll
    for (int i = 0; i < n; ++i)
#pragma unroll
      for (int j = 0; j < n; ++j) { Q[i][j] = d[i * n + j]; Qf[i][j] = d[n * n + i * n + j]; }
#pragma unroll
    for (int i = 0; i < m; ++i)
#pragma unroll
      for (int j = 0; j < m; ++j) R[i][j] = d[2 * n * n + i * m + j];
#pragma unroll
    for (int i = 0; i < n; ++i) { xnom[i] = d[2 * n * n + m * m + i]; qn[i] = d[2 * n * n + m * m + n + i]; qfn[i] = d[2 * n * n + m * m + 2 * n + i]; }
  }
};

// ---------------------------------------------------------------------------
// One line-search trial (ilqr.py:306-327) for this lane's eps.  G records are
// read at wave-uniform addresses (LDS broadcast), one step ahead of use, into
// two alternating register sets (manual 2x unroll: no register rotation moves).
// Lane 0 stores its trajectory into the T records; the other lanes' stores go
// to a per-lane dump slot so the loop carries no exec-mask branches.
// ---------------------------------------------------------------------------
// q_diag (wave-uniform: a kernel argument): Q is diagonal - the products with its zeros are left out.  Same bits: a row sum then is
// fma(0, dx_j, s) = s for every j != i (finite states; a diverged trial's cost is NaN or inf either way, and rejected either way).
// 12 of the 146 instructions of a cart-pole + wall rollout step - and, MEASURED in the fused kernel (same-box A/B,
// profiles/r06_c4_ab.txt), a line search that is 26 % LONGER (178.6 k -> 224.9 k cycles per iteration): the second arm of the branch
// lives in the same loop, and its registers push the loop's values into the accumulation file.  OFF by default (MI_STAGE_COST_DIAG).
template <class M>
__device__ __forceinline__ double stage_cost(const Consts<M>& c, const double (&x)[M::n], const double (&u)[M::m], bool q_diag = false) {
  constexpr int n = M::n, m = M::m;
  double dx[n];
#pragma unroll
  for (int i = 0; i < n; ++i) dx[i] = x[i] - c.xnom[i];
  double q = 0.0;
#ifndef MI_STAGE_COST_DIAG
#define MI_STAGE_COST_DIAG 0
#endif
  if (MI_STAGE_COST_DIAG && n >= 4 && q_diag) {
#pragma unroll
    for (int i = 0; i < n; ++i) {
      double s = 0.0;
      s += c.Q[i][i] * dx[i];                                 // (the same fma(Q_ii, dx_i, 0) the full row sum ends up with)
      q += dx[i] * s;
    }
  } else {
#pragma unroll
    for (int i = 0; i < n; ++i) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < n; ++j) s += c.Q[i][j] * dx[j];
      q += dx[i] * s;
    }
  }
  double ru = 0.0;
#pragma unroll
  for (int i = 0; i < m; ++i) {
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < m; ++j) s += c.R[i][j] * u[j];
    ru += u[i] * s;
  }
  return q + ru;
}

template <class M>
__device__ __forceinline__ double terminal_cost(const Consts<M>& c, const double (&x)[M::n]) {
  constexpr int n = M::n;
  double dx[n];
#pragma unroll
  for (int i = 0; i < n; ++i) dx[i] = x[i] - c.xnom[i];
  double q = 0.0;
#pragma unroll
  for (int i = 0; i < n; ++i) {
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < n; ++j) s += c.Qf[i][j] * dx[j];
    q += dx[i] * s;
  }
  return q;
}

template <class M>
struct GRegs {
  double xb[M::n], K[M::m][M::n], ub[M::m], kap[M::m], dv;
  __device__ __forceinline__ void load(const double* g) {
    using L = Lay<M::n, M::m>;
#pragma unroll
    for (int i = 0; i < M::n; ++i) xb[i] = g[L::XB + i];
#pragma unroll
    for (int k = 0; k < M::m; ++k) {
#pragma unroll
      for (int j = 0; j < M::n; ++j) K[k][j] = g[L::KK + k * M::n + j];
      ub[k] = g[L::UB + k];
      kap[k] = g[L::KAP + k];
    }
    dv = g[L::DV];
  }
};

template <class M, class = void>
struct HasStepPool : std::false_type {};
template <class M>
struct HasStepPool<M, std::void_t<decltype(M::kHasStepPool)>> : std::bool_constant<M::kHasStepPool> {};
struct NoPool {};
template <class M, bool = HasStepPool<M>::value>
struct PoolOf { using type = NoPool; __device__ __forceinline__ static NoPool make() { return {}; } };
template <class M>
struct PoolOf<M, true> { using type = typename M::StepPool; __device__ __forceinline__ static type make() { return M::StepPool::in_vgprs(); } };

template <class M, bool COST>
__device__ __forceinline__ void rollout_step(const GRegs<M>& r, const Consts<M>& c, const KArgs& a, double eps,
                                             double ce, double (&x)[M::n], double& L, double& expd, double* tw,
                                             const typename PoolOf<M>::type& pool) {
  constexpr int n = M::n, m = M::m;
  using Ly = Lay<n, m>;
  // u_t = u_bar_t - eps*kappa_t - K_t (x_t - x_bar_t)          (ilqr.py:313)
  double u[m];
#pragma unroll
  for (int k = 0; k < m; ++k) {
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < n; ++j) acc += r.K[k][j] * (x[j] - r.xb[j]);
    u[k] = (r.ub[k] - eps * r.kap[k]) - acc;
  }
  double xnext[n];
  if constexpr (HasStepPool<M>::value) M::step_pooled(x, u, xnext, a.params, a.dt, pool);   // ilqr.py:316
  else M::template step<double>(x, u, xnext, a.params, a.dt);
  if (COST) {
    // stage cost (no 1/2 factor, ilqr.py:325) and expected improvement (:326)
    L += stage_cost<M>(c, x, u, a.q_diag != 0);
    expd += ce * r.dv;
  }
  // T_t.u = u_t ; T_{t+1}.x = x_{t+1}
#pragma unroll
  for (int k = 0; k < m; ++k) tw[Ly::UN + k] = u[k];
#pragma unroll
  for (int i = 0; i < n; ++i) tw[Ly::TS + Ly::XN + i] = xnext[i];
#pragma unroll
  for (int i = 0; i < n; ++i) x[i] = xnext[i];
}

// `slot` >= 0: this lane stores its trajectory into T buffer `slot`; < 0: stores are parked.
template <class M, bool COST = true>
__device__ inline void rollout(const WS& w, const Consts<M>& c, const KArgs& a, const double* x0r,
                               double eps, int slot, double& L_out, double& exp_out) {
  constexpr int n = M::n, m = M::m;
  using Ly = Lay<n, m>;
  const int N = w.N;
  double x[n];
#pragma unroll
  for (int i = 0; i < n; ++i) x[i] = x0r[i];
  // store==true (lane 0): walk the T records; otherwise park on this lane's dump slot
  // COST == false is the optimistic trial: every lane carries the SAME eps = 1 trajectory, so all
  // lanes store to lane 0's address (one LDS word per bank pass instead of 64 parked slots:
  // tools/ubench/rollstep.hip, 385 -> 289 cycles/step with four waves per CU)
  const bool store = slot >= 0 || !COST;
  double* tw = store ? (w.T + (slot >= 0 ? slot : 0) * w.t_stride) : (w.dump + 2 * threadIdx.x);
  const int tstep = store ? Ly::TS : 0;
#pragma unroll
  for (int i = 0; i < n; ++i) tw[Ly::XN + i] = x[i];
  double L = 0.0, expd = 0.0;
  const double ce = -eps * (1.0 - eps / 2.0);

  const double* g = w.G;
  GRegs<M> A, B;
  A.load(g);
  const typename PoolOf<M>::type pool = PoolOf<M>::make();   // the model's polynomial constants in VGPRs for the loop
  int t = 0;
  for (; t + 1 < N - 1; t += 2) {
    B.load(g + Ly::GS);
    __builtin_amdgcn_sched_barrier(0);      // keep the prefetch a full step ahead of its first use
    rollout_step<M, COST>(A, c, a, eps, ce, x, L, expd, tw, pool);
    tw += tstep;
    A.load(g + 2 * Ly::GS);                 // t+2 <= N-1: a real record (or the pad at N)
    __builtin_amdgcn_sched_barrier(0);
    rollout_step<M, COST>(B, c, a, eps, ce, x, L, expd, tw, pool);
    tw += tstep;
    g += 2 * Ly::GS;
  }
  if (t < N - 1) rollout_step<M, COST>(A, c, a, eps, ce, x, L, expd, tw, pool);
  if (COST) L += terminal_cost<M>(c, x);            // ilqr.py:327
  L_out = L;
  exp_out = expd;
}

// Sum over each 16-lane row, result in every lane of the row: four DPP row rotations (8, 4, 2, 1)
// instead of four ds_bpermute round trips through the LDS crossbar.
template <int ROT>
__device__ __forceinline__ double row_ror_f64(double v) {
  union { double d; int i[2]; } u, r;
  u.d = v;
  r.i[0] = __builtin_amdgcn_mov_dpp(u.i[0], 0x120 + ROT, 0xF, 0xF, true);
  r.i[1] = __builtin_amdgcn_mov_dpp(u.i[1], 0x120 + ROT, 0xF, 0xF, true);
  return r.d;
}
__device__ __forceinline__ double row16_sum(double p) {
  p += row_ror_f64<8>(p);
  p += row_ror_f64<4>(p);
  p += row_ror_f64<2>(p);
  p += row_ror_f64<1>(p);
  return p;
}

// Sum over the wave, result in every lane, fixed order: DPP row sums, then the four row totals
// through v_readlane (SGPRs) - no ds_bpermute round trips.
__device__ __forceinline__ double readlane_f64(double v, int srclane);
__device__ __forceinline__ double wave_sum(double v) {
  v = row16_sum(v);
  return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

// Total cost (ilqr.py:325,327) and expected improvement (:326) of the trajectory stored in
// the T records, evaluated time-parallel (one time step per lane, fixed-order wave reduction).
template <class M>
__device__ inline void traj_cost(const WS& w, const Consts<M>& c, double eps, double& L_out, double& exp_out) {   // T buffer 0
  constexpr int n = M::n, m = M::m;
  using Ly = Lay<n, m>;
  const int N = w.N;
  double acc = 0.0, dvs = 0.0;
  for (int t = threadIdx.x; t < N; t += 64) {
    const double* tr = w.T + t * Ly::TS;
    double x[n];
#pragma unroll
    for (int i = 0; i < n; ++i) x[i] = tr[Ly::XN + i];
    if (t < N - 1) {
      double u[m];
#pragma unroll
      for (int k = 0; k < m; ++k) u[k] = tr[Ly::UN + k];
      acc += stage_cost<M>(c, x, u);
      dvs += w.G[t * Ly::GS + Ly::DV];
    } else {
      acc += terminal_cost<M>(c, x);
    }
  }
  L_out = wave_sum(acc);
  exp_out = -eps * (1.0 - eps / 2.0) * wave_sum(dvs);
}

// ---------------------------------------------------------------------------
// The eps = 1 trial rolled out PARALLEL IN TIME (n = 2): Newton's method on the whole trajectory.
//
// The rollout x_{t+1} = g_t(x_t) = f(x_t, u_bar_t - kappa_t - K_t (x_t - x_bar_t)) (ilqr.py:313-316)
// is a nonlinear recurrence, 255 cycles per step when one wavefront walks it.  Given a guess X of the
// whole trajectory, every lane evaluates g_t and its state Jacobian G_t (one forward-mode Dual2
// evaluation) at its own few steps; the linearized recurrence x_{t+1} = g_t(X_t) + G_t (x_t - X_t)
// is an affine map composition, i.e. a prefix scan over the lanes (Kogge-Stone, ds_bpermute); the
// result is the next guess.  Started from the previous nominal trajectory, the iteration converges
// quadratically: 4 sweeps (5 early in a solve) bring the update below 1e-9, after which the error
// is at round-off (prototype against the sequential rollout: 1e-14 absolute, tools/... DESIGN.md).
// The last sweep's result IS the trial trajectory (its linearized update misses the exact step by the quadratic
// remainder of a correction below 1e-7: round-off); one exact model step per lane - its last - checks that a
// posteriori.  Not converged within the cap (cold starts, the first iteration of hard problems), or the check
// fails -> the caller falls back to the sequential rollout.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double lane_read_f64(double v, int src);

// One column of [fx | fu] at (x, u): central differences (the build's stand-in for AutoDiff,
// ilqr.py:233-272) or one forward-mode dual evaluation.
template <class M, int JAC>
__device__ __forceinline__ void jac_column(const double (&x)[M::n], const double (&u)[M::m], int col, const KArgs& a,
                                           double (&d)[M::n]) {
  constexpr int n = M::n, m = M::m;
  if (JAC == MI_JAC_FD_CENTRAL) {
    const double h = a.fd_h, inv2h = 1.0 / (2.0 * h);
    double xp[n], up[m], xm[n], um[m], fp[n], fm_[n];
#pragma unroll
    for (int i = 0; i < n; ++i) { xp[i] = (col == i) ? x[i] + h : x[i]; xm[i] = (col == i) ? x[i] - h : x[i]; }
#pragma unroll
    for (int k = 0; k < m; ++k) { up[k] = (col == n + k) ? u[k] + h : u[k]; um[k] = (col == n + k) ? u[k] - h : u[k]; }
    M::template step<double>(xp, up, fp, a.params, a.dt);
    M::template step<double>(xm, um, fm_, a.params, a.dt);
#pragma unroll
    for (int i = 0; i < n; ++i) d[i] = (fp[i] - fm_[i]) * inv2h;
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
}

#ifdef MI_PROF_NEWTON
__device__ double mi_dbg_vals[8];
#endif
struct Aff2 {                     // x -> G x + c
  double G[2][2], c[2];
};
// later (o) earlier: apply `e` first, then `l`
__device__ __forceinline__ void aff2_compose(Aff2& o, const Aff2& l, const Aff2& e) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    o.c[i] = fma(l.G[i][0], e.c[0], fma(l.G[i][1], e.c[1], l.c[i]));
#pragma unroll
    for (int j = 0; j < 2; ++j) o.G[i][j] = fma(l.G[i][0], e.G[0][j], l.G[i][1] * e.G[1][j]);
  }
}

// The same maps with G stored as G - I ("deviation form"): the identity is all zeros, which is what a
// DPP move delivers to a lane without a source (bound_ctrl) - so the Kogge-Stone prefix below needs
// neither selects nor the LDS crossbar.  later (o) earlier:
//   G' = l.G' + e.G' + l.G' e.G',   c = l.c + e.c + l.G' e.c
__device__ __forceinline__ void aff2_compose_dev(Aff2& o, const Aff2& l, const Aff2& e) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    o.c[i] = fma(l.G[i][0], e.c[0], fma(l.G[i][1], e.c[1], l.c[i] + e.c[i]));
#pragma unroll
    for (int j = 0; j < 2; ++j) o.G[i][j] = fma(l.G[i][0], e.G[0][j], fma(l.G[i][1], e.G[1][j], l.G[i][j] + e.G[i][j]));
  }
}
template <int CTRL, int ROWS>
__device__ __forceinline__ double dpp_f64_or_zero(double v) {
  union { double d; int i[2]; } u, r;
  u.d = v;
  if constexpr (ROWS == 0xF) {                                              // no source lane: 0 (bound_ctrl)
    r.i[0] = __builtin_amdgcn_mov_dpp(u.i[0], CTRL, 0xF, 0xF, true);
    r.i[1] = __builtin_amdgcn_mov_dpp(u.i[1], CTRL, 0xF, 0xF, true);
  } else {                                                                  // row not selected: 0 (old value)
    r.i[0] = __builtin_amdgcn_update_dpp(0, u.i[0], CTRL, ROWS, 0xF, true);
    r.i[1] = __builtin_amdgcn_update_dpp(0, u.i[1], CTRL, ROWS, 0xF, true);
  }
  return r.d;
}
template <int CTRL, int ROWS>
__device__ __forceinline__ void aff2_prefix_level(Aff2& P) {
  Aff2 f, t_;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    f.c[i] = dpp_f64_or_zero<CTRL, ROWS>(P.c[i]);
#pragma unroll
    for (int j = 0; j < 2; ++j) f.G[i][j] = dpp_f64_or_zero<CTRL, ROWS>(P.G[i][j]);
  }
  aff2_compose_dev(t_, P, f);
  P = t_;
}
// In: this lane's map (deviation form).  Out: the composition of the maps of lanes 0..lane.
__device__ __forceinline__ void aff2_prefix_dpp(Aff2& P) {
  aff2_prefix_level<0x111, 0xF>(P);     // row_shr:1
  aff2_prefix_level<0x112, 0xF>(P);     // row_shr:2
  aff2_prefix_level<0x114, 0xF>(P);     // row_shr:4
  aff2_prefix_level<0x118, 0xF>(P);     // row_shr:8   -> prefix inside every 16-lane row
  aff2_prefix_level<0x142, 0xA>(P);     // row_bcast:15 into rows 1 and 3
  aff2_prefix_level<0x143, 0xC>(P);     // row_bcast:31 into rows 2 and 3
}

// Outcome of the time-parallel rollout.  With `fuse` = 0 it only stores the trial trajectory in T
// (NEWTON_STORED; the caller evaluates the cost).  With fuse >= 1 the final pass also sums the cost
// from its registers (ilqr.py:325-327), applies the acceptance test (:330-331) and, if accepted,
// writes the trajectory straight into the nominal records (:375-376) - no T records, no separate
// cost and commit passes; fuse = 2 additionally differentiates the dynamics at every step it holds
// (:380-415 with every step a key-point), again from registers.
#ifndef MI_NEWTON_MAX_SWEEPS
#define MI_NEWTON_MAX_SWEEPS 7
#endif
enum { NEWTON_FAILED = 0, NEWTON_STORED = 1, NEWTON_REJECTED = 2, NEWTON_ACCEPTED = 3 };
// Models on which the time-parallel rollout's remainder was measured (models.hpp: kNewtonRollout - the built-in smooth
// n = 2 models); every other n = 2 model (plugins) takes the rollout too, under the stricter guard of dynamics_hold.
template <class M, class = void>
struct NewtonMeasured { static constexpr bool value = false; };
template <class M>
struct NewtonMeasured<M, decltype((void)M::kNewtonRollout)> { static constexpr bool value = M::kNewtonRollout; };

template <class M, int JAC, int CH>
__device__ inline int rollout_newton_impl(const WS& w, const Consts<M>& c, const KArgs& a, const double* x0r, double eps,
                                          int fuse, double L_last, double& L_out, bool coarse) {
  constexpr int n = 2, m = 1;
  static_assert(M::n == 2 && M::m == 1, "2-state closed loop");
  using Ly = Lay<n, m>;
  const int N = w.N, lane = threadIdx.x & 63, steps = N - 1;
  const int t0 = lane * CH;
  // nominal data and the initial guess (= the nominal trajectory) of this lane's steps
  double xb[CH][n], Kk[CH][n], dd[CH], X[CH][n];
  bool valid[CH];
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    const int t = t0 + k;
    valid[k] = t < steps;
    const double* g = w.G + (valid[k] ? t : 0) * Ly::GS;
#pragma unroll
    for (int i = 0; i < n; ++i) { xb[k][i] = g[Ly::XB + i]; Kk[k][i] = g[Ly::KK + i]; X[k][i] = (t == 0) ? x0r[i] : xb[k][i]; }
    dd[k] = g[Ly::UB] - eps * g[Ly::KAP];                   // u_bar - eps kappa (ilqr.py:313)
  }
  if (coarse) {
    // First iteration of a cold solve: there is no nominal trajectory to start from.  The guess is a
    // COARSE sequential rollout - one model step of CH dt per lane chunk, control held at its value at
    // the chunk start (ceil(steps / CH) dependent steps instead of `steps`) - refined inside every
    // chunk by the plain steps from the coarse chunk start.  Newton's sweeps then pull the whole
    // trajectory onto the fine recurrence; not converged -> sequential rollout as before.
    double xc[n] = {x0r[0], x0r[1]};
    const int nchunks = (steps + CH - 1) / CH;
    struct Rec { double xb[n], kk[n], ub, kap; };
    auto fetch = [&](Rec& r, int j) __attribute__((always_inline)) {
      const double* g = w.G + (j * CH) * Ly::GS;              // wave-uniform address: LDS broadcast
      r.xb[0] = g[Ly::XB + 0]; r.xb[1] = g[Ly::XB + 1]; r.kk[0] = g[Ly::KK + 0]; r.kk[1] = g[Ly::KK + 1];
      r.ub = g[Ly::UB]; r.kap = g[Ly::KAP];
    };
    Rec cur, nxt;
    fetch(cur, 0);
    for (int j = 0; j < nchunks; ++j) {
      fetch(nxt, j + 1 < nchunks ? j + 1 : j);                // one record ahead of its use
      __builtin_amdgcn_sched_barrier(0);
      if (lane == j) { X[0][0] = xc[0]; X[0][1] = xc[1]; }
      double u[m], xn[n];
      u[0] = (cur.ub - eps * cur.kap) - (cur.kk[0] * (xc[0] - cur.xb[0]) + cur.kk[1] * (xc[1] - cur.xb[1]));
      M::template step<double>(xc, u, xn, a.params, (double)CH * a.dt);
      xc[0] = xn[0]; xc[1] = xn[1];
      cur = nxt;
    }
    if (lane >= nchunks) { X[0][0] = xc[0]; X[0][1] = xc[1]; }
#pragma unroll
    for (int k = 0; k + 1 < CH; ++k) {
      double u[m], xn[n];
      u[0] = dd[k] - (Kk[k][0] * (X[k][0] - xb[k][0]) + Kk[k][1] * (X[k][1] - xb[k][1]));
      M::template step<double>(X[k], u, xn, a.params, a.dt);
      X[k + 1][0] = xn[0]; X[k + 1][1] = xn[1];
    }
  }
#ifndef MI_NEWTON_NO_PREDICTOR
  else
  // Predictor: the first guess is the trajectory the backward pass itself predicts, the linearized
  // closed loop  dx_{t+1} = (fx_t - fu_t K_t) dx_t - fu_t kappa_t  around the nominal one (the same
  // scan, 6 multiply-adds per step instead of a Dual2 model evaluation) - it stands in for the first
  // Newton sweep.
  {
    Aff2 loc[CH], agg;
    agg.G[0][0] = 0.0; agg.G[0][1] = 0.0; agg.G[1][0] = 0.0; agg.G[1][1] = 0.0; agg.c[0] = 0.0; agg.c[1] = 0.0;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int t = valid[k] ? t0 + k : 0;
      const double* jr = w.J + t * Ly::JS;
      const double kap = eps * w.G[t * Ly::GS + Ly::KAP];
#pragma unroll
      for (int i = 0; i < n; ++i) {
        const double fui = jr[Ly::FU + i];
#pragma unroll
        for (int j = 0; j < n; ++j) {
          loc[k].G[i][j] = fma(-fui, Kk[k][j], jr[Ly::FX + i * n + j]) - ((i == j) ? 1.0 : 0.0);
        }
        loc[k].c[i] = -fui * kap;
      }
      Aff2 t_;
      aff2_compose_dev(t_, loc[k], agg);
      agg = t_;
    }
    Aff2 P = agg;
    aff2_prefix_dpp(P);
    const double d0[n] = {x0r[0] - w.G[Ly::XB + 0], x0r[1] - w.G[Ly::XB + 1]};   // MPC re-solves move x0
    double ds[n];
#pragma unroll
    for (int i = 0; i < n; ++i) {
      const double ye = fma(P.G[i][0], d0[0], fma(P.G[i][1], d0[1], P.c[i] + d0[i]));
      const double yp = dpp_f64_or_zero<0x138, 0xF>(ye);                         // wave_shr:1
      ds[i] = (lane == 0) ? d0[i] : yp;
    }
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      X[k][0] = xb[k][0] + ds[0]; X[k][1] = xb[k][1] + ds[1];
      const double n0 = fma(loc[k].G[0][0], ds[0], fma(loc[k].G[0][1], ds[1], loc[k].c[0] + ds[0]));
      const double n1 = fma(loc[k].G[1][0], ds[0], fma(loc[k].G[1][1], ds[1], loc[k].c[1] + ds[1]));
      ds[0] = n0; ds[1] = n1;
    }
  }
#endif
  // Stop when a sweep moved the guess by less than kTol: the iteration is quadratic (error after a
  // sweep ~ 0.03 x the squared error before it, measured on C2), the update of a sweep IS the error
  // before it, so an update < 1e-7 leaves an error < 1e-15 - round-off.
  constexpr int kMaxSweeps = MI_NEWTON_MAX_SWEEPS;
  constexpr double kTol = 1e-7;
  bool converged = false;
#ifdef MI_PROF_NEWTON
  const long long pn0 = clock64(); int nsw = 0;
#endif
  // The sweeps work on the CORRECTION d_t = x_t(new) - X_t:  d_{t+1} = G_t d_t + r_t,  r_t = g_t(X_t) - X_{t+1},
  // d_0 = 0 - the same linearized recurrence as x_{t+1} = g_t(X_t) + G_t (x_t - X_t), with the defect r_t of the
  // current guess as its offset.  In this form a sweep can REUSE the Jacobians of the sweep before it (a chord
  // step): once a full sweep has moved the guess by less than kFrozenTol, the next one evaluates the plain fp64
  // step instead of the Dual2 one - the guess is already so close that G at the previous guess is G at the
  // solution to ~1e-4, and the correction it computes (the defect ~c u^2 of the last full sweep, u its update)
  // is left with an error ~2c u * c u^2: 6e-14 at the u = 3.5e-4 typical of C2's third sweep, where a full sweep
  // leaves 3e-19 - both far below the 1e-11 the a-posteriori guard (dynamics_hold, below) accepts.  A chord sweep
  // that does not converge is followed by a full one.
#ifndef MI_NEWTON_FROZEN_TOL
#define MI_NEWTON_FROZEN_TOL 5e-4
#endif
  constexpr double kFrozenTol = MI_NEWTON_FROZEN_TOL;
  double Gs[CH][n][n];                                       // closed-loop Jacobians of the last full sweep
  double prev_upd = __builtin_inf();
  bool have_g = false, last_frozen = false;
  double x_end[n] = {0.0, 0.0};                              // the state after this lane's last step, as of the last sweep
  for (int sweep = 0; sweep < kMaxSweeps && !converged; ++sweep) {
#ifdef MI_PROF_NEWTON
    ++nsw;
#endif
#ifdef MI_NEWTON_RELOAD
    // A/B variant (tools/isa_mix.py): the loop-invariant nominal data of the lane's steps is read from
    // LDS again in every sweep instead of being held in registers across the loop - 20 LDS reads for a loop without
    // register parking in the accumulation file
    asm volatile("" ::: "memory");
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const double* g = w.G + (valid[k] ? t0 + k : 0) * Ly::GS;
#pragma unroll
      for (int i = 0; i < n; ++i) { xb[k][i] = g[Ly::XB + i]; Kk[k][i] = g[Ly::KK + i]; }
      dd[k] = g[Ly::UB] - eps * g[Ly::KAP];
    }
#endif
    const bool frozen = have_g && !last_frozen && prev_upd < kFrozenTol;      // wave-uniform
    // X_{t+1} of this lane's last step = the next lane's first guess
    const double nx0 = dpp_f64_or_zero<0x130, 0xF>(X[0][0]), nx1 = dpp_f64_or_zero<0x130, 0xF>(X[0][1]);   // wave_shl:1
    Aff2 loc[CH], agg;
    agg.G[0][0] = 1.0; agg.G[0][1] = 0.0; agg.G[1][0] = 0.0; agg.G[1][1] = 1.0; agg.c[0] = 0.0; agg.c[1] = 0.0;
    if (!frozen) {
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        // g_t and G_t at the current guess: one Dual2 evaluation of the closed-loop step
        Dual2 xd[n] = {Dual2(X[k][0], 1.0, 0.0), Dual2(X[k][1], 0.0, 1.0)};
        Dual2 ud[m] = {dd[k] - (Kk[k][0] * (xd[0] - xb[k][0]) + Kk[k][1] * (xd[1] - xb[k][1]))};
        Dual2 xn[n];
        M::template step<Dual2>(xd, ud, xn, a.params, a.dt);
#pragma unroll
        for (int i = 0; i < n; ++i) {
          // (steps past the horizon evaluate record 0's data: their maps only enter the prefixes of later
          // lanes, which hold no valid step - nothing forces them to the identity)
          Gs[k][i][0] = xn[i].d0; Gs[k][i][1] = xn[i].d1;
          const double nxt = (k + 1 < CH) ? X[(k + 1 < CH) ? k + 1 : k][i] : (i == 0 ? nx0 : nx1);
          loc[k].c[i] = xn[i].v - nxt;                           // the defect of the guess at this step
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        double u[m] = {dd[k] - (Kk[k][0] * (X[k][0] - xb[k][0]) + Kk[k][1] * (X[k][1] - xb[k][1]))};
        double xn[n];
        M::template step<double>(X[k], u, xn, a.params, a.dt);
#pragma unroll
        for (int i = 0; i < n; ++i) {
          const double nxt = (k + 1 < CH) ? X[(k + 1 < CH) ? k + 1 : k][i] : (i == 0 ? nx0 : nx1);
          loc[k].c[i] = xn[i] - nxt;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < CH; ++k) {
#pragma unroll
      for (int i = 0; i < n; ++i) { loc[k].G[i][0] = Gs[k][i][0]; loc[k].G[i][1] = Gs[k][i][1]; }
      Aff2 t_;
      aff2_compose(t_, loc[k], agg);
      agg = t_;
    }
    // inclusive prefix over the lanes: P_l = agg_l o agg_{l-1} o ... o agg_0, all in DPP moves
    Aff2 P = agg;
    P.G[0][0] -= 1.0; P.G[1][1] -= 1.0;
    aff2_prefix_dpp(P);
    // this lane's first correction = the END value of the previous lane's prefix applied to d_0 = 0: its offset
    double ds[n];
#pragma unroll
    for (int i = 0; i < n; ++i) { const double yp = dpp_f64_or_zero<0x138, 0xF>(P.c[i]); ds[i] = (lane == 0) ? 0.0 : yp; }   // wave_shr:1
    double upd = 0.0;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      if (valid[k]) upd = fmax(upd, fmax(fabs(ds[0]), fabs(ds[1])));
      X[k][0] += ds[0]; X[k][1] += ds[1];
      const double n0 = fma(loc[k].G[0][0], ds[0], fma(loc[k].G[0][1], ds[1], loc[k].c[0]));
      const double n1 = fma(loc[k].G[1][0], ds[0], fma(loc[k].G[1][1], ds[1], loc[k].c[1]));
      ds[0] = n0; ds[1] = n1;
    }
    // the corrected state after the lane's last step: (the next lane's first guess, as this sweep saw it) + its correction
    x_end[0] = nx0 + ds[0]; x_end[1] = nx1 + ds[1];
    // NaN -> not converged: a NaN anywhere in this lane's corrections has travelled down the chain into the
    // last one (v_max_f64 above drops NaN operands, so it is tested here, once)
    if (valid[0]) upd = (ds[0] == ds[0] && ds[1] == ds[1]) ? upd : __builtin_inf();
    // wave-wide max of the update: DPP row rotations, then the four row maxima through v_readlane
    upd = fmax(upd, row_ror_f64<8>(upd));
    upd = fmax(upd, row_ror_f64<4>(upd));
    upd = fmax(upd, row_ror_f64<2>(upd));
    upd = fmax(upd, row_ror_f64<1>(upd));
    upd = fmax(fmax(readlane_f64(upd, 0), readlane_f64(upd, 16)), fmax(readlane_f64(upd, 32), readlane_f64(upd, 48)));
    have_g = true; last_frozen = frozen; prev_upd = upd;
    converged = upd < kTol;
#ifdef MI_PROF_NEWTON
    if (lane == 0 && sweep < 4) mi_dbg_vals[3 + sweep] = upd;
#endif
  }
#ifdef MI_PROF_NEWTON
  const long long pn1 = clock64();
  if (lane == 0) { mi_dbg_vals[0] = (double)(pn1 - pn0); mi_dbg_vals[1] = nsw; }
#endif
  if (!converged) return NEWTON_FAILED;
  // The trial trajectory IS the last sweep's result: its update applied the linearized recurrence, i.e.
  //   x_{t+1} = g_t(X_t) + G_t (x_t - X_t),
  // to corrections below kTol = 1e-7, so it misses the exact step x_{t+1} = g_t(x_t) (ilqr.py:313-316) by the
  // quadratic remainder ~c |x_t - X_t|^2 (c = 0.03 on C2: < 3e-16; a chord sweep leaves 2c u_full u_chord, at most
  // 2 x 0.03 x 5e-4 x 1e-7 = 3e-12 by its entry rule kFrozenTol and typically 6e-14) - the size of the chunk-edge defect
  // the earlier re-step of every chunk left at its 49 edges, without that re-step's four model evaluations per lane.
  // A-posteriori guard of the stopping rule, model by model and trial by trial: exact steps against the states the
  // trajectory holds - ONE per lane (its last) for the built-in smooth models this was measured on, EVERY step of
  // the chunk for any other model (plugins); where the sweeps do not contract as measured, or a kink sits inside a
  // chunk, the defect shows it and the caller falls back to the sequential rollout.
  auto dynamics_hold = [&]() __attribute__((always_inline)) -> bool {
    constexpr double kEdgeTol = 1e-11;
    double dfc = 0.0;
    // built-in smooth models (M::kNewtonRollout: the quadratic remainder was MEASURED, see above): the lane's last step;
    // any other n = 2 model (plugins - a kink such as fmax or a contact inside a chunk would leave a defect of the size of
    // the last correction that the last step alone does not see): EVERY valid step of the chunk, CH - 1 more plain steps
    constexpr int kFirst = NewtonMeasured<M>::value ? CH - 1 : 0;
#pragma unroll
    for (int k = kFirst; k < CH; ++k) {
      if (valid[k]) {
        double u[m], xe[n];
        u[0] = dd[k] - (Kk[k][0] * (X[k][0] - xb[k][0]) + Kk[k][1] * (X[k][1] - xb[k][1]));
        M::template step<double>(X[k], u, xe, a.params, a.dt);
        const double nx_0 = (k + 1 < CH) ? X[(k + 1 < CH) ? k + 1 : k][0] : x_end[0];
        const double nx_1 = (k + 1 < CH) ? X[(k + 1 < CH) ? k + 1 : k][1] : x_end[1];
        const double d = fmax(fabs(xe[0] - nx_0), fabs(xe[1] - nx_1));
        const double sc = fmax(1.0, fmax(fabs(xe[0]), fabs(xe[1])));
        dfc = (d <= kEdgeTol * sc) ? dfc : 1.0;                 // NaN -> does not hold
      }
    }
    dfc = fmax(dfc, row_ror_f64<8>(dfc));
    dfc = fmax(dfc, row_ror_f64<4>(dfc));
    dfc = fmax(dfc, row_ror_f64<2>(dfc));
    dfc = fmax(dfc, row_ror_f64<1>(dfc));
    return fmax(fmax(readlane_f64(dfc, 0), readlane_f64(dfc, 16)), fmax(readlane_f64(dfc, 32), readlane_f64(dfc, 48))) == 0.0;
  };
  if (!dynamics_hold()) return NEWTON_FAILED;                  // (the caller's sequential rollout takes over)
  // final pass over this lane's steps: controls (and, fused, the cost) from the trajectory in registers
  if (fuse == 0) {
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < n; ++i) w.T[Ly::XN + i] = x0r[i];
    }
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      if (valid[k]) {
        const int t = t0 + k;
        const double u0 = dd[k] - (Kk[k][0] * (X[k][0] - xb[k][0]) + Kk[k][1] * (X[k][1] - xb[k][1]));
        double* tr = w.T + t * Ly::TS;
        tr[Ly::UN] = u0;
        tr[Ly::TS + Ly::XN + 0] = (k + 1 < CH) ? X[(k + 1 < CH) ? k + 1 : k][0] : x_end[0];
        tr[Ly::TS + Ly::XN + 1] = (k + 1 < CH) ? X[(k + 1 < CH) ? k + 1 : k][1] : x_end[1];
      }
    }
#ifdef MI_PROF_NEWTON
    if (lane == 0) mi_dbg_vals[2] = (double)(clock64() - pn1);
#endif
    return NEWTON_STORED;
  }
  double xk[CH][n], uk[CH][m], xlast[n] = {0.0, 0.0};
  double cost = 0.0, dvs = 0.0;
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    xk[k][0] = X[k][0]; xk[k][1] = X[k][1]; uk[k][0] = 0.0;
    if (valid[k]) {
      const int t = t0 + k;
      double u[m];
      u[0] = dd[k] - (Kk[k][0] * (X[k][0] - xb[k][0]) + Kk[k][1] * (X[k][1] - xb[k][1]));
      uk[k][0] = u[0];
      cost += stage_cost<M>(c, X[k], u);                       // ilqr.py:325
      dvs += w.G[t * Ly::GS + Ly::DV];                         // :326
      if (t == steps - 1) {                                    // :327
        const double xn[n] = {(k + 1 < CH) ? X[(k + 1 < CH) ? k + 1 : k][0] : x_end[0], (k + 1 < CH) ? X[(k + 1 < CH) ? k + 1 : k][1] : x_end[1]};
        cost += terminal_cost<M>(c, xn);
        xlast[0] = xn[0]; xlast[1] = xn[1];
      }
    }
  }
  const double L = wave_sum(cost);
  const double ex = -eps * (1.0 - eps / 2.0) * wave_sum(dvs);
  L_out = L;
  if (!((L_last - L) > a.gamma * ex)) return NEWTON_REJECTED;  // ilqr.py:330-331
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    if (valid[k]) {
      const int t = t0 + k;
      double* g = w.G + t * Ly::GS;
      g[Ly::XB + 0] = xk[k][0]; g[Ly::XB + 1] = xk[k][1];
      g[Ly::UB] = uk[k][0];
      if (t == steps - 1) { g[Ly::GS + Ly::XB + 0] = xlast[0]; g[Ly::GS + Ly::XB + 1] = xlast[1]; }
      if (fuse == 2) {
        double* j = w.J + t * Ly::JS;
#pragma unroll
        for (int col = 0; col < n + m; ++col) {
          double d[n];
          jac_column<M, JAC>(xk[k], uk[k], col, a, d);
#pragma unroll
          for (int i = 0; i < n; ++i) {
            if (col < n) j[Ly::FX + i * n + col] = d[i];
            else j[Ly::FU + i * m + (col - n)] = d[i];
          }
        }
      }
    }
  }
#ifdef MI_PROF_NEWTON
  if (lane == 0) mi_dbg_vals[2] = (double)(clock64() - pn1);
#endif
  return NEWTON_ACCEPTED;
}

template <class M>
__device__ __forceinline__ bool newton_capable(const WS& w, const KArgs& a) {
  // four steps per lane: horizons up to N = 257
  if constexpr (M::n == 2 && M::m == 1) return a.newton_rollout != 0 && w.N - 1 <= 64 * 4;
  return false;
}
template <class M, int JAC>
__device__ inline int rollout_newton(const WS& w, const Consts<M>& c, const KArgs& a, const double* x0r, double eps,
                                     int fuse, double L_last, double& L_out, bool coarse = false) {
  if constexpr (M::n == 2 && M::m == 1) {
    if (newton_capable<M>(w, a)) return rollout_newton_impl<M, JAC, 4>(w, c, a, x0r, eps, fuse, L_last, L_out, coarse);
  }
  return NEWTON_FAILED;
}

// ---------------------------------------------------------------------------
// Speculative parallel line search (ilqr.py:300-337).  Returns true on accept;
// the T records then hold the accepted trajectory.  `trials` is the
// reference-equivalent sequential trial count (accepted candidate index + 1).
// ---------------------------------------------------------------------------
// `optimistic`: the caller expects eps = 1 to be accepted (it was at the previous iteration, or
// this is the first one).  Then the first trial is rolled out WITHOUT the per-step cost
// arithmetic (a fifth of the sequential instruction stream) and its cost is evaluated
// time-parallel afterwards; only if that trial is rejected does the speculative 64-candidate
// pass run.  The accepted candidate is the same either way.
// `fuse` (see rollout_newton_impl): what the time-parallel rollout may do beyond the trial itself;
// `fused_out` reports what it did for the accepted trial (0: trajectory in T slot `slot_out`;
// 1: already committed to the nominal records; 2: committed and linearized).
template <class M, int JAC>
__device__ inline bool linesearch(const WS& w, const Consts<M>& c, const KArgs& a, const double* x0r,
                                  double L_last, bool optimistic, int fuse, double& L_out, double& eps_out, int& trials,
                                  int& slot_out, int& fused_out, bool cold_start = false, bool no_newton = false) {
  fused_out = 0;
  const int lane = threadIdx.x;
  int base = 0;
  double eps_base = 1.0;
  // the stored nominal trajectory is a usable first guess except at the first iteration of a solve
  const bool newton = !no_newton && newton_capable<M>(w, a) && L_last < __builtin_inf();
  if (optimistic) {
    double L, ex;
    int nr = NEWTON_FAILED;
    if (newton) nr = rollout_newton<M, JAC>(w, c, a, x0r, 1.0, fuse, L_last, L);
    else if (cold_start && !no_newton && newton_capable<M>(w, a)) nr = rollout_newton<M, JAC>(w, c, a, x0r, 1.0, fuse, L_last, L, true);
    if (nr == NEWTON_ACCEPTED) {
      L_out = L;
      eps_out = 1.0;
      trials = 1;
      slot_out = 0;
      fused_out = fuse;
      return true;
    }
    const bool done = nr == NEWTON_STORED;
    // Newton tried and not converged (a step that large is about to be rejected anyway) or rejected:
    // no second, sequential attempt at eps = 1 - the candidate pass below has it in lane 0
    if (done || !newton) {
#ifdef MI_PROF_NEWTON
      const long long pr0 = clock64();
#endif
      if (!done) rollout<M, false>(w, c, a, x0r, 1.0, lane == 0 ? 0 : -1, L, ex);
      wave_sync();
#ifdef MI_PROF_NEWTON
      const long long pr1 = clock64();
#endif
      traj_cost<M>(w, c, 1.0, L, ex);
#ifdef MI_PROF_NEWTON
      if constexpr (M::n == 2 && M::m == 1) { if (!done && lane == 0) { mi_dbg_vals[0] = (double)(pr1 - pr0); mi_dbg_vals[1] = 0; mi_dbg_vals[2] = (double)(clock64() - pr1); } }
#endif
      if ((L_last - L) > a.gamma * ex) {                         // ilqr.py:330-331
        L_out = L;
        eps_out = 1.0;
        trials = 1;
        slot_out = 0;
        return true;
      }
      wave_sync();
    }
  }
  for (;;) {
    double eps = eps_base;
    for (int i = 0; i < lane; ++i) eps *= a.beta;   // eps *= beta, repeated (ilqr.py:335): bit-identical sequence
    const bool valid = eps >= 1e-8;                 // while eps >= 1e-8 (ilqr.py:302)
    double L, ex;
    // the first n_store candidates keep their trajectories (coarse line searches, beta <= 0.75,
    // usually accept one of them: no second rollout needed)
    rollout<M>(w, c, a, x0r, eps, lane < w.n_store ? lane : -1, L, ex);
    const bool acc = valid && ((L_last - L) > a.gamma * ex);   // ilqr.py:330-331
    const unsigned long long mask = __ballot(acc);
    if (mask != 0ull) {
      const int k = __ffsll((long long)mask) - 1;
      if (k < w.n_store) {
        L_out = __shfl(L, k);
        eps_out = __shfl(eps, k);
        trials = base + k + 1;
        slot_out = k;
        return true;
      }
      // candidate base+k wins and its trajectory was not kept: roll it out once more, stored - parallel
      // in time when that converges (the cost is then re-evaluated on the stored trajectory) ...
      if (newton) {
        const double eps_k = __shfl(eps, k);
        wave_sync();
        // (the sequential rollout of lane k already passed the acceptance test: L_last = inf here)
        const int nr = rollout_newton<M, JAC>(w, c, a, x0r, eps_k, fuse, __builtin_inf(), L_out);
        if (nr == NEWTON_STORED) {
          wave_sync();
          double ex_;
          traj_cost<M>(w, c, eps_k, L_out, ex_);
        }
        if (nr == NEWTON_STORED || nr == NEWTON_ACCEPTED) {
          eps_out = eps_k;
          trials = base + k + 1;
          slot_out = 0;
          fused_out = (nr == NEWTON_ACCEPTED) ? fuse : 0;
          return true;
        }
      }
      // ... otherwise by another pass with it in lane 0
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

// Commit the accepted trial: x_bar <- x, u_bar <- u (ilqr.py:375-376).
template <int n, int m>
__device__ inline void commit_trial(const WS& w, int slot) {
  using Ly = Lay<n, m>;
  const double* Ts = w.T + slot * w.t_stride;
  for (int t = threadIdx.x; t < w.N; t += 64) {
    const double* s = Ts + t * Ly::TS;
    double* d = w.G + t * Ly::GS;
#pragma unroll
    for (int i = 0; i < n; ++i) d[Ly::XB + i] = s[Ly::XN + i];
    if (t < w.N - 1) {
#pragma unroll
      for (int k = 0; k < m; ++k) d[Ly::UB + k] = s[Ly::UN + k];
    }
  }
}

// ---------------------------------------------------------------------------
// Dynamics partials at the listed time steps of the NOMINAL trajectory in G
// (replaces _calc_dynamics_partials, ilqr.py:233-272): (list entry, column)
// items over lanes.
// ---------------------------------------------------------------------------
template <class M, int JAC>
__device__ __forceinline__ void jac_item(const WS& w, const KArgs& a, int t, int col) {
  constexpr int n = M::n, m = M::m;
  using Ly = Lay<n, m>;
  const double* g = w.G + t * Ly::GS;
  double x[n], u[m], d[n];
#pragma unroll
  for (int i = 0; i < n; ++i) x[i] = g[Ly::XB + i];
#pragma unroll
  for (int k = 0; k < m; ++k) u[k] = g[Ly::UB + k];
  jac_column<M, JAC>(x, u, col, a, d);
  double* j = w.J + t * Ly::JS;
  if (col < n) {
#pragma unroll
    for (int i = 0; i < n; ++i) j[Ly::FX + i * n + col] = d[i];
  } else {
#pragma unroll
    for (int i = 0; i < n; ++i) j[Ly::FU + i * m + (col - n)] = d[i];
  }
}

template <class M, int JAC>
__device__ __forceinline__ void jac_at(const WS& w, const KArgs& a, const int* list, int count) {
  constexpr int nc = M::n + M::m;
  for (int it = threadIdx.x; it < count * nc; it += 64) {
    const int ki = it / nc, col = it - ki * nc;
    jac_item<M, JAC>(w, a, list[ki], col);
  }
}

// ---------------------------------------------------------------------------
// Team linearization (setInterval, minN = 1: every step is a key-point).  The linearization is
// the one stage of an iteration that is parallel over time steps, and at the batch sizes of the
// wave-per-problem kernels three of a CU's four SIMDs have issue slots to spare at any moment:
// the problem's (step, column) items are dealt round-robin, 64 at a time, to the main wave and
// its helper waves.  Helpers are parked on the workgroup barrier at all other times; the
// protocol is barrier - work - barrier, no polling.
// ---------------------------------------------------------------------------
enum { TEAM_CMD_EXIT = 0, TEAM_CMD_LINEARIZE = 1 };
typedef __attribute__((address_space(3))) volatile int lds_vint_t;      // explicit LDS pointer: ds_read/ds_write, not flat

template <class M, int JAC>
__device__ __forceinline__ void jac_rounds(const WS& w, const KArgs& a, int first, int stride, int lane) {
  constexpr int nc = M::n + M::m;
  const int items = (w.N - 1) * nc;
  for (int it = 64 * first + lane; it < items; it += 64 * stride) {
    const int t = it / nc;
    jac_item<M, JAC>(w, a, t, it - t * nc);
  }
}

template <class M, int JAC>
__device__ inline void helper_wave(const WS& w, const KArgs& a, int wave, int team) {
  const int lane = threadIdx.x & 63;
  lds_vint_t* cmd = (lds_vint_t*)w.aux;
  for (;;) {
    team_barrier();
    if (__builtin_amdgcn_readfirstlane(cmd[0]) == TEAM_CMD_EXIT) return;
    jac_rounds<M, JAC>(w, a, wave, team, lane);
    team_barrier();
  }
}

// Accessor over the LDS records for the shared key-point code (keypoints.hpp).
template <int n_, int m_>
struct SmallAcc {
  static constexpr int n = n_, m = m_;
  using Ly = Lay<n_, m_>;
  double *G, *J;
  int *kp, *aux, *need, *binA, *binB;
  int N;
  __device__ SmallAcc(const WS& w) : G(w.G), J(w.J), kp(w.kp), aux(w.aux), need(w.need), binA(w.binA), binB(w.binB), N(w.N) {}
  __device__ __forceinline__ double x(int t, int i) const { return G[t * Ly::GS + Ly::XB + i]; }
  __device__ __forceinline__ double fx(int t, int r) const { return J[t * Ly::JS + Ly::FX + r]; }
  __device__ __forceinline__ double fu(int t, int r) const { return J[t * Ly::JS + Ly::FU + r]; }
  __device__ __forceinline__ void set_fx(int t, int r, double v) const { J[t * Ly::JS + Ly::FX + r] = v; }
  __device__ __forceinline__ void set_fu(int t, int r, double v) const { J[t * Ly::JS + Ly::FU + r] = v; }
};

// _get_derivatives (ilqr.py:380-415) at the nominal trajectory in G.  Returns key-point count.
template <class M, int JAC>
__device__ inline int linearize(const WS& w, const KArgs& a) {
  SmallAcc<M::n, M::m> acc(w);
  return linearize_generic(acc, a.kp_method, a.minN, a.maxN, a.jerk_thr, a.err_thr,
                           [&](const int* list, int count) __attribute__((always_inline)) { jac_at<M, JAC>(w, a, list, count); });
}

template <int m>
__device__ __forceinline__ void invert_small(const double (&A)[m][m], double (&Ai)[m][m]) {
  static_assert(m >= 1 && m <= 2, "wave-per-problem path covers m <= 2");
  if constexpr (m == 1) {
    Ai[0][0] = fast_rcp(A[0][0]);
  } else {
    const double id = fast_rcp(A[0][0] * A[1][1] - A[0][1] * A[1][0]);
    Ai[0][0] = A[1][1] * id; Ai[0][1] = -A[0][1] * id;
    Ai[1][0] = -A[1][0] * id; Ai[1][1] = A[0][0] * id;
  }
}

// ---------------------------------------------------------------------------
// Backward Riccati pass (ilqr.py:623-667) with the quadratic cost expansion
// (:161-206) fused in.  Wave-uniform: every lane carries the same Vx/Vxx in
// registers; G/J records are broadcast-read one step ahead into alternating
// register sets; lane 0 writes the gains back into the G record.
// ---------------------------------------------------------------------------
template <class M>
struct BRegs {
  double lx[M::n], lu[M::m], fx[M::n][M::n], fu[M::n][M::m];
  // `tr`: the T record of this step, holding (lx_t, lu_t) from cost_gradients()
  __device__ __forceinline__ void load(const double* tr, const double* j) {
    using L = Lay<M::n, M::m>;
#pragma unroll
    for (int i = 0; i < M::n; ++i) {
      lx[i] = tr[L::XN + i];
#pragma unroll
      for (int c = 0; c < M::n; ++c) fx[i][c] = j[L::FX + i * M::n + c];
#pragma unroll
      for (int k = 0; k < M::m; ++k) fu[i][k] = j[L::FU + i * M::m + k];
    }
#pragma unroll
    for (int k = 0; k < M::m; ++k) lu[k] = tr[L::UN + k];
  }
};

// Running-cost gradients (ilqr.py:180-181) lx_t = 2Q x_t - 2 x_nom^T Q, lu_t = 2R u_t of the
// nominal trajectory, time-parallel (one step per lane) into T buffer 0, which is idle between
// commit_trial and the next line search.  The sequential sweep then starts its Qx/Qu sums from
// these values: the same fma sequence as accumulating them in place, five fewer instructions on
// the critical path of every step.
template <class M>
__device__ inline void cost_gradients(const WS& w, const Consts<M>& c, const double (&Q2)[M::n][M::n],
                                      const double (&R2)[M::m][M::m]) {
  constexpr int n = M::n, m = M::m;
  using Ly = Lay<n, m>;
  for (int t = threadIdx.x; t < w.N - 1; t += 64) {
    const double* g = w.G + t * Ly::GS;
    double* tr = w.T + t * Ly::TS;
    double x[n], u[m];
#pragma unroll
    for (int i = 0; i < n; ++i) x[i] = g[Ly::XB + i];
#pragma unroll
    for (int k = 0; k < m; ++k) u[k] = g[Ly::UB + k];
#pragma unroll
    for (int i = 0; i < n; ++i) {
      double s = -c.qn[i];
#pragma unroll
      for (int j = 0; j < n; ++j) s += Q2[i][j] * x[j];
      tr[Ly::XN + i] = s;
    }
#pragma unroll
    for (int a_ = 0; a_ < m; ++a_) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < m; ++j) s += R2[a_][j] * u[j];
      tr[Ly::UN + a_] = s;
    }
  }
}

template <class M>
__device__ __forceinline__ void backward_step(const BRegs<M>& r, const Consts<M>& c, const double (&Q2)[M::n][M::n],
                                              const double (&R2)[M::m][M::m], double (&Vx)[M::n],
                                              double (&Vxx)[M::n][M::n], double* gw) {
  constexpr int n = M::n, m = M::m;
  using Ly = Lay<n, m>;
  // cost partials (ilqr.py:180-184): lx = 2Qx - 2x_nom^T Q, lu = 2Ru, lxx = 2Q, luu = 2R, lux = 0
  double Qx[n], Qu[m], Qxx[n][n], Quu[m][m], Qux[m][n];
#pragma unroll
  for (int i = 0; i < n; ++i) {
    double s = r.lx[i];
#pragma unroll
    for (int k = 0; k < n; ++k) s += r.fx[k][i] * Vx[k];
    Qx[i] = s;                                              // :651
  }
#pragma unroll
  for (int a_ = 0; a_ < m; ++a_) {
    double s = r.lu[a_];
#pragma unroll
    for (int k = 0; k < n; ++k) s += r.fu[k][a_] * Vx[k];
    Qu[a_] = s;                                             // :652
  }
  // A = fx^T Vxx (n x n), Bm = fu^T Vxx (m x n)  — the reference's association (fx.T@Vxx)@fx
  double A[n][n], Bm[m][n];
#pragma unroll
  for (int i = 0; i < n; ++i)
#pragma unroll
    for (int j = 0; j < n; ++j) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < n; ++k) s += r.fx[k][i] * Vxx[k][j];
      A[i][j] = s;
    }
#pragma unroll
  for (int a_ = 0; a_ < m; ++a_)
#pragma unroll
    for (int j = 0; j < n; ++j) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < n; ++k) s += r.fu[k][a_] * Vxx[k][j];
      Bm[a_][j] = s;
    }
#pragma unroll
  for (int i = 0; i < n; ++i)
#pragma unroll
    for (int j = 0; j < n; ++j) {
      double s = Q2[i][j];
#pragma unroll
      for (int k = 0; k < n; ++k) s += A[i][k] * r.fx[k][j];
      Qxx[i][j] = s;                                        // :653
    }
#pragma unroll
  for (int a_ = 0; a_ < m; ++a_) {
#pragma unroll
    for (int b_ = 0; b_ < m; ++b_) {
      double s = R2[a_][b_];
#pragma unroll
      for (int k = 0; k < n; ++k) s += Bm[a_][k] * r.fu[k][b_];
      Quu[a_][b_] = s;                                      // :654
    }
#pragma unroll
    for (int j = 0; j < n; ++j) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < n; ++k) s += Bm[a_][k] * r.fx[k][j];
      Qux[a_][j] = s;                                       // :656 (lux = 0)
    }
  }
  double Qi[m][m];
  invert_small<m>(Quu, Qi);                                 // :655 explicit inverse
  double kap[m], Kg[m][n], QuQi[m];
#pragma unroll
  for (int a_ = 0; a_ < m; ++a_) {
    double s = 0.0, q = 0.0;
#pragma unroll
    for (int b_ = 0; b_ < m; ++b_) { s += Qi[a_][b_] * Qu[b_]; q += Qu[b_] * Qi[b_][a_]; }
    kap[a_] = s;                                            // :659
    QuQi[a_] = q;                                           // Qu^T Quu_inv
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
#pragma unroll
  for (int a_ = 0; a_ < m; ++a_) {
    gw[Ly::KAP + a_] = kap[a_];
#pragma unroll
    for (int j = 0; j < n; ++j) gw[Ly::KK + a_ * n + j] = Kg[a_][j];
  }
  gw[Ly::DV] = dv;
  // Vx = Qx - Qu^T Quu_inv Qux ; Vxx = Qxx - Qux^T Quu_inv Qux   (:666-667; no symmetrization)
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
}

template <class M>
__device__ inline void backward_scalar(const WS& w, const Consts<M>& c) {
  constexpr int n = M::n, m = M::m;
  using Ly = Lay<n, m>;
  const int N = w.N;
  double Q2[n][n], R2[m][m];
#pragma unroll
  for (int i = 0; i < n; ++i)
#pragma unroll
    for (int j = 0; j < n; ++j) Q2[i][j] = 2.0 * c.Q[i][j];
#pragma unroll
  for (int i = 0; i < m; ++i)
#pragma unroll
    for (int j = 0; j < m; ++j) R2[i][j] = 2.0 * c.R[i][j];
  double Vx[n], Vxx[n][n];
  {
    const double* gT = w.G + (N - 1) * Ly::GS;
#pragma unroll
    for (int i = 0; i < n; ++i) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < n; ++j) { s += (2.0 * c.Qf[i][j]) * gT[Ly::XB + j]; Vxx[i][j] = 2.0 * c.Qf[i][j]; }
      Vx[i] = s - c.qfn[i];                                 // ilqr.py:203-204
    }
  }
  cost_gradients<M>(w, c, Q2, R2);
  wave_sync();
  const bool writer = threadIdx.x == 0;
  const double* g = w.T + (N - 2) * Ly::TS;
  const double* j = w.J + (N - 2) * Ly::JS;
  double* gw = writer ? (w.G + (N - 2) * Ly::GS) : (w.dump + 2 * threadIdx.x);
  const int gstep = writer ? Ly::GS : 0;
  BRegs<M> A, B;
  A.load(g, j);
  int t = N - 2;
  for (; t >= 1; t -= 2) {
    B.load(g - Ly::TS, j - Ly::JS);
    __builtin_amdgcn_sched_barrier(0);                   // keep the prefetch a full step ahead
    backward_step<M>(A, c, Q2, R2, Vx, Vxx, gw);
    gw -= gstep;
    A.load(g - 2 * Ly::TS, j - 2 * Ly::JS);              // t-2 >= -1: the leading pad record
    __builtin_amdgcn_sched_barrier(0);
    backward_step<M>(B, c, Q2, R2, Vx, Vxx, gw);
    gw -= gstep;
    g -= 2 * Ly::TS;
    j -= 2 * Ly::JS;
  }
  if (t == 0) backward_step<M>(A, c, Q2, R2, Vx, Vxx, gw);
}

typedef double d4s_t __attribute__((ext_vector_type(4)));

// value of lane LANE of this lane's 16-lane row, for a double: one v_mov_b64_dpp row_newbcast
// (gfx90a+ DPP64).  bound_ctrl with full row/bank masks: every lane is written.
template <int LANE>
__device__ __forceinline__ double row_share(double v) {
  return __builtin_amdgcn_update_dpp(v, v, 0x150 + LANE, 0xF, 0xF, true);
}

__device__ __forceinline__ double readlane_f64(double v, int srclane) {
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], srclane);
  u.i[1] = __builtin_amdgcn_readlane(u.i[1], srclane);
  return u.d;
}

// ---------------------------------------------------------------------------
// Backward Riccati pass on the fp64 matrix core, for 3 <= n <= 4, m = 1.
//
// The scalar formulation costs ~70 n^2 wave-uniform instructions per step at one
// instruction per ~5 cycles (n = 4: ~1100+ cycles).  Here the whole second-order
// expansion of a step is TWO v_mfma_f64_16x16x4_f64 with the matrices spread one
// element per lane (lane = 16*lk + lr):
//     F = [fx | fu]           held at lane (lk = k, lr = c)        (B layout == A^T layout)
//     S = [Vxx | Vx at col CV]  at lane (lk = i, lr = j)           (the D layout of a result)
//     T  = S^T F              : A = S (read through its D-layout registers), B = F
//     H  = F^T [T | Vx]       : A = F (as F^T), B = T with column CV replaced by Vx
// H holds Qxx-lxx (rows < n), Qux, Quu-luu, and F^T Vx in column CV.  The A operand of the second
// product carries fu a second time in rows 8..11, so output register 2 of EVERY lane holds
// H[n][lr] (the Qux row, Quu and Qu) and the gains need no cross-row shuffle: Quu, Qu and the
// column form of Qux arrive by one DPP64 row broadcast each.  S is consumed through its
// transpose and Qux^T is taken from column n of H (the D layout of one product is the A^T layout
// of the next); Vxx is symmetric up to round-off — the reference never symmetrizes it either —
// so results differ from the scalar path at the 1e-16 level only.
// ---------------------------------------------------------------------------
template <class M>
__device__ inline void backward_mfma(const WS& w, const Consts<M>& c) {
  constexpr int n = M::n, m = M::m, nm = n + m, CV = nm;
  static_assert(m == 1 && n <= 4 && nm + 1 <= 8, "shape covered by one 16x16x4 tile with rows 8..11 spare");
  using Ly = Lay<n, m>;
  const int N = w.N, lane = threadIdx.x, lr = lane & 15, lk = lane >> 4;
  const bool in_blk = lk < n && lr < n;
  const bool is_cv = lk < n && lr == CV;
  // lane constants
  double q2e = 0.0, q2row[n], qn_l = 0.0;
#pragma unroll
  for (int i = 0; i < n; ++i) q2row[i] = 0.0;
#pragma unroll
  for (int i = 0; i < n; ++i)
#pragma unroll
    for (int j = 0; j < n; ++j) {
      if (lk == i && lr == j) q2e = 2.0 * c.Q[i][j];
      if (lk == i) q2row[j] = 2.0 * c.Q[i][j];
    }
#pragma unroll
  for (int i = 0; i < n; ++i) if (lk == i) qn_l = c.qn[i];
  const double R2 = 2.0 * c.R[0][0];
  // F element of this lane inside a J record (clamped to a valid slot, masked by fvalid);
  // lanes lr = 8..11 carry fu[lk] again: rows 8..11 of F^T, see above
  const bool frep = lk < n && lr >= 8 && lr < 12;
  const bool fvalid = lk < n && (lr < nm || frep);
  int foff = Ly::FX;
  if (fvalid) foff = (lr < n) ? (Ly::FX + lk * n + lr) : (Ly::FU + lk * m);

  // terminal: S = [2 Qf | 2 Qf x_T - 2 x_nom^T Qf]   (ilqr.py:203-204, :638)
  double S = 0.0;
  {
    const double* gT = w.G + (N - 1) * Ly::GS;
    double vx = 0.0, qfe = 0.0;
#pragma unroll
    for (int i = 0; i < n; ++i) {
      if (lk == i) {
        double s_ = -c.qfn[i];
#pragma unroll
        for (int j = 0; j < n; ++j) { s_ += (2.0 * c.Qf[i][j]) * gT[Ly::XB + j]; if (lr == j) qfe = 2.0 * c.Qf[i][j]; }
        vx = s_;
      }
    }
    S = in_blk ? qfe : (is_cv ? vx : 0.0);
  }
  // write targets: K[j] from lanes (lk == 0, lr < n); kappa, dV from lane 0; everything else to the dump slot
  const bool kwriter = lk == 0 && lr < n;
  double* kw = kwriter ? (w.G + (N - 2) * Ly::GS + Ly::KK + lr) : (w.dump + 2 * lane);
  double* sw = lane == 0 ? (w.G + (N - 2) * Ly::GS) : (w.dump + 2 * lane);
  const int kstep = kwriter ? Ly::GS : 0, sstep = lane == 0 ? Ly::GS : 0;

  const double* g = w.G + (N - 2) * Ly::GS;
  const double* jrec = w.J + (N - 2) * Ly::JS;
  struct Regs { double f, xb[n], ub; };
  auto load = [&](Regs& r, const double* gp, const double* jp) __attribute__((always_inline)) {
    r.f = jp[foff];
#pragma unroll
    for (int i = 0; i < n; ++i) r.xb[i] = gp[Ly::XB + i];
    r.ub = gp[Ly::UB];
  };
  auto step = [&](const Regs& r) __attribute__((always_inline)) {
    const double f = fvalid ? r.f : 0.0;
    const double a1 = (lr < n) ? S : 0.0;                          // Vxx part of S (as S^T through the layout)
    d4s_t z = {0.0, 0.0, 0.0, 0.0};
    const d4s_t T = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, f, z, 0, 0, 0);
    const double b2 = (lr == CV) ? S : T[0];                       // [T | Vx]
    const d4s_t H = __builtin_amdgcn_mfma_f64_16x16x4f64(f, b2, z, 0, 0, 0);
    const double hq = H[2];                                        // H[n][lr] in every lane (rows 8..11 = row n)
    const double qux_col = hq;                                     // Qux[lr] for lr < n
    const double qux_row = row_share<n>(H[0]);                     // H[lk][n] = Qxu[lk] = Qux[lk] to round-off
    const double Quu = row_share<n>(hq) + R2;                      // :654
    const double Qu = row_share<CV>(hq) + R2 * r.ub;               // :652 (lu = 2 R u)
    const double inv = fast_rcp(Quu);                              // :655
    const double kap = inv * Qu;                                   // :659
    const double dv = Qu * kap;                                    // :663
    const double Kq = inv * qux_col;                               // :660
    // lx of this lane's row (ilqr.py:180)
    double lx = -qn_l;
#pragma unroll
    for (int i = 0; i < n; ++i) lx += q2row[i] * r.xb[i];
    const double sxx = fma(-qux_row, Kq, H[0] + q2e);              // Qxx - Qux^T K   (:653,:667)
    const double svx = fma(-qux_row, kap, H[0] + lx);              // Qx - Qux^T kappa (:651,:666)
    S = in_blk ? sxx : (is_cv ? svx : 0.0);
    kw[0] = Kq;
    sw[Ly::KAP] = kap;
    sw[Ly::DV] = dv;
    kw -= kstep;
    sw -= sstep;
  };
  Regs A, B;
  load(A, g, jrec);
  int t = N - 2;
  for (; t >= 1; t -= 2) {
    load(B, g - Ly::GS, jrec - Ly::JS);
    __builtin_amdgcn_sched_barrier(0);
    step(A);
    load(A, g - 2 * Ly::GS, jrec - 2 * Ly::JS);          // t-2 >= -1: the leading pad record
    __builtin_amdgcn_sched_barrier(0);
    step(B);
    g -= 2 * Ly::GS;
    jrec -= 2 * Ly::JS;
  }
  if (t == 0) step(A);
}

// ---------------------------------------------------------------------------
// Backward Riccati pass PARALLEL IN TIME (n = 2), the 64 lanes along the horizon.
//
// The sequential sweep keeps all 64 lanes busy with one and the same scalar recursion - 2(N-1)
// dependent steps at one instruction per ~5 cycles.  But the recursion is an associative
// composition (Sarkka & Garcia-Fernandez, "Temporal parallelization of dynamic programming and
// linear quadratic control", IEEE TAC 2023): the second-order expansion of step t is an element
//     a_t = (A, b, C, eta, J) = (fx_t, -fu_t luu^{-1} lu_t, fu_t luu^{-1} fu_t^T, -lx_t, lxx)
// (terminal: (0, 0, 0, -lf_x, lf_xx)), elements compose by
//     a_i (x) a_j :  M = (I + C_i J_j)^{-1},  W = A_j M,  V = (M A_i)^T
//         A = W A_i,  b = W (b_i + C_i eta_j) + b_j,  C = W C_i A_j^T + C_j,
//         eta = V (eta_j - J_j b_i) + eta_i,  J = V J_j A_i + J_i,
// and the value function at step t is (Vxx, Vx) = (J, -eta) of a_t (x) a_{t+1} (x) ... (x) a_{N-1}.
// So: (1) every lane composes its own chunk of ceil(N/64) consecutive elements; (2) a
// Kogge-Stone suffix scan over the lanes (6 compositions, operands fetched with ds_bpermute)
// gives every lane the value function at the right edge of its chunk; (3) from there each lane
// runs the REFERENCE recursion (ilqr.py:651-667, backward_step) over its own few steps and
// writes K_t, kappa_t, dV_t.  ~1.7 k instructions instead of ~12 k on the critical path.
// Only the chunk-edge value functions come out of the re-associated arithmetic: measured against
// the sequential sweep on C2 iterations the gains agree to <= 4e-13 relative
// (oracle-side prototype: same formulas in NumPy), inside every tolerance of the parity tests.
// ---------------------------------------------------------------------------
template <int n>
struct RicElem {
  double A[n][n], b[n], C[n][n], e[n], J[n][n];
};

template <int n>
__device__ __forceinline__ void ric_identity(RicElem<n>& r) {
#pragma unroll
  for (int i = 0; i < n; ++i) {
    r.b[i] = 0.0; r.e[i] = 0.0;
#pragma unroll
    for (int j = 0; j < n; ++j) { r.A[i][j] = (i == j) ? 1.0 : 0.0; r.C[i][j] = 0.0; r.J[i][j] = 0.0; }
  }
}

// M = P^-1 for P = I + C_i J_j.  n = 2: adjugate / determinant (det >= 1: C, J are PSD).  n > 2:
// Gauss-Jordan WITHOUT pivoting - P is not symmetric and a pivot can in principle come out small or
// negative although det P >= 1, so the smallest pivot magnitude is reported and the caller falls back
// to the sequential sweep when it is not comfortably away from zero (never seen on the configs).
template <int n>
__device__ __forceinline__ void ric_invert(const double (&P)[n][n], double (&Mi)[n][n], double& min_pivot) {
  if constexpr (n == 2) {
    const double idet = fast_rcp(P[0][0] * P[1][1] - P[0][1] * P[1][0]);
    Mi[0][0] = P[1][1] * idet; Mi[0][1] = -P[0][1] * idet; Mi[1][0] = -P[1][0] * idet; Mi[1][1] = P[0][0] * idet;
  } else {
    double a[n][n];
#pragma unroll
    for (int i = 0; i < n; ++i)
#pragma unroll
      for (int j = 0; j < n; ++j) { a[i][j] = P[i][j]; Mi[i][j] = (i == j) ? 1.0 : 0.0; }
#pragma unroll
    for (int k = 0; k < n; ++k) {
      min_pivot = fmin(min_pivot, fabs(a[k][k]));
      const double ip = fast_rcp(a[k][k]);
#pragma unroll
      for (int j = 0; j < n; ++j) { a[k][j] *= ip; Mi[k][j] *= ip; }   // (columns < k of a, > k of Mi: zeros, folded)
#pragma unroll
      for (int i = 0; i < n; ++i) {
        if (i == k) continue;
        const double f = a[i][k];
#pragma unroll
        for (int j = 0; j < n; ++j) { a[i][j] = fma(-f, a[k][j], a[i][j]); Mi[i][j] = fma(-f, Mi[k][j], Mi[i][j]); }
      }
    }
  }
}

// out = ei (x) ej   (ei earlier in time); out may alias neither input.  VALUE_ONLY: just (J, eta) -
// all that is read after the last level of the scan.
template <int n, bool VALUE_ONLY = false>
__device__ __forceinline__ void ric_combine(RicElem<n>& o, const RicElem<n>& ei, const RicElem<n>& ej, double& min_pivot) {
  double P[n][n], Mi[n][n], W[n][n], MA[n][n];
#pragma unroll
  for (int i = 0; i < n; ++i)
#pragma unroll
    for (int j = 0; j < n; ++j) {
      double s = (i == j) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < n; ++k) s += ei.C[i][k] * ej.J[k][j];
      P[i][j] = s;                                          // I + C_i J_j
    }
  ric_invert<n>(P, Mi, min_pivot);
#pragma unroll
  for (int i = 0; i < n; ++i)
#pragma unroll
    for (int j = 0; j < n; ++j) {
      double s = 0.0, q = 0.0;
#pragma unroll
      for (int k = 0; k < n; ++k) { if (!VALUE_ONLY) s += ej.A[i][k] * Mi[k][j]; q += Mi[i][k] * ei.A[k][j]; }
      W[i][j] = s;                                          // A_j M
      MA[i][j] = q;                                         // M A_i ; V = MA^T
    }
  double t1[n], t2[n], WC[n][n], VJ[n][n];
#pragma unroll
  for (int i = 0; i < n; ++i) {
    double s = ei.b[i], q = ej.e[i];
#pragma unroll
    for (int k = 0; k < n; ++k) { if (!VALUE_ONLY) s += ei.C[i][k] * ej.e[k]; q -= ej.J[i][k] * ei.b[k]; }
    t1[i] = s;                                              // b_i + C_i eta_j
    t2[i] = q;                                              // eta_j - J_j b_i
  }
#pragma unroll
  for (int i = 0; i < n; ++i)
#pragma unroll
    for (int j = 0; j < n; ++j) {
      double s = 0.0, q = 0.0;
#pragma unroll
      for (int k = 0; k < n; ++k) { if (!VALUE_ONLY) s += W[i][k] * ei.C[k][j]; q += MA[k][i] * ej.J[k][j]; }
      WC[i][j] = s;                                         // W C_i
      VJ[i][j] = q;                                         // V J_j
    }
#pragma unroll
  for (int i = 0; i < n; ++i) {
    double sb = ej.b[i], se = ei.e[i];
#pragma unroll
    for (int k = 0; k < n; ++k) { if (!VALUE_ONLY) sb += W[i][k] * t1[k]; se += MA[k][i] * t2[k]; }
    if (!VALUE_ONLY) o.b[i] = sb;
    o.e[i] = se;
#pragma unroll
    for (int j = 0; j < n; ++j) {
      if (!VALUE_ONLY) {
        double sa = 0.0;
#pragma unroll
        for (int k = 0; k < n; ++k) sa += W[i][k] * ei.A[k][j];
        o.A[i][j] = sa;
      }
      if (j >= i) {                                          // C and J are symmetric: upper triangle, mirrored
        double sc = ej.C[i][j], sj = ei.J[i][j];
#pragma unroll
        for (int k = 0; k < n; ++k) { if (!VALUE_ONLY) sc += WC[i][k] * ej.A[j][k]; sj += VJ[i][k] * ei.A[k][j]; }
        if (!VALUE_ONLY) { o.C[i][j] = sc; o.C[j][i] = sc; }
        o.J[i][j] = sj; o.J[j][i] = sj;
      }
    }
  }
}

// value of `v` in lane `src` (any lane; out-of-range callers mask the result)
__device__ __forceinline__ double lane_read_f64(double v, int src) {
  union { double d; int i[2]; } u, r;
  u.d = v;
  r.i[0] = __builtin_amdgcn_ds_bpermute(src << 2, u.i[0]);
  r.i[1] = __builtin_amdgcn_ds_bpermute(src << 2, u.i[1]);
  return r.d;
}

// One Kogge-Stone level of the scan over the lanes, operands moved with DPP: `dst` = the element of
// the source lane, or the IDENTITY element where there is none (the moves deliver zeros there, and
// the identity is all zeros once A is sent as A - I).
template <int CTRL, int ROWS, int n, bool VALUE_ONLY = false>
__device__ __forceinline__ void ric_fetch_dpp(RicElem<n>& dst, const RicElem<n>& src) {
#pragma unroll
  for (int i = 0; i < n; ++i) {
    dst.e[i] = dpp_f64_or_zero<CTRL, ROWS>(src.e[i]);
    if (!VALUE_ONLY) dst.b[i] = dpp_f64_or_zero<CTRL, ROWS>(src.b[i]);
#pragma unroll
    for (int j = 0; j < n; ++j) {
      const double idm = (i == j) ? 1.0 : 0.0;
      if (!VALUE_ONLY) dst.A[i][j] = dpp_f64_or_zero<CTRL, ROWS>(src.A[i][j] - idm) + idm;
      if (j >= i) {                                          // symmetric blocks: upper triangle only
        if (!VALUE_ONLY) { dst.C[i][j] = dpp_f64_or_zero<CTRL, ROWS>(src.C[i][j]); dst.C[j][i] = dst.C[i][j]; }
        dst.J[i][j] = dpp_f64_or_zero<CTRL, ROWS>(src.J[i][j]); dst.J[j][i] = dst.J[i][j];
      }
    }
  }
}

template <class M>
__device__ inline bool backward_scan(const WS& w, const Consts<M>& c) {
  constexpr int n = M::n, m = M::m;
  static_assert(m == 1, "rank-one control term: C = fu luu^-1 fu^T with scalar luu");
  double min_pivot = __builtin_inf();                       // smallest pivot of the n > 2 inversions
  using Ly = Lay<n, m>;
  const int N = w.N, lane = threadIdx.x & 63;
  const int chunk = (N + 63) >> 6;                          // elements 0..N-2 are steps, element N-1 is the terminal one
  // chunks are dealt to the lanes in REVERSE time order (lane 63 owns steps 0..chunk-1), so that the
  // suffix scan over time is a prefix scan over the lanes - the direction DPP row shifts and row
  // broadcasts move data in
  const int e0 = (63 - lane) * chunk;
  double Q2[n][n], R2[m][m];
#pragma unroll
  for (int i = 0; i < n; ++i)
#pragma unroll
    for (int j = 0; j < n; ++j) Q2[i][j] = 2.0 * c.Q[i][j];
  R2[0][0] = 2.0 * c.R[0][0];
  const double R2i = fast_rcp(R2[0][0]);
  // ---- (1) this lane's chunk aggregate
  // Each lane reads the records of its OWN consecutive steps: the lanes' addresses are 4 records =
  // 256 bytes apart, i.e. all in the same LDS banks (a 64-way conflict: ~64 cycles per 8-byte read).
  // So every step's operands are read exactly once, here, and kept in registers for phase (3) -
  // the wave runs alone on its SIMD at these batch sizes, registers are free.
  constexpr int CHMAX = 4;
  struct Raw { double x[n], u, fx[n][n], fu[n]; };
  Raw raw[CHMAX];
  const bool held = n == 2 && chunk <= CHMAX;               // longer horizons, larger n: phase (3) re-reads
  auto read_raw = [&](Raw& r, int t) __attribute__((always_inline)) {
    const int tg = t < N ? t : N - 1, tj = t < N - 1 ? t : N - 2;
    const double* g = w.G + tg * Ly::GS;
    const double* jr = w.J + tj * Ly::JS;
#pragma unroll
    for (int i = 0; i < n; ++i) {
      r.x[i] = g[Ly::XB + i];
      r.fu[i] = jr[Ly::FU + i * m];
#pragma unroll
      for (int j = 0; j < n; ++j) r.fx[i][j] = jr[Ly::FX + i * n + j];
    }
    r.u = g[Ly::UB];
  };
  // A step element, the terminal element (0, 0, 0, -lf_x, lf_xx; ilqr.py:203-204) or, past the horizon, the
  // identity.  Wave-uniform fast path: when no lane that holds real elements is at (or past) the terminal one,
  // every lane builds a plain step element - no selects (they were 28 v_cndmask per build, 4 builds per pass).
  // Lanes whose whole chunk lies past the horizon build garbage there; their aggregate is reset to the identity
  // after the local compositions (`pad_fix`).  Otherwise: field-by-field selects.
  const bool all_pad = e0 >= N;
  auto element = [&](RicElem<n>& r, const Raw& q, int t) __attribute__((always_inline)) {
    const bool is_step = t < N - 1, is_term = t == N - 1;
    // (n = 2 only: the n = 3..4 scan lives at the edge of its 512 registers, where the extra branch costs more
    // than the selects it saves - C4's pass 47 -> 66 k cycles when it was tried there)
    if (n == 2 && !__any(!is_step && !all_pad)) {
#pragma unroll
      for (int i = 0; i < n; ++i) {
        double s = -c.qn[i];
#pragma unroll
        for (int j = 0; j < n; ++j) {
          s += Q2[i][j] * q.x[j];
          r.A[i][j] = q.fx[i][j];
          r.J[i][j] = Q2[i][j];
          r.C[i][j] = (q.fu[i] * R2i) * q.fu[j];
        }
        r.e[i] = -s;
        r.b[i] = -q.fu[i] * q.u;
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < n; ++i) {
      double s = -c.qn[i], sf = -c.qfn[i];
#pragma unroll
      for (int j = 0; j < n; ++j) {
        s += Q2[i][j] * q.x[j];
        sf += (2.0 * c.Qf[i][j]) * q.x[j];
        const double idm = (i == j) ? 1.0 : 0.0;
        r.A[i][j] = is_step ? q.fx[i][j] : (is_term ? 0.0 : idm);
        r.J[i][j] = is_step ? Q2[i][j] : (is_term ? 2.0 * c.Qf[i][j] : 0.0);
        r.C[i][j] = is_step ? (q.fu[i] * R2i) * q.fu[j] : 0.0;
      }
      r.e[i] = is_step ? -s : (is_term ? -sf : 0.0);         // -lx_t | -lf_x
      r.b[i] = is_step ? -q.fu[i] * q.u : 0.0;               // -fu luu^{-1} lu = -fu u_bar
    }
  };
  auto pad_fix = [&](RicElem<n>& r) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < n; ++i) {
      r.b[i] = all_pad ? 0.0 : r.b[i]; r.e[i] = all_pad ? 0.0 : r.e[i];
#pragma unroll
      for (int j = 0; j < n; ++j) {
        r.A[i][j] = all_pad ? ((i == j) ? 1.0 : 0.0) : r.A[i][j];
        r.C[i][j] = all_pad ? 0.0 : r.C[i][j];
        r.J[i][j] = all_pad ? 0.0 : r.J[i][j];
      }
    }
  };
  RicElem<n> S, T, U;
  if (held) {
#pragma unroll
    for (int k = 0; k < CHMAX; ++k) {
      if (k < chunk) {
        read_raw(raw[k], e0 + k);
        if (k == 0) element(S, raw[0], e0);
        else { element(T, raw[k], e0 + k); ric_combine<n>(U, S, T, min_pivot); S = U; }
      }
    }
  } else {
    Raw q;
    read_raw(q, e0);
    element(S, q, e0);
    for (int k = 1; k < chunk; ++k) {
      read_raw(q, e0 + k);
      element(T, q, e0 + k);
      ric_combine<n>(U, S, T, min_pivot);
      S = U;
    }
  }
  if constexpr (n == 2) pad_fix(S);
  // ---- (2) inclusive scan: S_l <- g_l (x) g_{l-1} (x) ... (x) g_0  (lane l-1 holds the LATER chunk).
  // Combining with the identity reproduces the left operand exactly, so lanes without a partner need
  // no select.  Four levels inside the 16-lane rows, then the row totals into the following rows.
  auto level = [&](auto ctrl, auto rows) __attribute__((always_inline)) {
    ric_fetch_dpp<decltype(ctrl)::value, decltype(rows)::value, n>(T, S);
    ric_combine<n>(U, S, T, min_pivot);
    S = U;
  };
  using std::integral_constant;
  level(integral_constant<int, 0x111>{}, integral_constant<int, 0xF>{});   // row_shr:1
  level(integral_constant<int, 0x112>{}, integral_constant<int, 0xF>{});   // row_shr:2
  level(integral_constant<int, 0x114>{}, integral_constant<int, 0xF>{});   // row_shr:4
  level(integral_constant<int, 0x118>{}, integral_constant<int, 0xF>{});   // row_shr:8
  level(integral_constant<int, 0x142>{}, integral_constant<int, 0xA>{});   // row_bcast:15 -> rows 1, 3
  ric_fetch_dpp<0x143, 0xC, n, true>(T, S);                                // row_bcast:31 -> rows 2, 3: only
  ric_combine<n, true>(U, S, T, min_pivot);                                // (J, eta) are read from here on
#pragma unroll
  for (int i = 0; i < n; ++i) {
    S.e[i] = U.e[i];
#pragma unroll
    for (int j = 0; j < n; ++j) S.J[i][j] = U.J[i][j];
  }
  if constexpr (n > 2) {
    // an inversion without pivoting met a small pivot somewhere in the wave: the caller redoes the pass
    // with the sequential sweep (nothing has been written yet)
    if (!__all(min_pivot > 1e-3)) return false;
  }
  // value function at the right edge of this chunk = (J, -eta) of the scan value one lane down
  double Vx[n], Vxx[n][n];
#pragma unroll
  for (int i = 0; i < n; ++i) {
    Vx[i] = -dpp_f64_or_zero<0x138, 0xF>(S.e[i]);                          // wave_shr:1
#pragma unroll
    for (int j = 0; j < n; ++j) Vxx[i][j] = dpp_f64_or_zero<0x138, 0xF>(S.J[i][j]);
  }
  // ---- (3) the reference recursion over this lane's own steps
  int t_hi = e0 + chunk - 1;                                // last element of the chunk
  if (t_hi >= N - 1) {                                      // chunk holds the terminal element: start from it
    t_hi = N - 2;
    const double* gT = w.G + (N - 1) * Ly::GS;
#pragma unroll
    for (int i = 0; i < n; ++i) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < n; ++j) { s += (2.0 * c.Qf[i][j]) * gT[Ly::XB + j]; Vxx[i][j] = 2.0 * c.Qf[i][j]; }
      Vx[i] = s - c.qfn[i];
    }
  }
  auto riccati_step = [&](const Raw& q, int t) __attribute__((always_inline)) {
    BRegs<M> r;
#pragma unroll
    for (int i = 0; i < n; ++i) {
      double s = -c.qn[i];
#pragma unroll
      for (int j = 0; j < n; ++j) { s += Q2[i][j] * q.x[j]; r.fx[i][j] = q.fx[i][j]; }
      r.lx[i] = s;
      r.fu[i][0] = q.fu[i];
    }
    r.lu[0] = R2[0][0] * q.u;
    backward_step<M>(r, c, Q2, R2, Vx, Vxx, w.G + t * Ly::GS);
  };
  if (held) {
#pragma unroll
    for (int k = CHMAX - 1; k >= 0; --k) {
      if (k < chunk && e0 + k <= t_hi) riccati_step(raw[k], e0 + k);
    }
  } else {
    for (int t = t_hi; t >= e0; --t) {
      Raw q;
      read_raw(q, t);
      riccati_step(q, t);
    }
  }
  return true;
}

// n = 3..4: the scan keeps three n x n elements and the temporaries of a composition live - all 512
// registers of a wave.  Inlined into the kernel that allocation drags the line-search loops down with
// it (C4: line search 208 k -> 259 k cycles per iteration), so it is a real function with its own
// register allocation: it rebuilds its view of the workgroup's LDS and reads the cost constants from
// their LDS image.
template <class M>
__device__ __attribute__((noinline)) bool backward_scan_outlined(int N, int n_store) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const WS w = carve<M::n, M::m>(smem, N, n_store);
  Consts<M> c;
  c.from_lds(w.cst);
  return backward_scan<M>(w, c);
}

// `mode` (KArgs::seq_backward): 0 the fastest form; 1 sequential in time (A/B measurements); 2 the
// reference's recursion verbatim (backward_scalar: no use of Vxx = Vxx^T, no scan elements that assume
// symmetric positive semi-definite Q/Qf) - chosen by the host for asymmetric or indefinite cost matrices,
// which the reference accepts without symmetrizing (ilqr.py:182,653-667).  For n = 3..4 that form lives in
// its own kernel instantiations (ExactCost<M>, picked by the host like LongHorizon<M>), so the regular
// kernels carry neither its code nor its registers.
template <class M>
__device__ inline void backward(const WS& w, const Consts<M>& c, int mode = 0) {
  const bool sequential = mode != 0;
  if constexpr (UsesExactBackward<M>::value) {
    backward_scalar<M>(w, c);
  } else
  if constexpr (M::n >= 3 && M::n <= 4 && M::m == 1) {
    // LongHorizon<M> kernels (N > 128, two or more steps per lane): the scan's 9 compositions of n x n
    // elements (~640 fused multiply-adds each at n = 4) beat N - 1 sequential MFMA steps
    if constexpr (UsesScanBackward<M>::value) {
      // (N > 128 always holds for these kernels.  The loop-invariant test is kept on purpose: LLVM unswitches
      // the solve loop on it, and the copy of the loop that contains the call then keeps the line-search
      // loops' register allocation - measured 202 k vs 238 k cycles of line search per C4 iteration.)
      if (!sequential && w.N > 128 && backward_scan_outlined<M>(w.N, w.n_store)) return;
    }
    backward_mfma<M>(w, c);
  }
  else if constexpr (M::n == 2 && M::m == 1) { if (sequential) backward_scalar<M>(w, c); else backward_scan<M>(w, c); }
  else backward_scalar<M>(w, c);
}

// ---------------------------------------------------------------------------
// Batch statistics without a second kernel: every workgroup publishes its problem's results, takes a
// ticket, and the holder of the last ticket - all other results are then visible - reduces the B
// per-problem records (same rules as stats_kernel: best cost among the converged problems, ties to
// the lower index) and writes the aggregate.  Called by the main wave, all 64 lanes.
// ---------------------------------------------------------------------------
// The four per-problem records travel as device-scope atomic stores / loads (write-through sc1 stores,
// coherent across the eight XCDs' L2 caches on their own) and the ticket is taken once they are
// ACKNOWLEDGED: an explicit s_waitcnt vmcnt(0) between the stores and the ticket's atomic add (stores
// count in vmcnt on gfx9; a workgroup-scope fence alone emits no wait, and the stores and the counter
// live in different L2 channels).  NOT a device-scope release fence: that would write back the whole
// L2, i.e. wait for the results other workgroups are streaming out at that moment (measured: +12 us per
// launch).
__device__ inline void batch_stats_by_last_workgroup(const KArgs& a, int b, double L, int iters, int status, int ls_total) {
  const int lane = threadIdx.x & 63;
  int ticket = 0;
  if (lane == 0) {
    __hip_atomic_store(a.cost + b, L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(a.iters + b, iters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(a.status + b, status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(a.ls_trials + b, ls_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // compiler ordering; no cache write-back
    __builtin_amdgcn_s_waitcnt(0x0F70);                        // vmcnt(0): the four write-through stores are acknowledged
    ticket = __hip_atomic_fetch_add(a.done_counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  ticket = __builtin_amdgcn_readfirstlane(ticket);
  if (ticket != (int)gridDim.x - 1) return;
  const int B = a.B;
  long long it = 0, ls = 0;
  int c = 0, mxi = 0, nm = 0, nf = 0, bi = -1;
  double bc = __builtin_inf();
  constexpr int U = 4;                                         // 4 x 4 independent loads in flight per lane
  for (int q0 = lane; q0 < B; q0 += 64 * U) {
    int iq[U], sq[U], lq[U]; double cq[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int q = (q0 + 64 * u < B) ? q0 + 64 * u : 0;
      iq[u] = __hip_atomic_load(a.iters + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      sq[u] = __hip_atomic_load(a.status + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      lq[u] = __hip_atomic_load(a.ls_trials + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      cq[u] = __hip_atomic_load(a.cost + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int q = q0 + 64 * u;
      if (q < B) {
        it += iq[u]; ls += lq[u];
        mxi = iq[u] > mxi ? iq[u] : mxi;
        if (sq[u] == MI_STATUS_CONVERGED) { c++; if (cq[u] < bc) { bc = cq[u]; bi = q; } }
        else if (sq[u] == MI_STATUS_MAX_ITERS) nm++;
        else nf++;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    it += __shfl_xor(it, o); ls += __shfl_xor(ls, o);
    c += __shfl_xor(c, o); nm += __shfl_xor(nm, o); nf += __shfl_xor(nf, o);
    const int m2 = __shfl_xor(mxi, o); mxi = m2 > mxi ? m2 : mxi;
    const double bc2 = __shfl_xor(bc, o); const int bi2 = __shfl_xor(bi, o);
    if (bc2 < bc || (bc2 == bc && bi2 >= 0 && (bi < 0 || bi2 < bi))) { bc = bc2; bi = bi2; }
  }
  if (lane == 0) {
    DevStats* o = a.stats_out;
    o->total_iters = it; o->total_ls = ls; o->n_conv = c; o->n_max = nm; o->n_fail = nf;
    o->max_iters_seen = mxi; o->best_index = bi; o->best_cost = bc; o->n_internal = 0; o->n_not_pd = 0;   // (these kernels have no such exits)
    __hip_atomic_store(a.done_counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
  }
}

// ---------------------------------------------------------------------------
// The kernel: stage, run MODE, write back.
// ---------------------------------------------------------------------------
template <class M, int JAC, int MODE>
__global__ void __launch_bounds__(256) ilqr_small_kernel(const KArgs a) {
  constexpr int n = M::n, m = M::m;
  using Ly = Lay<n, m>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef MI_PROF_NEWTON
  const long long c_kstart = clock64();
#endif
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  const int N = a.N;
  WS w = carve<n, m>(smem, N, a.n_store);
  // Optional helper wavefronts (threads 64.., MODE_SOLVE / MODE_MPC with every step a key-point):
  // they only ever run helper_wave() - their share of the linearization.
  const int team = ((MODE == MODE_SOLVE || MODE == MODE_MPC) && a.helpers > 0) ? 1 + a.helpers : 1;
  if (team > 1 && threadIdx.x >= 64) {
    helper_wave<M, JAC>(w, a, __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), team);
    return;
  }
  lds_vint_t* team_cmd = (lds_vint_t*)w.aux;
  // setInterval with minN = 1 (the default, ilqr.py:396): every step is a key-point, no list, no interpolation
  const bool every_step = (MODE == MODE_SOLVE || MODE == MODE_MPC) && a.kp_method == MI_KP_SET_INTERVAL && a.minN == 1;
  // what the time-parallel rollout folds in (rollout_newton_impl): cost + commit, and the linearization
  // unless helper wavefronts share that
  const int fuse = (MODE == MODE_SOLVE || MODE == MODE_MPC) ? ((every_step && team == 1) ? 2 : 1) : 0;
  const size_t oX = (size_t)b * n * N, oU = (size_t)b * m * (N - 1), oK = (size_t)b * m * n * (N - 1);
  const size_t oFx = (size_t)b * n * n * (N - 1), oFu = (size_t)b * n * m * (N - 1), oT = (size_t)b * (N - 1);

  const bool cold = a.cold != 0;
  Consts<M> c;
  double x0r[n];
  if constexpr (n <= 2) {
    // Everything that comes from HBM is requested up front - cost matrices, x0, the first 256 steps of
    // the control sequence - and consumed after the LDS arrays have been initialized: one memory
    // latency instead of one per array (a cold solve reads nothing else).  (n = 2 only: in the
    // larger kernels the extra live registers cost more in the line-search loops than this saves.)
    c.load(a.costmat);
#pragma unroll
    for (int i = 0; i < n; ++i) x0r[i] = a.x0[(size_t)b * n + i];
    constexpr int UPQ = 4;
    const double* usrc = (a.u_pending ? a.u_guess : a.u_bar) + oU;
    double upre[m][UPQ];
#pragma unroll
    for (int r = 0; r < m; ++r)
#pragma unroll
      for (int q = 0; q < UPQ; ++q) {
        const int t = lane + 64 * q;
        upre[r][q] = (t < N - 1) ? usrc[(size_t)r * (N - 1) + t] : 0.0;
      }
    if (cold) {
      // all-zero solver state: whole records at once (one pass over t instead of one per array row)
      for (int t = lane; t < N; t += 64) {
        double* g = w.G + t * Ly::GS;
        double* jr = w.J + t * Ly::JS;
#pragma unroll
        for (int k = 0; k < Ly::GS; ++k) g[k] = 0.0;
#pragma unroll
        for (int k = 0; k < Ly::JS; ++k) jr[k] = 0.0;
      }
    } else {
      stage_in(w.G, Ly::GS, Ly::XB, a.x_bar + oX, n, N, false);
      stage_in(w.G, Ly::GS, Ly::KK, a.K + oK, m * n, N - 1, false);
      stage_in(w.G, Ly::GS, Ly::KAP, a.kappa + oU, m, N - 1, false);
      stage_in(w.G, Ly::GS, Ly::DV, a.dV + oT, 1, N - 1, false);
      stage_in(w.J, Ly::JS, Ly::FX, a.fx + oFx, n * n, N - 1, false);
      stage_in(w.J, Ly::JS, Ly::FU, a.fu + oFu, n * m, N - 1, false);
    }
    wave_sync();
#pragma unroll
    for (int r = 0; r < m; ++r) {
#pragma unroll
      for (int q = 0; q < UPQ; ++q) {
        const int t = lane + 64 * q;
        if (t < N - 1) w.G[t * Ly::GS + Ly::UB + r] = upre[r][q];
      }
      for (int t = lane + 64 * UPQ; t < N - 1; t += 64) w.G[t * Ly::GS + Ly::UB + r] = usrc[(size_t)r * (N - 1) + t];
    }
  } else {
    stage_in(w.G, Ly::GS, Ly::XB, a.x_bar + oX, n, N, cold);
    stage_in(w.G, Ly::GS, Ly::UB, (a.u_pending ? a.u_guess : a.u_bar) + oU, m, N - 1, false);
    stage_in(w.G, Ly::GS, Ly::KK, a.K + oK, m * n, N - 1, cold);
    stage_in(w.G, Ly::GS, Ly::KAP, a.kappa + oU, m, N - 1, cold);
    stage_in(w.G, Ly::GS, Ly::DV, a.dV + oT, 1, N - 1, cold);
    stage_in(w.J, Ly::JS, Ly::FX, a.fx + oFx, n * n, N - 1, cold);
    stage_in(w.J, Ly::JS, Ly::FU, a.fu + oFu, n * m, N - 1, cold);
    c.load(a.costmat);
#pragma unroll
    for (int i = 0; i < n; ++i) x0r[i] = a.x0[(size_t)b * n + i];
    if constexpr (UsesScanBackward<M>::value) c.to_lds(w.cst);
  }
  wave_sync();

  if (MODE == MODE_ROLLOUT) {
    double L, ex;
    rollout<M>(w, c, a, x0r, a.stage_in[b], lane == 0 ? 0 : -1, L, ex);
    wave_sync();
    stage_out(a.x_trial + oX, w.T, Ly::TS, Ly::XN, n, N);
    stage_out(a.u_trial + oU, w.T, Ly::TS, Ly::UN, m, N - 1);
    if (lane == 0) { a.trial_cost[2 * b] = L; a.trial_cost[2 * b + 1] = ex; }
    return;
  }
  if (MODE == MODE_LINEARIZE) {
    const int nk = linearize<M, JAC>(w, a);
    stage_out(a.fx + oFx, w.J, Ly::JS, Ly::FX, n * n, N - 1);
    stage_out(a.fu + oFu, w.J, Ly::JS, Ly::FU, n * m, N - 1);
    for (int i = lane; i < nk; i += 64) a.kp_list[(size_t)b * (N - 1) + i] = w.kp[i];
    if (lane == 0) a.kp_count[b] = nk;
    return;
  }
  if (MODE == MODE_BACKWARD) {
    backward<M>(w, c, a.seq_backward);
    wave_sync();
    stage_out(a.K + oK, w.G, Ly::GS, Ly::KK, m * n, N - 1);
    stage_out(a.kappa + oU, w.G, Ly::GS, Ly::KAP, m, N - 1);
    stage_out(a.dV + oT, w.G, Ly::GS, Ly::DV, 1, N - 1);
    return;
  }

  // MODE_SOLVE / MODE_FORWARD / MODE_MPC: the Solve loop (ilqr.py:680-708); MODE_MPC wraps it in
  // the receding-horizon loop with the solver state staying in LDS between re-solves.
  double L = (MODE == MODE_FORWARD) ? a.stage_in[b] : __builtin_inf();
  int iters = 0, ls_total = 0, nk = 0;
  int status = MI_STATUS_CONVERGED;
  double* hist = a.hist + (size_t)b * a.hist_cap * 4;
  // the reference's stopwatches (time_fp / time_getDerivs / time_backwardsPass, ilqr.py:364-372,696-699)
  long long c_ls = 0, c_lin = 0, c_bp = 0;
  const long long c_begin = clock64();
  const int n_solves = (MODE == MODE_MPC) ? a.mpc_resolves : 1;
  for (int rs = 0; rs < n_solves; ++rs) {
    if (MODE == MODE_MPC) {
      // warm start (acrobot.py:147-152): x0 <- x_bar[:, replan]; u_bar <- [u_bar[:, replan:], repeat(last)]
      const int r = a.mpc_replan;
      constexpr int TPL = 8;                       // time steps per lane held in registers (N <= 512)
      double ush[TPL][m];
#pragma unroll
      for (int q = 0; q < TPL; ++q) {
        const int t = lane + 64 * q;
        const int src = (t + r < N - 1) ? t + r : N - 2;
#pragma unroll
        for (int k = 0; k < m; ++k) ush[q][k] = (t < N - 1) ? w.G[src * Ly::GS + Ly::UB + k] : 0.0;
      }
#pragma unroll
      for (int i = 0; i < n; ++i) x0r[i] = w.G[r * Ly::GS + Ly::XB + i];
      wave_sync();
#pragma unroll
      for (int q = 0; q < TPL; ++q) {
        const int t = lane + 64 * q;
        if (t < N - 1) {
#pragma unroll
          for (int k = 0; k < m; ++k) w.G[t * Ly::GS + Ly::UB + k] = ush[q][k];
        }
      }
      // moving target (mini_cheetah.py:151-156) and the constants derived from it (ilqr.py:180,203)
#pragma unroll
      for (int i = 0; i < n; ++i) c.xnom[i] += a.mpc_target_step[i];
#pragma unroll
      for (int j = 0; j < n; ++j) {
        double s_ = 0.0, sf_ = 0.0;
#pragma unroll
        for (int i = 0; i < n; ++i) { s_ += (2.0 * c.xnom[i]) * c.Q[i][j]; sf_ += (2.0 * c.xnom[i]) * c.Qf[i][j]; }
        c.qn[j] = s_; c.qfn[j] = sf_;
      }
      if constexpr (UsesScanBackward<M>::value) c.to_lds(w.cst);
      wave_sync();
      L = __builtin_inf();
      status = MI_STATUS_CONVERGED;                // per re-solve, as a host loop of Solve() calls would leave it
    }
    double improvement = __builtin_inf();
    bool optimistic = true, first_try_streak = true;
    int it_this = 0;
    long long c_prev = 0;
    while (improvement > a.delta) {
      if (it_this >= a.max_iters) { status = MI_STATUS_MAX_ITERS; break; }
      double L_new, eps; int trials, slot = 0;
      // stopwatch reads cost an s_memtime round trip each: an iteration starts where the previous one
      // ended, and a rollout that linearized on the way has no separate linearization span
      const long long c0 = (it_this == 0) ? clock64() : c_prev;
      int fused = 0;
      // first iteration of the first solve on all-zero solver state (no gains, no nominal trajectory)
      const bool cold_start = cold && rs == 0 && it_this == 0 && (MODE == MODE_SOLVE || MODE == MODE_MPC);
      const bool ok = linesearch<M, JAC>(w, c, a, x0r, L, optimistic, fuse, L_new, eps, trials, slot, fused, cold_start);
      // expect eps = 1 next time if it was accepted now - or whenever the attempt is the cheap one
      // (sequential attempts: only after TWO first-trial acceptances in a row - on coarse line
      // searches a lone one is usually followed by a backtrack, and the failed attempt costs a rollout)
      optimistic = (ok && trials == 1 && first_try_streak) || newton_capable<M>(w, a);
      first_try_streak = ok && trials == 1;
      ls_total += trials;
      if (!ok) { status = MI_STATUS_LINESEARCH_FAILED; break; }
      wave_sync();
      const long long c1 = (fused == 2) ? c0 : clock64();
      if (fused == 0) commit_trial<n, m>(w, slot);                // u_bar <- u, x_bar <- x (:375-376)
      wave_sync();
      if (fused == 2) {
        nk = N - 1;                                               // linearized by the rollout itself
      } else if (team > 1) {
        nk = N - 1;                                               // every step is a key-point (:396, :414)
        if (lane == 0) team_cmd[0] = TEAM_CMD_LINEARIZE;
        team_barrier();
        jac_rounds<M, JAC>(w, a, 0, team, lane);
        team_barrier();
      } else if (every_step) {                                    // same, without helpers or the key-point list
        nk = N - 1;
        jac_rounds<M, JAC>(w, a, 0, 1, lane);
        wave_sync();
      } else {
        nk = linearize<M, JAC>(w, a);                             // at the ACCEPTED trajectory (:370)
      }
      const long long c2 = clock64();
      if (MODE != MODE_FORWARD) { backward<M>(w, c, a.seq_backward); wave_sync(); } // :697
      // LongHorizon kernels: backward() is a real call; re-reading the constants from their LDS image instead
      // of keeping 90 registers alive across it leaves the line-search loops their old allocation
      if constexpr (UsesScanBackward<M>::value) c.from_lds(w.cst);
      const long long c3 = clock64();
      c_prev = c3;
      if (fused == 2) c_ls += c2 - c0;                            // line search + commit + linearization
      else { c_ls += c1 - c0; c_lin += c2 - c1; }
      c_bp += c3 - c2;
      if (lane == 0 && it_this < a.hist_cap) {                    // history of the LAST solve
        hist[4 * it_this + 0] = L_new; hist[4 * it_this + 1] = eps;
        hist[4 * it_this + 2] = (double)trials; hist[4 * it_this + 3] = (double)nk / (double)(N - 1) * 100.0;   // :406
        // the reference's per-iteration stopwatches (ilqr.py:364-372, 696-702), in shader-clock cycles
        double* ic = a.iter_cyc + ((size_t)b * a.hist_cap + it_this) * 4;
        ic[0] = (double)((fused == 2 ? c2 : c1) - c0); ic[1] = (fused == 2) ? 0.0 : (double)(c2 - c1);
        ic[2] = (double)(c3 - c2); ic[3] = (double)(c3 - c0);
#ifdef MI_PROF_NEWTON
        hist[4 * it_this + 0] = (double)(c1 - c0); hist[4 * it_this + 1] = mi_dbg_vals[0]; hist[4 * it_this + 2] = mi_dbg_vals[1]; hist[4 * it_this + 3] = mi_dbg_vals[2];
        ic[0] = mi_dbg_vals[3]; ic[1] = mi_dbg_vals[4]; ic[2] = mi_dbg_vals[5]; ic[3] = mi_dbg_vals[6];   // the sweeps' updates
        mi_dbg_vals[3] = mi_dbg_vals[4] = mi_dbg_vals[5] = mi_dbg_vals[6] = 0.0;
#endif
      }
      improvement = L - L_new;                                    // :706
      L = L_new;
      it_this += 1;
      if (MODE == MODE_FORWARD) break;
    }
    iters += it_this;
    if (MODE == MODE_MPC) {
      double* lg = a.mpc_log + ((size_t)b * a.mpc_resolves + rs) * (n + 2);
      if (lane == 0) {
#pragma unroll
        for (int i = 0; i < n; ++i) lg[i] = x0r[i];
        lg[n] = L; lg[n + 1] = (double)it_this;
      }
      if (status == MI_STATUS_LINESEARCH_FAILED) break;
    }
  }
  if (MODE == MODE_MPC && lane < n) {               // the re-solves moved x0: keep the HBM copy consistent
    double* x0g = const_cast<double*>(a.x0) + (size_t)b * n;
#pragma unroll
    for (int i = 0; i < n; ++i) if (lane == i) x0g[i] = x0r[i];
  }
  if (team > 1) {
    if (lane == 0) team_cmd[0] = TEAM_CMD_EXIT;
    team_barrier();
  }
  if (every_step && (MODE == MODE_SOLVE || MODE == MODE_MPC)) {
    for (int i = lane; i < N - 1; i += 64) w.kp[i] = i;          // what keypoints_set_interval(minN = 1) lists
  }
  wave_sync();
#ifdef MI_PROF_NEWTON
  const long long c_loop_end = clock64();
#endif
  // Batch statistics first, while this wave has next to nothing in flight (the ticket waits for the
  // wave's outstanding stores).
  if ((MODE == MODE_SOLVE || MODE == MODE_MPC) && a.stats_out != nullptr)
    batch_stats_by_last_workgroup(a, b, L, iters, status, ls_total);
  if constexpr (n <= 2) {
    // Write-back, one time step per lane: a step's whole G and J records are read from LDS together
    // (one LDS latency per 64 steps instead of one per array row), then scattered to the reference's
    // time-last arrays - every store instruction still writes 64 consecutive doubles.
    for (int tb = 0; tb < N; tb += 64) {
      const int t = tb + lane;
      if (t < N) {
        const double* g = w.G + t * Ly::GS;
        const double* jr = w.J + t * Ly::JS;                      // t = N-1: the pad record, read but not stored
        double xb[n], ub[m], kk[m * n], kap[m], dv, fxr[n * n], fur[n * m];
  #pragma unroll
        for (int i = 0; i < n; ++i) xb[i] = g[Ly::XB + i];
  #pragma unroll
        for (int k = 0; k < m; ++k) { ub[k] = g[Ly::UB + k]; kap[k] = g[Ly::KAP + k]; }
  #pragma unroll
        for (int k = 0; k < m * n; ++k) kk[k] = g[Ly::KK + k];
        dv = g[Ly::DV];
  #pragma unroll
        for (int k = 0; k < n * n; ++k) fxr[k] = jr[Ly::FX + k];
  #pragma unroll
        for (int k = 0; k < n * m; ++k) fur[k] = jr[Ly::FU + k];
  #pragma unroll
        for (int i = 0; i < n; ++i) wt_store(&a.x_bar[oX + (size_t)i * N + t], xb[i]);
        if (a.sink_x != nullptr) {                                   // result sink in host memory (optional)
  #pragma unroll
          for (int i = 0; i < n; ++i) a.sink_x[oX + (size_t)i * N + t] = xb[i];
          if (t < N - 1) {
  #pragma unroll
            for (int k = 0; k < m; ++k) a.sink_u[oU + (size_t)k * (N - 1) + t] = ub[k];
          }
        }
        if (t < N - 1) {
  #pragma unroll
          for (int k = 0; k < m; ++k) wt_store(&a.u_bar[oU + (size_t)k * (N - 1) + t], ub[k]);
  #pragma unroll
          for (int k = 0; k < n * n; ++k) wt_store(&a.fx[oFx + (size_t)k * (N - 1) + t], fxr[k]);
  #pragma unroll
          for (int k = 0; k < n * m; ++k) wt_store(&a.fu[oFu + (size_t)k * (N - 1) + t], fur[k]);
          if (MODE == MODE_SOLVE || MODE == MODE_MPC) {
  #pragma unroll
            for (int k = 0; k < m * n; ++k) wt_store(&a.K[oK + (size_t)k * (N - 1) + t], kk[k]);
  #pragma unroll
            for (int k = 0; k < m; ++k) wt_store(&a.kappa[oU + (size_t)k * (N - 1) + t], kap[k]);
            wt_store(&a.dV[oT + t], dv);
          }
        }
      }
    }
  } else {
    stage_out(a.x_bar + oX, w.G, Ly::GS, Ly::XB, n, N);
    stage_out(a.u_bar + oU, w.G, Ly::GS, Ly::UB, m, N - 1);
    if (a.sink_x != nullptr) {                                       // result sink in host memory (optional)
      stage_out(a.sink_x + oX, w.G, Ly::GS, Ly::XB, n, N);
      stage_out(a.sink_u + oU, w.G, Ly::GS, Ly::UB, m, N - 1);
    }
    stage_out(a.fx + oFx, w.J, Ly::JS, Ly::FX, n * n, N - 1);
    stage_out(a.fu + oFu, w.J, Ly::JS, Ly::FU, n * m, N - 1);
    if (MODE == MODE_SOLVE || MODE == MODE_MPC) {
      stage_out(a.K + oK, w.G, Ly::GS, Ly::KK, m * n, N - 1);
      stage_out(a.kappa + oU, w.G, Ly::GS, Ly::KAP, m, N - 1);
      stage_out(a.dV + oT, w.G, Ly::GS, Ly::DV, 1, N - 1);
    }
  }
  for (int i = lane; i < nk; i += 64) a.kp_list[(size_t)b * (N - 1) + i] = w.kp[i];
  if (lane == 0) {
    a.cost[b] = L; a.iters[b] = iters; a.status[b] = status; a.ls_trials[b] = ls_total; a.kp_count[b] = nk;
    if (a.sink_cost != nullptr) a.sink_cost[b] = L;
    a.prof[4 * b + 0] = c_ls; a.prof[4 * b + 1] = c_lin; a.prof[4 * b + 2] = c_bp; a.prof[4 * b + 3] = clock64() - c_begin;
#ifdef MI_PROF_NEWTON
    // debug build: launch phases instead of the stage stopwatches
    a.prof[4 * b + 0] = c_begin - c_kstart; a.prof[4 * b + 1] = c_loop_end - c_begin; a.prof[4 * b + 2] = clock64() - c_loop_end;
#endif
  }
}

}  // namespace mi
