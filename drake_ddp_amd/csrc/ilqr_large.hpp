// Workgroup-per-problem iLQR kernels for large state dimension (n ~ 36, m ~ 12), gfx950.
//
// One 256-thread workgroup (4 wavefronts, one per SIMD of a CU) owns one problem.
// The per-problem solver state of the reference (/root/reference/ilqr.py:70-83) is
// too large for LDS at this size (fx alone is n*n*(N-1)*8 = 404 KB at n=36,N=40), so
// it stays in HBM/L2 in a TIME-MAJOR layout ([t][row][col]: one time step's matrices
// are contiguous, loads are coalesced) and each sequential step stages what it needs
// through LDS:
//   rollout  (ilqr.py:306-327): K_t(x-x_bar) as 16-lane partial dot products (DPP row
//            reductions), one lane per degree of freedom for the dynamics, per-thread cost
//            partials reduced once per trial; operands prefetched two steps ahead;
//   linearize(ilqr.py:380-415): (key-point, column) items over the 256 threads,
//            central differences or forward-mode duals, only the dofs that read the
//            perturbed input when the model declares its sparsity; shared key-point code;
//   backward (ilqr.py:623-667): per step  T1 = Vxx F,  H = F^T T1  with  F = [fx | fu]
//            (one (n+m)x(n+m) product yields Qxx and Qux at once) as 16x16x4 fp64 MFMA tiles on three matrix-core
//            waves that keep their column tile in the accumulators from product to product, while a fourth "solver"
//            wave inverts Quu = 2R + fu^T Vxx fu (Gauss-Jordan, one row per lane, DPP broadcasts) beside them;
//            K = Quu^{-1} Qux and Vxx <- Qxx - Qux^T K again on the matrix core.  See large_backward().
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
constexpr int kPdFlag = 8;          // slot of the reduction scratch (LLay::oRed) where a backward pass leaves "a Quu was not positive definite"

// The thread index of a STAGE (a backward pass, a rollout, a linearization): see stage_lane (ilqr_small.hpp).
__device__ __forceinline__ int stage_tid() { return stage_lane(); }

template <int n, int m>
struct LLay {
  static constexpr int nm = n + m;
  // doubles
  static constexpr int QC = m * (m + 1) / 2;               // packed lower triangle of Quu
  static constexpr int T16 = 16;                           // MFMA tile edge
  static constexpr int NP = ((n + 15) / 16) * 16;          // n padded to whole tiles (rows of Vxx)
  static constexpr int KN = (n + 3) / 4, NK = 4 * KN;      // MFMA k-steps over a contraction of length n, n padded to them
  // Columns of the augmented matrices F = [fx | fu], T1, H.  COMPACT: u follows x directly and the last column tile
  // holds the tail of x together with all of u (n = 36, m = 12: three tiles).  SPLIT (that tile would not start inside
  // x, or n is not a multiple of 4: n = 37): x is padded to whole tiles and u gets a tile of its own - the pad
  // rows / columns are zero and never stored, so Quu still sits at the corner of the last diagonal tile.
  // MID (n <= 32: one or two row tiles, any m <= 16 - mid_backward): always split, and always 48 columns - a row stride
  // of 48 doubles keeps the four rows of a k-step on disjoint banks for the 64-bit reads (32 would put them on the same).
  static constexpr bool kMid = n <= 32;
  static constexpr int NMPc = ((nm + 15) / 16) * 16;
  // (COMPACT only when x's tail and u fill the last tile exactly - (36, 12), (40, 8): with pad columns behind u, Quu would not
  //  end at the tile's corner, which the solver wave's row mapping relies on; (36, 4), (36, 8), (40, 4) take the split layout)
  static constexpr bool kSplit = kMid || !(NMPc - 16 <= n && n % 4 == 0 && nm == NMPc);
  static constexpr int UC = kSplit ? NP : n;               // column of u_0
  static constexpr int NMP = kMid ? 48 : (kSplit ? NP + ((m + 15) / 16) * 16 : NMPc);
  static constexpr int TS = NMP + 4;                       // row stride of T1 / H: whole tiles + the Vx/first-order column
  static constexpr int VS = NK | 1;                        // odd row stride of Vxx: conflict-free column-of-tile reads
  static constexpr int oQ = 0, oQf = oQ + n * n, oR = oQf + n * n, oXnom = oR + m * m, oQn = oXnom + n,
                       oQfn = oQn + n, oVxx = oQfn + n, oVx = oVxx + NP * VS, oF = oVx + NK + (NK & 1),
                       oT1 = oF + NK * NMP, oH = oT1 + NK * TS, oXs = oH + NMP * TS, oUs = oXs + n,
                       oRed = oUs + m, oXb = oRed + kLargeThreads, oQc = oXb + n + m + ((n + m) & 1),
                       oQT = oQc + m * m + m + (m & 1), oS = oQT + (kSplit ? 0 : n * n + ((n * n) & 1)),
                       oEnd = oS + 16 * 17 + 1;                // 16x16 tile, odd row stride
  static constexpr size_t doubles = oEnd + 8;
};

// Integer scratch of the key-point code: five arrays of N (or 2 N) ints - and, while every step is a key-point, the home of the
// cluster hand-shake's state (aux[0..3], the leader's six counters / a helper's last round in `need`): each array is therefore
// at least kIntRowMin ints long.  (Round 5's last session moved that state from registers into these arrays; with N = 3 aux[3]
// WAS need[0] - the leader's round counter - and every clustered solve of a three-step horizon ended with MI_STATUS_INTERNAL.  Found
// in round 6 by running the GPU suite with clusters forced: test_shortest_horizons_vs_c_oracle swallowed the RuntimeError.)
constexpr int kIntRowMin = 8;
__host__ __device__ constexpr int int_row(int N) { return N > kIntRowMin ? N : kIntRowMin; }
template <int n, int m>
__host__ __device__ constexpr size_t large_lds_bytes(int N) {
  // fixed block + per-step cost gradients [N][n+m] + integer scratch of the key-point code
  return (LLay<n, m>::doubles + (size_t)N * (n + m)) * 8 + (size_t)7 * int_row(N) * 4 + 16;
}

// Horizons whose cost gradients do not fit next to the fixed block any more (N > 148 for (36, 12), > 319 for (27, 7)): the
// gradients go to HBM (KArgs::lxu) and LDS keeps the fixed block + the key-point scratch - a slower backward step (one L2 read
// of lx_t | lu_t per step on the wave that forms the first-order column), but no horizon limit short of 160 KB of integers.
template <int n, int m>
__host__ __device__ constexpr size_t large_lds_bytes_hbm(int N) {
  return (size_t)LLay<n, m>::doubles * 8 + (size_t)7 * int_row(N) * 4 + 16;
}

// Per-problem views of the time-major HBM arrays.
template <int n, int m>
struct LView {
  double *X, *U, *K, *kap, *dV, *Fx, *Fu, *Xn, *Un;
  double* LxG;      // the cost gradients [N-1][n+m] in HBM (long horizons), or nullptr: in LDS behind the fixed block
  int N;
  int pd_continue;  // a Quu that is not positive definite: 1 = invert it with partial pivoting and carry on like np.linalg.inv (ilqr.py:655)
  int asym;         // Q, R or Qf is not symmetric (mid-size kernels only): the reference's recursion without any use of symmetry
};

// lx_t | lu_t: LDS (the address space stays known to the compiler) or HBM, chosen per launch
// (the HBM side as relaxed device-scope atomics: plain accesses let the compiler fold the two branches into ONE access through a
//  selected flat pointer - slower for the LDS case, and the address-space cast it needs trips an instruction-selection error
//  of this hipcc on some instantiations: "V_CMP_NE_U32 0, $src_shared_base")
template <class V>
__device__ __forceinline__ void lxu_store(const V& v, double* lds_area, int idx, double val) {
  if (v.LxG) __hip_atomic_store(v.LxG + idx, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else lds_area[idx] = val;
}
template <class V>
__device__ __forceinline__ double lxu_load(const V& v, const double* lds_area, int idx) {
  if (v.LxG) return __hip_atomic_load(v.LxG + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return lds_area[idx];
}

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

// ---- Data that crosses workgroups of one launch (cluster hand-shake): the memory-model argument --------------------------------
// What the hardware does was MEASURED before this was written (round 6, tools/ubench/l1_probe.hip, profiles/r06_l1_probe.txt:
// a consumer workgroup that keeps a 4 KB block L1-warm, a producer on the same / another XCD rewriting it, every word checked):
//   * a CU's vector L1 is never refreshed by another CU's stores: without an acquire every re-read is stale (100 %);
//   * `buffer_inv sc0` - the workgroup-scope invalidate rounds 5 used on the one-XCD path - has NO effect on such lines outside
//     threadgroup-split mode: 100 % stale, same XCD or not.  That hand-shake passed its tests only where the 32 KB L1 had been
//     thrashed in between (large models), and failed on small plugin shapes - both stale-data failures of round 5 were this;
//   * an agent-scope acquire fence (`buffer_inv sc1`) followed by plain loads, or agent-scope (sc1) loads without any fence,
//     read fresh data - from a same-XCD producer after `s_waitcnt vmcnt(0)` alone, from a producer on ANOTHER XCD only if its
//     stores were agent-scope (sc1, write-through) stores or were followed by an agent-scope release fence (`buffer_wbl2 sc1`).
// So every hand-off below takes one of the two forms that are correct for ANY placement of the workgroups (LLVM's AMDGPU memory
// model for gfx942 / gfx950: an agent-scope atomic store is `global_store sc1`, complete at agent scope when the vector-memory
// counter has counted it; an agent-scope atomic load is `global_load sc1`; an agent-scope acquire fence is `buffer_inv sc1`):
//   (A) payload in agent-scope stores (st_shared<true>), every wave drains its own (drain_stores), a barrier collects the waves,
//       ONE thread then raises the flag (relaxed agent-scope atomic); the reader polls the flag (relaxed), and either reads the
//       payload with agent-scope loads (ld_shared) or executes ONE agent-scope acquire fence + barrier and reads it normally;
//   (B) payload in plain stores (the gains a backward pass leaves, a candidate group's trial trajectories): every wave drains, a
//       barrier, ONE thread executes an agent-scope release fence (cluster_release) and then raises the flag; the reader: poll,
//       agent-scope acquire fence, barrier, plain loads.
// Nothing depends on which XCD a workgroup runs on, on the size of a cache or on how long a step takes; the one-XCD placement of a
// cluster (launch_large.hpp) is a speed choice (the leaders spread over the eight L2s).
template <bool COHERENT>
__device__ __forceinline__ void st_shared(double* p, double v) {
  if constexpr (COHERENT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
__device__ __forceinline__ double ld_shared(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void wait_stores() { __builtin_amdgcn_s_waitcnt(0x0F70); }   // vmcnt(0): this thread's stores acknowledged
// This wave's vector-memory operations are complete (inline assembly: the compiler's wait-count pass cannot drop or move it -
// ROCm 7.2 drops the wait behind a release fence when its own scoreboard believes the counter is empty).
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// Form (A): the agent-scope stores of EVERY thread of the workgroup are complete once this returns; the flag store that follows
// may be relaxed.
__device__ __forceinline__ void cluster_publish_barrier() {
  drain_stores();
  __syncthreads();
}
// Form (B), after cluster_publish_barrier(), by the ONE thread that raises the flag next: writes back the dirty lines of this XCD's
// L2 (the plain stores of every wave of the workgroup have reached it: drained + barrier) so that a reader behind another L2 finds
// them; a reader behind the same L2 needs only the drain.
__device__ __forceinline__ void cluster_release() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  drain_stores();
}
// The reader's side of either form, by ONE thread after it has seen the flag, followed by a barrier: drops this CU's vector L1
// (and whatever of this XCD's L2 may be stale with respect to another XCD's write-through stores); later plain loads of the
// workgroup read what the writer published.
__device__ __forceinline__ void cluster_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
// XCC (= XCD) this wave runs on
__device__ __forceinline__ int xcc_id() {
#ifdef MI_NO_XCC
  return 0;
#else
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return (int)(x & 15u);
#endif
}

// Models whose step is an articulated-body algorithm cut into chains (models.hpp: PlanarQuad): cooperative
// step in the rollout, accessor-driven whole-tree evaluation in the linearization.
template <class M, class = void>
struct IsChainModel { static constexpr bool value = false; };
template <class M>
struct IsChainModel<M, decltype((void)M::kChainCooperative)> { static constexpr bool value = M::kChainCooperative; };
// Models whose step is cut per LEG of a floating-base body (models.hpp: Quad3D): one lane per leg in the rollout, the
// legs' wrenches summed over the 16-lane row; whole-step evaluation per (key-point, column) item in the linearization.
template <class M, class = void>
struct IsLegModel { static constexpr bool value = false; };
template <class M>
struct IsLegModel<M, decltype((void)M::kLegCooperative)> { static constexpr bool value = M::kLegCooperative; };
// Models that only provide the whole step (plugins, include/mi_ilqr.h: open model interface): one lane advances the
// dynamics in the rollout, whole-step evaluation per (key-point, column) item in the linearization.
template <class M, class = void>
struct IsWholeStepModel { static constexpr bool value = false; };
template <class M>
struct IsWholeStepModel<M, decltype((void)M::kWholeStep)> { static constexpr bool value = M::kWholeStep; };
// Models whose step starts with the sines / cosines of kJoints independent angles (models.hpp: Arm27): the rollout evaluates
// them on 2 kJoints lanes at once and every lane of the first 16-lane row runs the rest of the step (M::core) on the shared
// values; lane 0 publishes the result.  Bitwise M::step.
template <class M, class = void>
struct IsTrigModel { static constexpr bool value = false; };
template <class M>
struct IsTrigModel<M, decltype((void)M::kTrigCooperative)> { static constexpr bool value = M::kTrigCooperative; };
// Models that can declare a step infeasible (Drake's update throwing -> L = inf, ilqr.py:315-323).
template <class M, class = void>
struct CanFail { static constexpr bool value = false; };
template <class M>
struct CanFail<M, decltype((void)M::kCanFail)> { static constexpr bool value = M::kCanFail; };

// Models whose rollout leaves lx_t, lu_t of the accepted trial in the backward pass's cost-gradient area.
template <class M>
constexpr bool kLxFromRollout = !IsChainModel<M>::value;

__device__ __forceinline__ double block_sum(double v, double* red) {
  // deterministic fixed-order tree: 64-lane butterfly, then 4 wave partials
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// The model's parameters, dt and the difference step as scalars of their OWN.  Left inside the kernel-argument block
// they belong to a 16-register tuple that the register allocator spills and restores as a whole once scalar registers
// run out - 16 v_readlane per use inside the step loops of the quadruped kernels (9 uses per rollout step of the 3-D
// quadruped: 144 of its 1319 instructions).  Opaque copies are allocated pair by pair.  (Models with a handful of
// parameters keep the plain reads: for the 4-parameter chain the copies measured 4 % slower in the MPC kernel.)
template <class M>
struct ModelScalars {
  double p[M::n_params], dt, fd_h;
  __device__ __forceinline__ explicit ModelScalars(const KArgs& a) {
#pragma unroll
    for (int i = 0; i < M::n_params; ++i) { p[i] = a.params[i]; if constexpr (M::n_params > 4) asm volatile("" : "+s"(p[i])); }
    dt = a.dt; if constexpr (M::n_params > 4) asm volatile("" : "+s"(dt));
    fd_h = a.fd_h; if constexpr (M::n_params > 4) asm volatile("" : "+s"(fd_h));
  }
};

// lane i's value -> S[i], lane 8 + i's -> C[i], in every lane of the 16-lane row (DPP row_share)
template <int NJ, int I>
__device__ __forceinline__ void trig_gather(double v, double (&S)[NJ], double (&C)[NJ]) {
  if constexpr (I < NJ) {
    S[I] = row_share<I>(v);
    C[I] = row_share<8 + I>(v);
    trig_gather<NJ, I + 1>(v, S, C);
  }
}

// One line-search trial (ilqr.py:306-327).  Returns L on every thread; trajectory -> Xn/Un.
// Per step: (1) 16 lanes per control row form K_t(x-x_bar) partial dots — K_t, x_bar_t,
// u_bar_t, kappa_t come from HBM/L2 and are prefetched one step ahead into registers;
// (2) one lane per degree of freedom advances the dynamics while other waves add the
// stage-cost rows (their Q/R rows live in registers); (3) the new state is published.
// prog != nullptr: the rollout PUBLISHES the trial as it goes, for helper workgroups that linearize it while it is still being
// rolled out (ilqr_large_kernel: early linearization).  x_t, u_t then leave as agent-scope (write-through) stores - all from the
// fourth wave - and lane 192 stores `tag | s` into *prog once the steps 0 .. s-1 are complete at agent scope (form (A) of the
// hand-shake above; the helpers read the rows with agent-scope loads).  How it knows:
//   MI_PUB_LAG = 0 (default): at the top of step t, BEFORE this step's stores are issued, the wave drains its vector-memory counter
//     - s_waitcnt vmcnt(0): every store it has issued (steps 0 .. t-1) is complete - and publishes s = t.  No assumption about the
//     order in which operations complete, none about how long a step takes (round 5's form rested on both, and was opened for the
//     built-in models only).  What the wave waits for is at least half a step old (u_{t-1} left in the middle of step t-1).
//   MI_PUB_LAG = k > 0 (A/B builds): the counted form - s_waitcnt vmcnt(3 k) "all but the youngest 3 k operations are complete" and
//     s = t - k.  Right only while the wave has nothing but stores in flight (operations of one kind complete in the order issued;
//     a load - a register spill's reload, say - would not be ordered with them).
#ifndef MI_PUB_LAG
#define MI_PUB_LAG 0
#endif
constexpr int kPubLag = MI_PUB_LAG;
static_assert(kPubLag >= 0 && 3 * kPubLag <= 15, "the counted wait of large_rollout's publisher sits in the 4-bit low field of s_waitcnt's vmcnt");
constexpr unsigned long long kPubAbort = 0x80000000ull;      // *prog = tag | kPubAbort: the trial was rejected, stop linearizing it
// Blocks at the END of the horizon that the leader linearizes itself once the trial is accepted (early rounds): models whose
// helpers cannot keep up with the rollout (planar quadruped: an item is two passes over the tree, ~78 k cycles) - the last
// block comes out last anyway, and the leader's hands are free by then.
template <class M, class = void>
struct EarlyLeaderBlocks { static constexpr int value = 0; };
template <class M>
struct EarlyLeaderBlocks<M, decltype((void)M::kEarlyLeaderBlocks)> { static constexpr int value = M::kEarlyLeaderBlocks; };
template <class M>
constexpr bool kEarlyLin = M::m * 16 <= 192;                 // (the fourth wave holds no control-law lanes: no prefetch loads on it)
template <class M>
__device__ inline double large_rollout(const LView<M::n, M::m>& v, double* lds, const KArgs& a,
                                       const double* x0g, double eps, double& expd_out,
                                       unsigned long long* prog = nullptr, unsigned long long tag = 0ull) {
  constexpr int n = M::n, m = M::m;
  using Ly = LLay<n, m>;
  constexpr int JR = (n + 15) / 16;    // K-row elements per lane
  const int tid = stage_tid(), N = v.N;
  double* xs = lds + Ly::oXs;
  double* us = lds + Ly::oUs;
  const double* xnom = lds + Ly::oXnom;
  const bool urole = tid < m * 16;
  const int uk = tid >> 4, ul = tid & 15;
  const bool qrole = tid >= 64 && tid < 64 + n;      // cost row i = tid-64, first half of its dot product
  const bool q2role = tid >= 192 && tid < 192 + n;   // cost row i = tid-192, second half (the fourth wave)
  constexpr int nh = n / 2;
  const bool rrole = tid >= 128 && tid < 128 + m;    // control-cost row k = tid-128
  static_assert(n + m <= 64, "u_t leaves from the fourth wave's lanes behind the n state lanes");
  const bool uorole = tid >= 192 + n && tid < 192 + n + m;   // u_t -> HBM (the fourth wave: with x_t, every trajectory store of the step)
  // cost rows -> registers (one-off)
  double qrow[n], rrow[m];
  if (qrole || q2role) {
    const int i = qrole ? tid - 64 : tid - 192;
#pragma unroll
    for (int j = 0; j < n; ++j) qrow[j] = lds[Ly::oQ + i * n + j];
  }
  if (rrole) {
#pragma unroll
    for (int j = 0; j < m; ++j) rrow[j] = lds[Ly::oR + (tid - 128) * m + j];
  }
  // the state ping-pongs between two LDS buffers so the new state is written while the old one
  // is still being read: two workgroup barriers per step instead of three
  double* xs2 = lds + Ly::oXb;
  static_assert(Ly::oQc - Ly::oXb >= n, "second state buffer");
  // x_t - x_nom is formed ONCE per step - by the fourth wave, idle while the other three form the control law -
  // and published in the backward pass's idle T1 area: the 36 cost-row lanes used to form it themselves, 2 x 36
  // LDS reads + 36 subtractions per step each, and their wave was the step's critical path (1981 of the synthetic
  // chain's 2685 cycles per step, against 1065 for the dynamics wave); same operands, same bits.
  double* dxc = lds + Ly::oT1;
  static_assert(n <= 64 && Ly::NK * Ly::TS >= 64, "an n-vector inside T1");
  const bool drole = tid >= 192 && tid < 192 + n;
  const double xnr = drole ? xnom[tid - 192] : 0.0;
  // The cost rows have (Q (x_t - x_nom))_i and (R u_t)_k in hand: twice that IS lx_t / lu_t of the trial (symmetric
  // Q, the only kind this kernel accepts), and the last trial rolled out is the accepted one - so the rows go
  // straight to the backward pass's cost-gradient area and its prologue need not stage x_bar, u_bar and form the
  // (N-1) x n x n product again (14 k cycles per pass).  The two halves of a row meet through a parity-indexed
  // scratch pair one step later.  (Not for chain models: their linearization uses that area as a cache.)
  constexpr bool kLx = kLxFromRollout<M>;
  double* Lxu = lds + Ly::doubles;
  double* r2buf = lds + Ly::oT1 + 64;                                  // [2][64]
  static_assert(Ly::NK * Ly::TS >= 192, "dx + two half-row scratch vectors inside T1");
  double r1_prev = 0.0;
  const ModelScalars<M> ms(a);
  const double* prm = ms.p;
  const double dt_ = ms.dt;
  // Models whose whole state is advanced by ONE lane (whole-step plugins) or redundantly by every lane of a row (trig models):
  // that lane carries x_t in registers from step to step (it does not read back what it has just published), and the
  // trajectory goes to HBM from the lanes that form x_t - x_nom - one coalesced store instead of n scalar ones on the
  // lane every other thread is waiting for.
  // (every model: x_t goes to HBM from the x - x_nom lanes, the dynamics lanes only publish to LDS)
  constexpr bool kCarry = IsTrigModel<M>::value || IsWholeStepModel<M>::value;
  if (tid < n) xs[tid] = x0g[tid];
  double xr[kCarry ? n : 1];
  if constexpr (kCarry) {
#pragma unroll
    for (int i = 0; i < n; ++i) xr[i] = x0g[i];
  }
  double acc = 0.0;                    // per-thread cost partial over all time steps
  bool bad = false;                    // this thread saw an infeasible step (models that can fail)
  // Operands of the control law (K_t row slice, x_bar_t slice, u_bar_t, kappa_t) come from
  // L2/MALL: they are requested TWO steps ahead into two alternating register sets (a step is
  // ~1.5 k cycles of work; an Infinity-Cache hit costs about that much on its own).
  struct Pf { double kr[JR], xbr[JR], ubk, kpk; };
  Pf pfA, pfB;
  auto prefetch = [&](Pf& f, int t) __attribute__((always_inline)) {
    if (urole) {
      const double* Kr = v.K + ((size_t)t * m + uk) * n;
      const double* xbt = v.X + (size_t)t * n;
#pragma unroll
      for (int q = 0; q < JR; ++q) {
        const int j = ul + 16 * q;
        f.kr[q] = (j < n) ? Kr[j] : 0.0;
        f.xbr[q] = (j < n) ? xbt[j] : 0.0;
      }
      f.ubk = 0.0; f.kpk = 0.0;
      if (ul == 0) { f.ubk = v.U[(size_t)t * m + uk]; f.kpk = v.kap[(size_t)t * m + uk]; }
    }
  };
  double* xc = xs;                     // current state
  double* xn_ = xs2;                   // next state
  auto one_step = [&](Pf& f, int t) __attribute__((always_inline)) {
    // u_t = u_bar_t - eps*kappa_t - K_t (x_t - x_bar_t)   (ilqr.py:313)
    if (urole) {
      double p = 0.0;
#pragma unroll
      for (int q = 0; q < JR; ++q) {
        const int j = ul + 16 * q;
        if (j < n) p += f.kr[q] * (xc[j] - f.xbr[q]);
      }
      p = row16_sum(p);
      if (ul == 0) us[uk] = (f.ubk - eps * f.kpk) - p;
    }
    if (drole) {
      const double xv_ = xc[tid - 192];
      dxc[tid - 192] = xv_ - xnr;
      if (prog) {
        if (tid == 192 && t >= 1) {
          if constexpr (kPubLag == 0) {
            drain_stores();                                  // steps 0 .. t-1: every store this wave has issued is complete
            __hip_atomic_store(prog, tag | (unsigned long long)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } else {
            // every step from the second on publishes (a count of zero claims nothing), so the wave has issued exactly
            // (publish, x, u) x kPubLag operations since u_{t-1-kPubLag}: all but that many complete = steps 0 .. t-1-kPubLag out
            if (t > kPubLag) __builtin_amdgcn_s_waitcnt(0x0F70 | (3 * kPubLag));
            asm volatile("" ::: "memory");
            __hip_atomic_store(prog, tag | (unsigned long long)(t > kPubLag ? t - kPubLag : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
      // a published trial leaves in agent-scope stores (form (A) of the hand-shake: the helpers read it with agent-scope loads)
      if (prog) st_shared<true>(v.Xn + (size_t)t * n + (tid - 192), xv_);
      else v.Xn[(size_t)t * n + (tid - 192)] = xv_;
    }
    if (t + 2 < N - 1) prefetch(f, t + 2);       // this set is free again
    lds_barrier();
    // dynamics (ilqr.py:316); cost rows on the other waves (:325)
    bool dyn_done = false;
    if constexpr (IsChainModel<M>::value) {
      // articulated body: ONE LANE PER CHAIN of the tree (lanes 0..kChains-1 of wave 0; the other lanes of the
      // first 16-lane row shadow the last chain and contribute zeros).  Leaves-to-root pass per chain, the
      // chains' articulated inertias / bias forces summed over the row with DPP rotations, the trunk's 3x3
      // system solved redundantly in every lane, root-to-leaves pass per chain.  No LDS traffic, no barrier.
      if (tid < 16) {
        const int ch = tid < M::kChains ? tid : M::kChains - 1;
        typename M::template Trunk<double> tr;
        M::template trunk_state<double>(xc, tr);
        typename M::template Agg<double> ag;
        typename M::template Saved<double> sv;
        M::template chain_up<double>(ch, tr, xc, us, prm, ag, sv);
        const double keep = tid < M::kChains ? 1.0 : 0.0;
        typename M::template Agg<double> tot;
        M::template trunk_agg<double>(prm, tot);
        tot.J += row16_sum(keep * ag.J); tot.hx += row16_sum(keep * ag.hx); tot.hz += row16_sum(keep * ag.hz);
        tot.mxx += row16_sum(keep * ag.mxx); tot.mxz += row16_sum(keep * ag.mxz); tot.mzz += row16_sum(keep * ag.mzz);
        tot.bn += row16_sum(keep * ag.bn); tot.bx += row16_sum(keep * ag.bx); tot.bz += row16_sum(keep * ag.bz);
        double ax, az, alpha;
        M::template base_solve<double>(tot, ax, az, alpha);
        double q3[3];
        M::template chain_down<double>(sv, alpha, ax, az, q3);
        if (tid < M::kChains) {
#pragma unroll
          for (int b_ = 0; b_ < 3; ++b_) {
            const int i = 3 + 3 * tid + b_;
            const double vn_ = xc[M::nq + i] + dt_ * q3[b_];
            const double qn_ = xc[i] + dt_ * vn_;
            xn_[i] = qn_; xn_[M::nq + i] = vn_;
            bad = bad || M::infeasible_velocity(vn_, prm);
          }
        }
        if (tid < 3) {                                     // the trunk's own coordinates: x, z, pitch
          const double acc_ = tid == 0 ? ax : (tid == 1 ? az : alpha);
          const double vn_ = xc[M::nq + tid] + dt_ * acc_;
          const double qn_ = xc[tid] + dt_ * vn_;
          xn_[tid] = qn_; xn_[M::nq + tid] = vn_;
          bad = bad || M::infeasible_velocity(vn_, prm);
        }
        dyn_done = true;
      }
    } else if constexpr (IsLegModel<M>::value) {
      // floating base + legs: ONE LANE PER LEG (lanes 0..kLegs-1 of wave 0; the rest of the first 16-lane row shadows
      // the last leg and contributes zeros).  Foot kinematics, contact force and joint accelerations per leg, the legs'
      // wrenches summed over the row with DPP rotations - (0 + 2) + (1 + 3), the order M::step uses - and the trunk
      // advanced redundantly by every lane.  No LDS traffic, no barrier inside the step.
      if (tid < 16) {
        const int k = tid < M::kLegs ? tid : M::kLegs - 1;
        double Rm[3][3];
        M::template rotation<double>(xc, Rm);
        typename M::template LegOut<double> lo;
        M::template leg<double>(k, Rm, xc, us, prm, lo);
        const double keep = tid < M::kLegs ? 1.0 : 0.0;
        double Fw[3], Tq[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) { Fw[i] = row16_sum(keep * lo.fw[i]); Tq[i] = row16_sum(keep * lo.tq[i]); }
        double xt[n];
        M::template trunk<double>(xc, Fw, Tq, xt, prm, dt_);
        if (tid < M::kLegs) {
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const int j = 3 * k + i;
            const double jdn = xc[25 + j] + dt_ * lo.ja[i];
            const double jn = xc[7 + j] + dt_ * jdn;
            xn_[25 + j] = jdn; xn_[7 + j] = jn;
            bad = bad || M::infeasible_velocity(jdn, prm);
          }
        }
        if (tid == 0) {
#pragma unroll
          for (int i = 0; i < 7; ++i) xn_[i] = xt[i];
#pragma unroll
          for (int i = 19; i < 25; ++i) { xn_[i] = xt[i]; bad = bad || M::infeasible_velocity(xt[i], prm); }
        }
        dyn_done = true;
      }
    } else if constexpr (IsTrigModel<M>::value) {
      if (tid < 16) {
        static_assert(M::kJoints <= 8, "sines on lanes 0.., cosines on lanes 8.. of a 16-lane row");
        const int j = (tid & 7) < M::kJoints ? (tid & 7) : M::kJoints - 1;
        double xj_ = xr[0];                                // (joint angle j of the carried state: a select chain, no LDS round trip)
#pragma unroll
        for (int q_ = 1; q_ < M::kJoints; ++q_) xj_ = (j == q_) ? xr[q_] : xj_;
        const double sc_ = fast_sin_or_cos(xj_, tid >= 8);
        double S_[M::kJoints], C_[M::kJoints];
        trig_gather<M::kJoints, 0>(sc_, S_, C_);
        double xt[n];
        M::template core<double, true>(S_, C_, xr, us, xt, prm, dt_);
#pragma unroll
        for (int i = 0; i < n; ++i) xr[i] = xt[i];
        if (tid == 0) {
#pragma unroll
          for (int i = 0; i < n; ++i) xn_[i] = xt[i];
        }
        dyn_done = true;
      }
    } else if constexpr (IsWholeStepModel<M>::value) {
      if (tid == 0) {                                      // (plugin models without cooperative hooks)
        double xt[n];
        M::template step<double>(xr, us, xt, prm, dt_);
#pragma unroll
        for (int i = 0; i < n; ++i) { xn_[i] = xt[i]; xr[i] = xt[i]; }
        dyn_done = true;
      }
    } else {
      if (tid < M::nq) {                                   // one lane per degree of freedom
        double qn_ = 0.0, vn_ = 0.0;
        M::template dof<double>(tid, xc, us, qn_, vn_, prm, dt_);
        xn_[tid] = qn_; xn_[M::nq + tid] = vn_;
        dyn_done = true;
      }
    }
    if (dyn_done) {
    } else if (qrole) {                                  // (x - x_nom)^T Q (x - x_nom): row i, columns 0..n/2-1
      const int i = tid - 64;
      double r = 0.0;
#pragma unroll
      for (int j = 0; j < nh; ++j) r += qrow[j] * dxc[j];
      acc += dxc[i] * r;
      if constexpr (kLx) {
        if (t > 0) lxu_store(v, Lxu, (t - 1) * (n + m) + i, 2.0 * (r1_prev + r2buf[((t - 1) & 1) * 64 + i]));
        r1_prev = r;
      }
    } else if (q2role) {                                 // ... and columns n/2..n-1, on the wave that has nothing else to do here
      const int i = tid - 192;
      double r = 0.0;
#pragma unroll
      for (int j = nh; j < n; ++j) r += qrow[j] * dxc[j];
      acc += dxc[i] * r;
      if constexpr (kLx) r2buf[(t & 1) * 64 + i] = r;
    } else if (rrole) {
      const int k = tid - 128;
      double r = 0.0;
#pragma unroll
      for (int j = 0; j < m; ++j) r += rrow[j] * us[j];
      acc += us[k] * r;
      if constexpr (kLx) lxu_store(v, Lxu, t * (n + m) + n + k, 2.0 * r);
#ifdef MI_UN_RROLE
      v.Un[(size_t)t * m + k] = us[k];
#endif
    } else if (uorole) {
      const int k = tid - 192 - n;
      if (prog) st_shared<true>(v.Un + (size_t)t * m + k, us[k]);
      else v.Un[(size_t)t * m + k] = us[k];
    }
    lds_barrier();
    double* tmp_ = xc; xc = xn_; xn_ = tmp_;
  };
  prefetch(pfA, 0);
  if (1 < N - 1) prefetch(pfB, 1);
  __syncthreads();
  for (int t = 0; t < N - 1; t += 2) {
    one_step(pfA, t);
    if (t + 1 < N - 1) one_step(pfB, t + 1);
  }
  xs = xc;                             // final state x_{N-1}
  if (tid < n) v.Xn[(size_t)(N - 1) * n + tid] = xs[tid];
  if (prog && tid == 192) {            // every step of the trial is out
    drain_stores();
    __hip_atomic_store(prog, tag | (unsigned long long)(N - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if constexpr (kLx) {
    if (qrole && N >= 2) lxu_store(v, Lxu, (N - 2) * (n + m) + (tid - 64), 2.0 * (r1_prev + r2buf[((N - 2) & 1) * 64 + (tid - 64)]));
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
  double L = block_sum(acc, red);
  const double dvs = block_sum(dvp, red);
  expd_out = -eps * (1.0 - eps / 2.0) * dvs;             // ilqr.py:326
  if constexpr (CanFail<M>::value) {
    // a step was declared infeasible: the trial's cost is +inf (ilqr.py:317-323; the reference stops simulating
    // there - what this rollout computed past that step is never looked at: L = inf is never accepted, :330)
    // (workgroup-wide OR through the reduction scratch: HIP's __syncthreads_or carries a static LDS variable,
    // which would take the kernel's dynamic LDS below the 160 KB the layout is sized for)
    const double flagged = block_sum(bad ? 1.0 : 0.0, red);
    if (flagged > 0.0) L = __builtin_inf();
  }
  return L;
}

// Sequential line search (ilqr.py:300-337); on accept Xn/Un hold the trajectory.
template <class M>
// prog: the FIRST trial is published step by step (large_rollout); a rejected first trial is called off at once.
__device__ inline bool large_linesearch(const LView<M::n, M::m>& v, double* lds, const KArgs& a, const double* x0g,
                                        double L_last, double& L_out, double& eps_out, int& trials,
                                        unsigned long long* prog = nullptr, unsigned long long tag = 0ull) {
  double eps = 1.0;
  trials = 0;
  while (eps >= 1e-8) {
    trials += 1;
    double ex;
    const double L = large_rollout<M>(v, lds, a, x0g, eps, ex, trials == 1 ? prog : nullptr, tag);
    if ((L_last - L) > a.gamma * ex) { L_out = L; eps_out = eps; return true; }
    if (prog && trials == 1 && threadIdx.x == 192) __hip_atomic_store(prog, tag | kPubAbort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    eps *= a.beta;
    __syncthreads();
  }
  return false;
}

// ---- mid-size kernels: FOUR line-search candidates per pass ------------------------------------------------------
// A rollout is a chain of N-1 dependent steps on a handful of lanes - the other lanes of the workgroup wait with it.
// A problem that backtracks (the arm + ball: 2.4 trials per iteration) pays that chain once per trial.  mid_rollout4
// walks the candidates eps, eps beta, eps beta^2, eps beta^3 through the SAME step loop: the control-law lanes form
// four dot products instead of one (independent DPP reductions), the dynamics run on four lanes (whole-step models)
// or four 16-lane rows (trig models) of wave 0 in one instruction stream, the cost rows accumulate four sums.  Per
// candidate the arithmetic is large_rollout's, operation for operation - the same bits - and the search takes the
// FIRST candidate in order that passes the test of ilqr.py:330, which is what the sequential search returns.
constexpr int kSpec = 4;
template <class M>
constexpr bool kSpecRollout = LLay<M::n, M::m>::kMid && (IsTrigModel<M>::value || IsWholeStepModel<M>::value) && !CanFail<M>::value;
// KArgs::spec_policy - 0: never; 1 (default): once the problem has backtracked in this launch (a problem that never
// backtracks never pays for candidates it does not need; a rule on the problem's own history, so results do not depend
// on the batch); 2: whenever there is a cost to beat.  MI_ILQR_SPEC=0|1|2 overrides (A/B runs, tests).

template <class M>
__device__ inline void mid_rollout4(const LView<M::n, M::m>& v, double* xsp, double* usp, size_t sx, size_t su, double* lds,
                                    const KArgs& a, const double* x0g, const double (&eps4)[kSpec], double (&L4)[kSpec],
                                    double& dvs_out) {
  constexpr int n = M::n, m = M::m;
  using Ly = LLay<n, m>;
  constexpr int JR = (n + 15) / 16;
  constexpr int XS = 34;               // candidate stride of the state buffers: the candidates' copies of an entry on different banks
  const int tid = stage_tid(), N = v.N;
  // state (two buffers), controls and x - x_nom of the four candidates: the backward pass's H area is idle
  double* xc = lds + Ly::oH;           // [4][XS] current
  double* xn_ = xc + kSpec * XS;       // [4][XS] next
  double* us4 = xn_ + kSpec * XS;      // [4][16]
  double* dx4 = us4 + kSpec * 16;      // [4][32]
  static_assert(n <= 32 && m <= 16 && 2 * kSpec * XS + kSpec * 48 <= Ly::NMP * Ly::TS, "candidate buffers inside H");
  const double* xnom = lds + Ly::oXnom;
  const bool urole = tid < m * 16;
  const int uk = tid >> 4, ul = tid & 15;
  const bool qrole = tid >= 64 && tid < 64 + n;
  const bool q2role = tid >= 192 && tid < 192 + n;
  constexpr int nh = n / 2;
  const bool rrole = tid >= 128 && tid < 128 + m;
  double qrow[n], rrow[m];
  if (qrole || q2role) {
    const int i = qrole ? tid - 64 : tid - 192;
#pragma unroll
    for (int j = 0; j < n; ++j) qrow[j] = lds[Ly::oQ + i * n + j];
  }
  if (rrole) {
#pragma unroll
    for (int j = 0; j < m; ++j) rrow[j] = lds[Ly::oR + (tid - 128) * m + j];
  }
  const bool drole = tid >= 192 && tid < 192 + n;
  const double xnr = drole ? xnom[tid - 192] : 0.0;
  const ModelScalars<M> ms(a);
  const double* prm = ms.p;
  const double dt_ = ms.dt;
  auto Xo = [&](int c) __attribute__((always_inline)) { return c == 0 ? v.Xn : xsp + (size_t)(c - 1) * sx; };
  auto Uo = [&](int c) __attribute__((always_inline)) { return c == 0 ? v.Un : usp + (size_t)(c - 1) * su; };
  if (tid < n) {
    const double x0v = x0g[tid];
#pragma unroll
    for (int c = 0; c < kSpec; ++c) xc[c * XS + tid] = x0v;
  }
  double xr[n];                                            // the dynamics lanes carry their candidate's state (large_rollout: kCarry)
#pragma unroll
  for (int i = 0; i < n; ++i) xr[i] = x0g[i];
  double acc[kSpec] = {0.0, 0.0, 0.0, 0.0};
  struct Pf { double kr[JR], xbr[JR], ubk, kpk; };
  Pf pfA, pfB;
  auto prefetch = [&](Pf& f, int t) __attribute__((always_inline)) {
    if (urole) {
      const double* Kr = v.K + ((size_t)t * m + uk) * n;
      const double* xbt = v.X + (size_t)t * n;
#pragma unroll
      for (int q = 0; q < JR; ++q) {
        const int j = ul + 16 * q;
        f.kr[q] = (j < n) ? Kr[j] : 0.0;
        f.xbr[q] = (j < n) ? xbt[j] : 0.0;
      }
      f.ubk = 0.0; f.kpk = 0.0;
      if (ul == 0) { f.ubk = v.U[(size_t)t * m + uk]; f.kpk = v.kap[(size_t)t * m + uk]; }
    }
  };
  auto one_step = [&](Pf& f, int t) __attribute__((always_inline)) {
    if (urole) {                                           // u_t = u_bar_t - eps kappa_t - K_t (x_t - x_bar_t)   (ilqr.py:313)
      double p[kSpec];
#pragma unroll
      for (int c = 0; c < kSpec; ++c) {
        p[c] = 0.0;
#pragma unroll
        for (int q = 0; q < JR; ++q) {
          const int j = ul + 16 * q;
          if (j < n) p[c] += f.kr[q] * (xc[c * XS + j] - f.xbr[q]);
        }
      }
#pragma unroll
      for (int c = 0; c < kSpec; ++c) p[c] = row16_sum(p[c]);
      if (ul == 0) {
#pragma unroll
        for (int c = 0; c < kSpec; ++c) us4[c * 16 + uk] = (f.ubk - eps4[c] * f.kpk) - p[c];
      }
    }
    if (drole) {
#pragma unroll
      for (int c = 0; c < kSpec; ++c) {
        const double xv_ = xc[c * XS + tid - 192];
        dx4[c * 32 + tid - 192] = xv_ - xnr;
        Xo(c)[(size_t)t * n + (tid - 192)] = xv_;
      }
    }
    if (t + 2 < N - 1) prefetch(f, t + 2);
    lds_barrier();
    if constexpr (IsTrigModel<M>::value) {
      if (tid < 16 * kSpec) {                              // candidate c on the c-th 16-lane row of wave 0
        const int c = tid >> 4, l = tid & 15;
        const double* usc = us4 + c * 16;
        const int j = (l & 7) < M::kJoints ? (l & 7) : M::kJoints - 1;
        double xj_ = xr[0];
#pragma unroll
        for (int q_ = 1; q_ < M::kJoints; ++q_) xj_ = (j == q_) ? xr[q_] : xj_;
        const double sc_ = fast_sin_or_cos(xj_, l >= 8);
        double S_[M::kJoints], C_[M::kJoints];
        trig_gather<M::kJoints, 0>(sc_, S_, C_);
        double xt[n];
        M::template core<double, true>(S_, C_, xr, usc, xt, prm, dt_);
#pragma unroll
        for (int i = 0; i < n; ++i) xr[i] = xt[i];
        if (l == 0) {
#pragma unroll
          for (int i = 0; i < n; ++i) xn_[c * XS + i] = xt[i];
        }
      }
    } else {
      if (tid < kSpec) {                                   // whole-step models: candidate c on lane c
        const int c = tid;
        double xt[n];
        M::template step<double>(xr, us4 + c * 16, xt, prm, dt_);
#pragma unroll
        for (int i = 0; i < n; ++i) { xn_[c * XS + i] = xt[i]; xr[i] = xt[i]; }
      }
    }
    if (qrole) {                                           // (x - x_nom)^T Q (x - x_nom): row i, columns 0 .. n/2 - 1
      const int i = tid - 64;
#pragma unroll
      for (int c = 0; c < kSpec; ++c) {
        double r = 0.0;
#pragma unroll
        for (int j = 0; j < nh; ++j) r += qrow[j] * dx4[c * 32 + j];
        acc[c] += dx4[c * 32 + i] * r;
      }
    } else if (q2role) {                                   // ... columns n/2 .. n - 1
      const int i = tid - 192;
#pragma unroll
      for (int c = 0; c < kSpec; ++c) {
        double r = 0.0;
#pragma unroll
        for (int j = nh; j < n; ++j) r += qrow[j] * dx4[c * 32 + j];
        acc[c] += dx4[c * 32 + i] * r;
      }
    } else if (rrole) {
      const int k = tid - 128;
#pragma unroll
      for (int c = 0; c < kSpec; ++c) {
        double r = 0.0;
#pragma unroll
        for (int j = 0; j < m; ++j) r += rrow[j] * us4[c * 16 + j];
        acc[c] += us4[c * 16 + k] * r;
        Uo(c)[(size_t)t * m + k] = us4[c * 16 + k];
      }
    }
    lds_barrier();
    double* tmp_ = xc; xc = xn_; xn_ = tmp_;
  };
  prefetch(pfA, 0);
  if (1 < N - 1) prefetch(pfB, 1);
  __syncthreads();
  for (int t = 0; t < N - 1; t += 2) {
    one_step(pfA, t);
    if (t + 1 < N - 1) one_step(pfB, t + 1);
  }
  if (tid < n) {
#pragma unroll
    for (int c = 0; c < kSpec; ++c) Xo(c)[(size_t)(N - 1) * n + tid] = xc[c * XS + tid];
  }
  if (qrole) {                                             // terminal cost (ilqr.py:327)
    const int i = tid - 64;
    const double* Qf = lds + Ly::oQf;
#pragma unroll
    for (int c = 0; c < kSpec; ++c) {
      double r = 0.0;
      for (int j = 0; j < n; ++j) r += Qf[i * n + j] * (xc[c * XS + j] - xnom[j]);
      acc[c] += (xc[c * XS + i] - xnom[i]) * r;
    }
  }
  double dvp = 0.0;
  for (int t = tid; t < N - 1; t += kLargeThreads) dvp += v.dV[t];
  double* red = lds + Ly::oRed;
#pragma unroll
  for (int c = 0; c < kSpec; ++c) L4[c] = block_sum(acc[c], red);
  dvs_out = block_sum(dvp, red);
}

// The line search of ilqr.py:300-337, four candidates per pass; `win` = which candidate's buffers hold the accepted trial.
// groups > 1 (ilqr_large_kernel: candidate groups): while this workgroup rolls out the candidates 0 .. 3 of the FIRST pass, the
// helper workgroups g = 1 .. groups - 1 of its cluster roll out the candidates 4 g .. 4 g + 3 (same arithmetic, their own trial
// buffers); collect() waits for them, their costs are then at Lh[4 (g - 1) + c] and the scan simply goes on in order - the first
// candidate that passes is the one the sequential search returns.  Later passes (the step sizes below beta^(4 groups)) are this
// workgroup's alone.
template <class M, class Collect>
__device__ inline bool mid_linesearch4(const LView<M::n, M::m>& v, double* xsp, double* usp, size_t sx, size_t su, double* lds,
                                       const KArgs& a, const double* x0g, double L_last, double& L_out, double& eps_out,
                                       int& trials, int& win, int groups, const unsigned long long* Lh, Collect&& collect) {
  double eps = 1.0;
  trials = 0;
  win = 0;
  bool first = true;
  while (eps >= 1e-8) {
    double e4[kSpec], L4[kSpec], dvs;
    e4[0] = eps;
#pragma unroll
    for (int c = 1; c < kSpec; ++c) e4[c] = e4[c - 1] * a.beta;               // (the sequence eps *= beta produces, :335)
    mid_rollout4<M>(v, xsp, usp, sx, su, lds, a, x0g, e4, L4, dvs);
#pragma unroll
    for (int c = 0; c < kSpec; ++c) {
      if (!(e4[c] >= 1e-8)) return false;                                     // the search has run out of step sizes (:300)
      trials += 1;
      const double ex = -e4[c] * (1.0 - e4[c] / 2.0) * dvs;                   // :326
      if ((L_last - L4[c]) > a.gamma * ex) { L_out = L4[c]; eps_out = e4[c]; win = c; return true; }
    }
    double e = e4[kSpec - 1];
    if (first && groups > 1) {
      collect();
      for (int q = 0; q < kSpec * (groups - 1); ++q) {
        e *= a.beta;
        if (!(e >= 1e-8)) return false;
        trials += 1;
        const double Lq = __longlong_as_double((long long)__hip_atomic_load(Lh + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        const double ex = -e * (1.0 - e / 2.0) * dvs;
        if ((L_last - Lq) > a.gamma * ex) { L_out = Lq; eps_out = e; win = kSpec + q; return true; }
      }
    }
    first = false;
    eps = e * a.beta;
    __syncthreads();
  }
  return false;
}

// Dynamics partials at the listed time steps of the nominal trajectory (X,U).
// Accessors that present [x | u] with one entry perturbed (finite differences) or seeded
// (forward-mode dual), reading the nominal values where they lie (L2).
struct PertAcc {
  const double* p; int col; double dh;
  __device__ __forceinline__ double operator[](int i) const { const double v_ = p[i]; return i == col ? v_ + dh : v_; }
};
struct SeedAcc {
  const double* p; int col;
  __device__ __forceinline__ Dual1 operator[](int i) const { return Dual1(p[i], i == col ? 1.0 : 0.0); }
};
template <class M, class = void>
struct HasSparsity { static constexpr bool value = false; };
template <class M>
struct HasSparsity<M, decltype((void)M::kMaxAffected)> { static constexpr bool value = true; };

// Sparse variant: only the dofs that read input column `col` are evaluated (M::affected); the
// rest of the column is written as the exact zeros the dense evaluation produces.
template <class M, int JAC, bool COH = false>
__device__ __forceinline__ void large_jac_at_sparse(const LView<M::n, M::m>& v, const KArgs& a, const int* list, int count,
                                                    const double* Xsrc, const double* Usrc, int first = 0, int stride = 1) {
  constexpr int n = M::n, m = M::m, nc = n + m, nq = M::nq;
  const ModelScalars<M> ms(a);
  const double h = ms.fd_h, inv2h = 1.0 / (2.0 * h);
  for (int it = first * kLargeThreads + stage_tid(); it < count * nc; it += stride * kLargeThreads) {
    const int ki = it / nc, col = it - ki * nc;
    const int t = list[ki];
    const double* xg = Xsrc + (size_t)t * n;
    const double* ug = Usrc + (size_t)t * m;
    double* o;
    int stride;
    if (col < n) { o = v.Fx + (size_t)t * n * n + col; stride = n; }
    else { o = v.Fu + (size_t)t * n * m + (col - n); stride = m; }
    // the whole column is (re)written: zeros first, overlapped with the evaluations below (a
    // separate coalesced zero-fill pass was measured slower: the CU's 64 B/clk store path is the
    // floor for the 540 KB of Jacobians either way, and here it hides under the arithmetic)
#pragma unroll
    for (int i = 0; i < n; ++i) st_shared<COH>(o + i * stride, 0.0);
    int dofs[M::kMaxAffected];
    const int na = M::affected(col, dofs);
#pragma unroll
    for (int a_ = 0; a_ < M::kMaxAffected; ++a_) {
      if (a_ < na) {
        const int i = dofs[a_];
        double dq, dv;
        if (JAC == MI_JAC_FD_CENTRAL) {
          double qp, vp, qm, vm;
          M::template dof<double>(i, PertAcc{xg, col, h}, PertAcc{ug, col - n, h}, qp, vp, ms.p, ms.dt);
          M::template dof<double>(i, PertAcc{xg, col, -h}, PertAcc{ug, col - n, -h}, qm, vm, ms.p, ms.dt);
          dq = (qp - qm) * inv2h;
          dv = (vp - vm) * inv2h;
        } else {
          Dual1 qd, vd;
          M::template dof<Dual1>(i, SeedAcc{xg, col}, SeedAcc{ug, col - n}, qd, vd, ms.p, ms.dt);
          dq = qd.d;
          dv = vd.d;
        }
        st_shared<COH>(o + i * stride, dq);       // same thread, same address as the zero above: program order holds
        st_shared<COH>(o + (nq + i) * stride, dv);
      }
    }
  }
}

// Chain models (articulated-body algorithm cut into chains, models.hpp): one (key-point, column) item per thread.
// A perturbed input belongs either to the trunk (its height, pitch and velocities: every chain's pass changes) or
// to ONE chain (a joint angle, joint velocity or joint torque: only that chain's leaves-to-root pass changes -
// the other chains enter the trunk's 3x3 system with their UNPERTURBED articulated inertia and bias, which a
// pre-pass computes once per key-point into `cache` (kChains x 9 doubles per key-point), and their joint
// accelerations follow the perturbed base acceleration through their unperturbed root-to-leaves data).
// 6 chain passes instead of 20 for 42 of the 48 columns; bitwise the same Jacobians as evaluating the whole tree
// for every perturbation (a chain's pass reads nothing but the trunk state and its own joints).
// Results go to memory chain by chain: no per-thread arrays, no scratch.
template <class M, int JAC, bool COH = false>
__device__ __forceinline__ void large_jac_at_tree(const LView<M::n, M::m>& v, const KArgs& a, const int* list, int count,
                                                  const double* Xsrc, const double* Usrc, int xstride, int ustride, double* cache,
                                                  int first = 0, int stride = 1) {
  constexpr int n = M::n, m = M::m, nc = n + m, nq = M::nq, NCH = M::kChains;
  const ModelScalars<M> ms(a);
  const double h = ms.fd_h, inv2h = 1.0 / (2.0 * h), dt = ms.dt;
  using AggD = typename M::template Agg<double>;
  if (JAC == MI_JAC_FD_CENTRAL) {
    for (int it = stage_tid(); it < count * NCH; it += kLargeThreads) {
      const int ki = it / NCH, c = it - ki * NCH;
      const int t = list[ki];
      const double* xg = Xsrc + (size_t)t * xstride;
      const double* ug = Usrc + (size_t)t * ustride;
      typename M::template Trunk<double> tr;
      M::template trunk_state<double>(xg, tr);
      AggD ag;
      typename M::template Saved<double> sv;
      M::template chain_up<double>(c, tr, xg, ug, ms.p, ag, sv);
      double* cc = cache + (size_t)it * 9;
      cc[0] = ag.J; cc[1] = ag.hx; cc[2] = ag.hz; cc[3] = ag.mxx; cc[4] = ag.mxz; cc[5] = ag.mzz; cc[6] = ag.bn; cc[7] = ag.bx; cc[8] = ag.bz;
    }
    __syncthreads();
  }
  // items are dealt key-point fastest, columns in the order "chain by chain, then the trunk's": the lanes of a
  // wavefront then share the owner of their column (two owners at most), so the per-chain branches below are
  // wave-uniform - dealt column-fastest they diverge and every wave runs both sides of every branch
  for (int it = first * kLargeThreads + stage_tid(); it < count * nc; it += stride * kLargeThreads) {
    const int rank = it / count, ki = it - rank * count;
    const int col = M::input_by_owner(rank);
    const int t = list[ki];
    const double* xg = Xsrc + (size_t)t * xstride;
    const double* ug = Usrc + (size_t)t * ustride;
    double* o;
    int stride;
    if (col < n) { o = v.Fx + (size_t)t * n * n + col; stride = n; }
    else { o = v.Fu + (size_t)t * n * m + (col - n); stride = m; }
    // the chain the perturbed input belongs to; -1: the trunk (every chain is affected)
    const int owner = M::chain_of_input(col);
    if (JAC == MI_JAC_FD_CENTRAL) {
      const PertAcc xp{xg, col, h}, up{ug, col - n, h}, xm{xg, col, -h}, um{ug, col - n, -h};
      typename M::template Trunk<double> trp, trm;
      M::template trunk_state<double>(xp, trp);
      M::template trunk_state<double>(xm, trm);
      AggD totp, totm;
      M::template trunk_agg<double>(ms.p, totp);
      M::template trunk_agg<double>(ms.p, totm);
      for (int c = 0; c < NCH; ++c) {
        AggD agp, agm;
        if (owner < 0 || owner == c) {
          typename M::template Saved<double> sv;
          M::template chain_up<double>(c, trp, xp, up, ms.p, agp, sv);
          M::template chain_up<double>(c, trm, xm, um, ms.p, agm, sv);
        } else {
          const double* cc = cache + ((size_t)ki * NCH + c) * 9;
          agp.J = cc[0]; agp.hx = cc[1]; agp.hz = cc[2]; agp.mxx = cc[3]; agp.mxz = cc[4]; agp.mzz = cc[5];
          agp.bn = cc[6]; agp.bx = cc[7]; agp.bz = cc[8];
          agm = agp;
        }
        M::agg_add(totp, agp);
        M::agg_add(totm, agm);
      }
      double axp, azp, alp, axm, azm, alm;
      M::template base_solve<double>(totp, axp, azp, alp);
      M::template base_solve<double>(totm, axm, azm, alm);
      auto emit = [&](int i, double accp, double accm) __attribute__((always_inline)) {   // (f(x+h e) - f(x-h e)) / 2h on
        const double vp = xp[nq + i] + dt * accp, vm = xm[nq + i] + dt * accm;          //  v+ = v + dt a, q+ = q + dt v+
        const double qp = xp[i] + dt * vp, qm = xm[i] + dt * vm;
        st_shared<COH>(o + i * stride, (qp - qm) * inv2h);
        st_shared<COH>(o + (nq + i) * stride, (vp - vm) * inv2h);
      };
      emit(0, axp, axm); emit(1, azp, azm); emit(2, alp, alm);
      for (int c = 0; c < NCH; ++c) {
        double q3p[3], q3m[3];
        AggD ag;
        typename M::template Saved<double> sv;
        if (owner < 0 || owner == c) {
          M::template chain_up<double>(c, trp, xp, up, ms.p, ag, sv);
          M::template chain_down<double>(sv, alp, axp, azp, q3p);
          M::template chain_up<double>(c, trm, xm, um, ms.p, ag, sv);
          M::template chain_down<double>(sv, alm, axm, azm, q3m);
        } else {
          M::template chain_up<double>(c, trp, xp, up, ms.p, ag, sv);      // unperturbed for this chain: trp == trm, same joints
          M::template chain_down<double>(sv, alp, axp, azp, q3p);
          M::template chain_down<double>(sv, alm, axm, azm, q3m);
        }
        emit(3 + 3 * c, q3p[0], q3m[0]); emit(4 + 3 * c, q3p[1], q3m[1]); emit(5 + 3 * c, q3p[2], q3m[2]);
      }
    } else {
      const SeedAcc xs{xg, col}, us{ug, col - n};
      typename M::template Trunk<Dual1> tr;
      M::template trunk_state<Dual1>(xs, tr);
      typename M::template Agg<Dual1> tot;
      M::template trunk_agg<Dual1>(ms.p, tot);
      for (int c = 0; c < NCH; ++c) {
        typename M::template Agg<Dual1> ag;
        typename M::template Saved<Dual1> sv;
        M::template chain_up<Dual1>(c, tr, xs, us, ms.p, ag, sv);
        M::agg_add(tot, ag);
      }
      Dual1 ax, az, al;
      M::template base_solve<Dual1>(tot, ax, az, al);
      auto emit = [&](int i, Dual1 acc) __attribute__((always_inline)) {
        const Dual1 vn = xs[nq + i] + dt * acc;
        const Dual1 qn = xs[i] + dt * vn;
        st_shared<COH>(o + i * stride, qn.d);
        st_shared<COH>(o + (nq + i) * stride, vn.d);
      };
      emit(0, ax); emit(1, az); emit(2, al);
      for (int c = 0; c < NCH; ++c) {
        typename M::template Agg<Dual1> ag;
        typename M::template Saved<Dual1> sv;
        Dual1 q3[3];
        M::template chain_up<Dual1>(c, tr, xs, us, ms.p, ag, sv);
        M::template chain_down<Dual1>(sv, al, ax, az, q3);
        emit(3 + 3 * c, q3[0]); emit(4 + 3 * c, q3[1]); emit(5 + 3 * c, q3[2]);
      }
    }
  }
}

// Leg models (Quad3D): one (key-point, column) item per thread, the step evaluated LEG BY LEG through accessors that
// perturb / seed one entry of the LDS copy of the nominal trajectory on the fly, every leg's six Jacobian entries stored
// as soon as they exist (two whole next-state vectors per thread never do - they cost the kernel its register
// allocation).  A perturbed input belongs to the trunk (quaternion, position, angular / linear velocity: every leg
// changes) or to ONE leg (a joint angle, rate or torque): the other legs then see identical inputs on both sides of the
// difference - evaluated once, their wrench enters both trunk sums and their six entries are the exact zeros the
// two-sided evaluation produces.  Same arithmetic, same summation order (0 + 2) + (1 + 3) as M::step: bitwise the
// Jacobians of the whole-step form.  Items are dealt key-point fastest with the columns ordered by owner, so the
// owner test is (nearly) wave-uniform.
template <class M, int JAC, bool COH = false>
__device__ __forceinline__ void large_jac_at_legs(const LView<M::n, M::m>& v, const KArgs& a, const int* list, int count,
                                                  const double* Xsrc, const double* Usrc, int xstride, int ustride,
                                                  int first = 0, int cstride = 1) {
  constexpr int n = M::n, m = M::m, nc = n + m;
  const ModelScalars<M> ms(a);
  const double h = ms.fd_h, inv2h = 1.0 / (2.0 * h), dt = ms.dt;
  for (int it = first * kLargeThreads + stage_tid(); it < count * nc; it += cstride * kLargeThreads) {
    const int rank = it / count, ki = it - rank * count;
    const int col = M::input_by_owner(rank);
    const int owner = M::leg_of_input(col);                    // -1: the trunk's own coordinates
    const int t = list[ki];
    const double* xg = Xsrc + (size_t)t * xstride;
    const double* ug = Usrc + (size_t)t * ustride;
    double* o;
    int stride;
    if (col < n) { o = v.Fx + (size_t)t * n * n + col; stride = n; }
    else { o = v.Fu + (size_t)t * n * m + (col - n); stride = m; }
    if (JAC == MI_JAC_FD_CENTRAL) {
      const PertAcc xp{xg, col, h}, up{ug, col - n, h}, xm{xg, col, -h}, um{ug, col - n, -h};
      double Rp[3][3], Rm[3][3];
      M::template rotation<double>(xp, Rp);
      M::template rotation<double>(xm, Rm);
      double Fp[3], Tp[3], Fm[3], Tm[3];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        double hFp[3], hTp[3], hFm[3], hTm[3];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int k = half + 2 * j;                           // legs 0, 2 | 1, 3
          typename M::template LegOut<double> lp, lm;
          M::template leg<double>(k, Rp, xp, up, ms.p, lp);
          if (owner < 0 || owner == k) M::template leg<double>(k, Rm, xm, um, ms.p, lm);
          else lm = lp;                                         // identical inputs: identical outputs
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const int q = 3 * k + i;
            const double jdp = xp[25 + q] + dt * lp.ja[i], jdm = xm[25 + q] + dt * lm.ja[i];
            const double jqp = xp[7 + q] + dt * jdp, jqm = xm[7 + q] + dt * jdm;
            st_shared<COH>(o + (25 + q) * stride, (jdp - jdm) * inv2h);
            st_shared<COH>(o + (7 + q) * stride, (jqp - jqm) * inv2h);
            if (j == 0) { hFp[i] = lp.fw[i]; hTp[i] = lp.tq[i]; hFm[i] = lm.fw[i]; hTm[i] = lm.tq[i]; }
            else { hFp[i] = hFp[i] + lp.fw[i]; hTp[i] = hTp[i] + lp.tq[i]; hFm[i] = hFm[i] + lm.fw[i]; hTm[i] = hTm[i] + lm.tq[i]; }
          }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          if (half == 0) { Fp[i] = hFp[i]; Tp[i] = hTp[i]; Fm[i] = hFm[i]; Tm[i] = hTm[i]; }
          else { Fp[i] = Fp[i] + hFp[i]; Tp[i] = Tp[i] + hTp[i]; Fm[i] = Fm[i] + hFm[i]; Tm[i] = Tm[i] + hTm[i]; }
        }
      }
      double tp[n], tm[n];                                      // (only the trunk's 13 entries are ever touched)
      M::template trunk<double>(xp, Fp, Tp, tp, ms.p, dt);
      M::template trunk<double>(xm, Fm, Tm, tm, ms.p, dt);
#pragma unroll
      for (int i = 0; i < 7; ++i) st_shared<COH>(o + i * stride, (tp[i] - tm[i]) * inv2h);
#pragma unroll
      for (int i = 19; i < 25; ++i) st_shared<COH>(o + i * stride, (tp[i] - tm[i]) * inv2h);
    } else {
      const SeedAcc xs{xg, col}, us{ug, col - n};
      Dual1 R[3][3];
      M::template rotation<Dual1>(xs, R);
      Dual1 F[3], Tq[3];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        Dual1 hF[3], hT[3];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int k = half + 2 * j;
          typename M::template LegOut<Dual1> lo;
          M::template leg<Dual1>(k, R, xs, us, ms.p, lo);
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const int q = 3 * k + i;
            const Dual1 jd = xs[25 + q] + dt * lo.ja[i];
            const Dual1 jq = xs[7 + q] + dt * jd;
            st_shared<COH>(o + (25 + q) * stride, jd.d);
            st_shared<COH>(o + (7 + q) * stride, jq.d);
            if (j == 0) { hF[i] = lo.fw[i]; hT[i] = lo.tq[i]; } else { hF[i] = hF[i] + lo.fw[i]; hT[i] = hT[i] + lo.tq[i]; }
          }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          if (half == 0) { F[i] = hF[i]; Tq[i] = hT[i]; } else { F[i] = F[i] + hF[i]; Tq[i] = Tq[i] + hT[i]; }
        }
      }
      Dual1 tn[n];
      M::template trunk<Dual1>(xs, F, Tq, tn, ms.p, dt);
#pragma unroll
      for (int i = 0; i < 7; ++i) st_shared<COH>(o + i * stride, tn[i].d);
#pragma unroll
      for (int i = 19; i < 25; ++i) st_shared<COH>(o + i * stride, tn[i].d);
    }
  }
}

// Dense models (whole-step plugins, the arm + ball): one (key-point, column) item per thread, the whole step evaluated at
// the perturbed / seeded point.  Xsrc / Usrc: the nominal trajectory in HBM, or its LDS copy (row strides xstride / ustride);
// first / cstride: this workgroup's share of the items when a cluster of workgroups linearizes one problem (chunks of 256
// dealt round-robin; COH: write-through stores, read by another workgroup).
template <class M, int JAC, bool COH = false>
__device__ __forceinline__ void large_jac_at(const LView<M::n, M::m>& v, const KArgs& a, const int* list, int count,
                                             const double* Xsrc, const double* Usrc, int xstride, int ustride,
                                             int first = 0, int cstride = 1) {
  constexpr int n = M::n, m = M::m, nc = n + m;
  const ModelScalars<M> ms(a);
  const double h = ms.fd_h, inv2h = 1.0 / (2.0 * h);
  for (int it = first * kLargeThreads + stage_tid(); it < count * nc; it += cstride * kLargeThreads) {
    const int ki = it / nc, col = it - ki * nc;
    const int t = list[ki];
    const double* xg = Xsrc + (size_t)t * xstride;
    const double* ug = Usrc + (size_t)t * ustride;
    double d[n];
    if (JAC == MI_JAC_FD_CENTRAL) {
      double x[n], u[m], f[n];
#pragma unroll
      for (int i = 0; i < n; ++i) x[i] = (col == i) ? xg[i] + h : xg[i];
#pragma unroll
      for (int k = 0; k < m; ++k) u[k] = (col == n + k) ? ug[k] + h : ug[k];
      M::template step<double>(x, u, d, ms.p, ms.dt);
#pragma unroll
      for (int i = 0; i < n; ++i) x[i] = (col == i) ? xg[i] - h : xg[i];
#pragma unroll
      for (int k = 0; k < m; ++k) u[k] = (col == n + k) ? ug[k] - h : ug[k];
      M::template step<double>(x, u, f, ms.p, ms.dt);
#pragma unroll
      for (int i = 0; i < n; ++i) d[i] = (d[i] - f[i]) * inv2h;
    } else {
      Dual1 xd[n], ud[m], fd[n];
#pragma unroll
      for (int i = 0; i < n; ++i) xd[i] = Dual1(xg[i], (col == i) ? 1.0 : 0.0);
#pragma unroll
      for (int k = 0; k < m; ++k) ud[k] = Dual1(ug[k], (col == n + k) ? 1.0 : 0.0);
      M::template step<Dual1>(xd, ud, fd, ms.p, ms.dt);
#pragma unroll
      for (int i = 0; i < n; ++i) d[i] = fd[i].d;
    }
    if (col < n) {
      double* o = v.Fx + (size_t)t * n * n + col;
#pragma unroll
      for (int i = 0; i < n; ++i) st_shared<COH>(o + i * n, d[i]);
    } else {
      double* o = v.Fu + (size_t)t * n * m + (col - n);
#pragma unroll
      for (int i = 0; i < n; ++i) st_shared<COH>(o + i * m, d[i]);
    }
  }
}

#if defined(MI_PROF_BACKWARD) && defined(MI_PROF_BACKWARD_LIGHT)
// light mode: only the stopwatches on either side of the two barriers of a step - busy / wait per half-step and wave, at
// next to no perturbation (a wave reads the clock where it is about to wait anyway); tools/bp_prof.py --light
#define BP_TICK(k) do { if ((k) == 2 || (k) == 3 || (k) == 12 || (k) == 11 || (k) == 6 || (k) == 8 || (k) == 7) { const long long c_ = clock64(); if (bp_acc) bp_acc[k] += c_ - bp_last; bp_last = c_; } } while (0)
#elif defined(MI_PROF_BACKWARD)
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

// In-place Gauss-Jordan inverse (no pivoting: the matrix is symmetric positive definite) of an m x m matrix held one
// ROW PER LANE (lane i of a 16-lane row holds A[i][0..m-1]); the pivot row travels by DPP row_share, no LDS, no
// barriers.  Row scaling is deferred: lane i keeps a factor s (1 until its own pivot, 1 / pivot after) and the true
// row is s * a - the elimination  a_i[j] -= (a_i[k] / p) a_k[j]  of the other lanes' rows does not see their factors,
// and the pivot lane itself only sets a[k] = 1.  On exit  A^{-1}[i][j] = s * a[j]  on lane i.
// The serial part is pivot -> reciprocal (16-cycle v_rcp_f64 + four dependent FMAs) -> multiplier -> next pivot, and the
// wave issues in order: each step updates the column of the next pivot first, starts that pivot's reciprocal and places
// the other column updates between its dependent instructions (sched_barrier pins the order).  Columns are visited in the order K+1, K+2, ..., K-1 (mod m).
// One column update  a[J] -= g * (lane K's a[J])  is ONE instruction: v_fmac_f64 is a VOP2 opcode on gfx90a+ and takes a DPP64
// row_newbcast source - v_fmac_f64_dpp a, -a(row_newbcast:K), g - where the compiler emits v_mov_b64_dpp + v_fma_f64 for
// fma(-g, row_share<K>(a), a): half the elimination's instructions (m (m - 1) updates).  (-x) * g + a and fma(-g, x, a) are the
// same bits.  Inline assembly is invisible to the compiler's hazard recognizer (a DPP read needs two wait states after a VALU
// write of the same register), so the order is fixed where it matters: the s_nop at the top of every pivot covers the
// compiler's own writes before it (the rows' assembly, the previous pivot's a[K] = select), the pivot read sits inside the block
// that updates its column, and a column written in pivot K is read again no earlier than m - 2 + 8 instructions later.
// (m < 4: the compiler's form - the elimination is bound by its serial chain there and the fixed s_nops only add to it.  Measured,
// cycles per inverse, tools/ubench/gj_fmac.hip: m = 7 845 -> 829, m = 12 2 053 -> 1 609, m = 16 3 435 -> 2 416; same bits.)
#ifndef MI_GJ_MOV_FMA
template <int m> constexpr bool kGjAsm = m >= 4;
#else
template <int m> constexpr bool kGjAsm = false;
#endif
template <int K, bool ASM>
__device__ __forceinline__ void fmac_row_share(double& a, double g) {
  if constexpr (ASM) asm("v_fmac_f64_dpp %0, -%0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(g), "n"(K));
  else a = fma(-g, row_share<K>(a), a);
}
template <int m, int K, int NTH>
__device__ __forceinline__ void gj_update(double (&a)[m], double g) {
  if constexpr (NTH < m) fmac_row_share<K, kGjAsm<m>>(a[(K + NTH) % m], g);
}
template <int m, int K, int NTH>
struct GjRest {
  static __device__ __forceinline__ void run(double (&a)[m], double g) {
    gj_update<m, K, NTH>(a, g);
    GjRest<m, K, NTH + 1>::run(a, g);
  }
};
template <int m, int K>
struct GjRest<m, K, m> {
  static __device__ __forceinline__ void run(double (&)[m], double) {}
};
// the head of pivot K: the first three column updates and the read of the NEXT pivot out of the first of them, in one block
template <int m, int K>
__device__ __forceinline__ double gj_head(double (&a)[m], double g) {
  static_assert(kGjAsm<m> && m >= 4, "three columns besides the pivot's");
  double d;
  asm("s_nop 1\n\tv_fmac_f64_dpp %1, -%1, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %2, -%2, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %3, -%3, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b64_dpp %0, %1 row_newbcast:%6 row_mask:0xf bank_mask:0xf"
      : "=&v"(d), "+v"(a[(K + 1) % m]), "+v"(a[(K + 2) % m]), "+v"(a[(K + 3) % m]) : "v"(g), "n"(K), "n"(K + 1));
  return d;
}
template <int m, int K>
struct GjOuter {
  // `inv` = 1 / pivot K, already computed; `i` = this lane's row
  static __device__ __forceinline__ void run(double (&a)[m], double& s, int i, double inv) {
    const bool piv = i == K;
    const double g = piv ? 0.0 : a[K] * inv;
    double r = 0.0;
    if constexpr (K + 1 < m) {
      double d;
      if constexpr (kGjAsm<m>) {
        d = gj_head<m, K>(a, g);                        // columns K+1 (the next pivot is in it), K+2, K+3 and the pivot read
        __builtin_amdgcn_sched_barrier(0);
        r = __builtin_amdgcn_rcp(d);
      } else {
        gj_update<m, K, 1>(a, g);                       // column K+1: the next pivot is in it
        gj_update<m, K, 2>(a, g);                       // (also covers the DPP-after-VALU wait states of the pivot read)
        __builtin_amdgcn_sched_barrier(0);
        d = row_share<K + 1>(a[K + 1]);
        r = __builtin_amdgcn_rcp(d);
        gj_update<m, K, 3>(a, g);
      }
      gj_update<m, K, 4>(a, g);
      __builtin_amdgcn_sched_barrier(0);
      double e = fma(-d, r, 1.0);
      gj_update<m, K, 5>(a, g);
      __builtin_amdgcn_sched_barrier(0);
      r = fma(r, e, r);
      gj_update<m, K, 6>(a, g);
      __builtin_amdgcn_sched_barrier(0);
      e = fma(-d, r, 1.0);
      gj_update<m, K, 7>(a, g);
      __builtin_amdgcn_sched_barrier(0);
      r = fma(r, e, r);                                 // == fast_rcp(d)
      GjRest<m, K, (8 < m ? 8 : m)>::run(a, g);
    } else {
      if constexpr (kGjAsm<m>) {
        asm volatile("s_nop 1");
        __builtin_amdgcn_sched_barrier(0);
      }
      GjRest<m, K, 1>::run(a, g);
    }
    a[K] = piv ? 1.0 : -g;
    s = piv ? inv : s;
    GjOuter<m, K + 1>::run(a, s, i, r);
  }
  static __device__ __forceinline__ void run(double (&a)[m], double& s, int i) {
    static_assert(K == 0, "entry point");
    run(a, s, i, fast_rcp(row_share<0>(a[0])));
  }
};
template <int m>
struct GjOuter<m, m> {
  static __device__ __forceinline__ void run(double (&)[m], double&, int, double) {}
};
// Positive definiteness for free: after the elimination lane i's factor s IS 1 / (pivot i), and the matrix is positive definite
// exactly when every pivot is > 0 - i.e. every row's s is (NaN or an infinite pivot fail the comparison as well).
__device__ __forceinline__ bool gj_row_positive(double s) { return s > 0.0; }

// Inverse of a Quu that is NOT positive definite, the way np.linalg.inv computes it (ilqr.py:655: the reference inverts whatever
// comes out and carries on): LAPACK's getrf + getri - LU with partial pivoting (row exchanges), inv(U), then inv(A) from
// inv(A) L = inv(U), then the exchanges undone on the columns.  (A Gauss-Jordan sweep with the same pivoting was the first
// version: measurably less accurate than LU once cond(Quu) passes 1e5 - 3e-8 against the oracle's 2e-11 on the (27, 7) plugin
// with asymmetric costs.)  The COLD path of the backward passes: entered only after the unpivoted elimination above has met
// a non-positive pivot AND the caller asked to continue (mi_ilqr_desc.on_indefinite = 1) - a positive definite Quu never pays
// for it.  It therefore works in LDS, not in registers, in plain run-time loops: a few dozen instructions and a handful of
// registers inside kernels that have none to spare (a register formulation - rows exchanged between lanes with ds_bpermute -
// cost the solve kernels up to 70 more spilled scalars and 80 B more scratch per lane on the paths that never run it; this one
// is within +-10 of the build without it).  W[i * ws + j], i, j < m: Quu on entry, its inverse on exit.  One wave; lane i < m
// owns row i (eliminations, triangular solves) or column i (row exchanges).  Ties take the lowest row like idamax; a NaN
// candidate wins (the result is NaN like the reference's).  ~m^2 dependent LDS round trips.
template <int m>
__device__ __forceinline__ void quu_inverse_pivoted(double* W_, int ws, int lane) {
  static_assert(m <= 16, "one row per lane of a 16-lane row; the permutation packs 4 bits per step");
  // W_ points into LDS: say so.  Left generic, one instantiation ((34, 12) plugin) kept a flat access and this hipcc then fails in
  // instruction selection on the aperture it needs ("Illegal instruction detected: Operand has incorrect register class").
  typedef __attribute__((address_space(3))) double lds_double_t;
  lds_double_t* const W = (lds_double_t*)W_;
  auto fence = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
  };
  const bool on = lane < m;
  const int i = on ? lane : 0;
  unsigned long long perm = 0ull;
  // ---- getrf: P A = L U in place (unit lower L below the diagonal)
#pragma unroll 1
  for (int k = 0; k < m; ++k) {
    fence();
    double best = (on && lane >= k) ? fabs(W[i * ws + k]) : -1.0;
    best = best == best ? best : __builtin_inf();
    int idx = lane;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      const double ob = __shfl_xor(best, o, 16);
      const int oi = __shfl_xor(idx, o, 16);
      const bool take = ob > best || (ob == best && oi < idx);
      best = take ? ob : best;
      idx = take ? oi : idx;
    }
    const int p = __builtin_amdgcn_readfirstlane(idx);
    perm |= (unsigned long long)p << (4 * k);
    if (p != k) {                                            // rows k and p change places (lane i: column i)
      const double t1 = W[k * ws + i], t2 = W[p * ws + i];
      fence();
      if (on) { W[k * ws + i] = t2; W[p * ws + i] = t1; }
      fence();
    }
    if (on && i > k) {                                       // lane i: row i
      const double l = W[i * ws + k] * (1.0 / W[k * ws + k]);
      W[i * ws + k] = l;
#pragma unroll 1
      for (int j = k + 1; j < m; ++j) W[i * ws + j] = fma(-l, W[k * ws + j], W[i * ws + j]);
    }
  }
  // ---- trtri: inv(U) over U, column by column (lane i: entry (i, j), from its own row of inv(U) and column j of U)
#pragma unroll 1
  for (int j = 0; j < m; ++j) {
    fence();
    const double ujj = 1.0 / W[j * ws + j];
    double acc = 0.0;
    if (on && i < j) {
#pragma unroll 1
      for (int k = i; k < j; ++k) acc = fma(W[i * ws + k], W[k * ws + j], acc);
    }
    fence();
    if (on && i < j) W[i * ws + j] = -acc * ujj;
    if (on && i == j) W[i * ws + j] = ujj;
  }
  // ---- getri: inv(A) L = inv(U), columns from the last to the first (lane i: entry (i, j))
#pragma unroll 1
  for (int j = m - 2; j >= 0; --j) {
    fence();
    double acc = 0.0;
    if (on) {
#pragma unroll 1
      for (int k = j + 1; k < m; ++k) acc = fma(W[i * ws + k], W[k * ws + j], acc);
    }
    const double base = (on && i <= j) ? W[i * ws + j] : 0.0;
    fence();
    if (on) W[i * ws + j] = base - acc;
  }
  fence();
#pragma unroll 1
  for (int k = m - 1; k >= 0; --k) {                         // the row exchanges, undone as column exchanges of the inverse (own row only)
    const int p = (int)((perm >> (4 * k)) & 15ull);
    if (p != k && on) { const double t1 = W[i * ws + k]; W[i * ws + k] = W[i * ws + p]; W[i * ws + p] = t1; }
  }
  fence();
}

// Backward Riccati pass (ilqr.py:623-667), cost expansion (:161-206) fused.
//
// Per time step, with F = [fx | fu] (n x (n+m)):
//     T1  = Vxx F[:, x]               (n x n)      9 tiles x KN k-steps
//     H   = F^T T1                    rows x: Qxx - lxx (symmetric: 6 tiles computed, 3 mirrored), rows u: Qux
//     Quu = 2R + fu^T Vxx fu          from the Vxx' accumulators of the previous step; Quu^{-1} on the solver wave
//     K   = Quu^{-1} Qux,  kappa = Quu^{-1} Qu                                            (:659-660)
//     Vxx' = Qxx - Qux^T K,  Vx' = Qx - Qux^T kappa                                        (:666-667)
// The products run as 16x16 tiles of v_mfma_f64_16x16x4_f64.  On gfx950 the fp64 matrix rate equals the fp64 VALU
// rate (64 cycles per 16x16x4 = 16 FMA/clk/SIMD, tools/ubench/mfma_cu.hip), so the point of the matrix core here is
// OPERAND DELIVERY: two 8-byte operands per lane feed 1024 FMAs, where a VALU formulation needs a (broadcast) LDS
// read per 1-2 FMAs and is LDS-issue-bound at one wave per SIMD (tools/ubench/t1.hip: 4.5-10.7 k cycles for T1
// alone) - and since round 3 most operands do not even come from LDS: see "Fused chain" below.
#ifdef MI_BACKWARD_NOINLINE
#define MI_BP_INLINE __attribute__((noinline))
#else
#define MI_BP_INLINE inline
#endif
template <class M, bool PIV = true>
__device__ MI_BP_INLINE void large_backward(const LView<M::n, M::m>& v, double* lds, long long* bp_acc = nullptr, bool lx_ready = false) {
  constexpr int n = M::n, m = M::m, nm = n + m;
  using Ly = LLay<n, m>;
  constexpr int TS = Ly::TS, VS = Ly::VS, FS = Ly::NMP, NP = Ly::NP;
  constexpr int RT = NP / 16, CT = Ly::NMP / 16;       // row tiles of an n-row matrix, col tiles of an nm-col one
  constexpr bool SPLIT = Ly::kSplit;                   // u in a column tile of its own (LLay)
  constexpr int UC = Ly::UC, KN = Ly::KN, NK = Ly::NK;
  constexpr int CX = SPLIT ? RT : CT;                  // column tiles the three matrix-core waves own
  static_assert(m % 4 == 0, "k-steps of 4 over m");
  const int tid = stage_tid(), N = v.N, wave = tid >> 6, lane = tid & 63;
  const int lr = lane & 15, lk = lane >> 4;
  const double* Q = lds + Ly::oQ;
  const double* R = lds + Ly::oR;
  const double* Qf = lds + Ly::oQf;
  const double* qn = lds + Ly::oQn;
  const double* qfn = lds + Ly::oQfn;
  double* Vxx = lds + Ly::oVxx;      // [NP][VS], rows >= n are zero
  double* Vx = lds + Ly::oVx;
  double* F = lds + Ly::oF;          // [n][FS]  = [fx | fu | 0-pad]
  double* T1 = lds + Ly::oT1;        // the area T1 = Vxx F used to occupy: staging in the prologue, then the SECOND F buffer
  double* H = lds + Ly::oH;          // the area H = F^T T1 used to occupy: staging, then the waves' exchange buffers (below)
  double* QT = lds + Ly::oQT;        // Q^T (compact layout only: the horizon's cost-gradient product)
#ifdef MI_PROF_BACKWARD
  long long bp_last = clock64();
#endif

  // terminal: Vx = 2 Qf x_T - 2 x_nom^T Qf ; Vxx = 2 Qf   (ilqr.py:203-204, :638); pads = 0
  if (tid == 0) lds[Ly::oRed + kPdFlag] = 0.0;
  for (int e = tid; e < NP * VS; e += kLargeThreads) {
    const int i = e / VS, j = e - i * VS;
    Vxx[e] = (i < n && j < n) ? 2.0 * Qf[i * n + j] : 0.0;
  }
  if constexpr (!SPLIT) { for (int e = tid; e < n * n; e += kLargeThreads) { const int i = e / n, j = e - i * n; QT[j * n + i] = Q[e]; } }
  if (tid >= n && tid < NK) Vx[tid] = 0.0;                           // pad of the k-steps
  if (tid < n) {
    const double* xT = v.X + (size_t)(N - 1) * n;
    double xr[n];
#pragma unroll
    for (int j = 0; j < n; ++j) xr[j] = xT[j];               // all loads in flight at once (L2 latency paid once)
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < n; ++j) s += (2.0 * Qf[tid * n + j]) * xr[j];
    Vx[tid] = s - qfn[tid];
  }
  __syncthreads();
  BP_TICK(13);
  // cost gradients for ALL steps, off the recursion (ilqr.py:180-181): lx_t = 2Q x_bar_t - 2 x_nom^T Q,
  // lu_t = 2R u_bar_t.  lx for the whole horizon is one (N-1) x n x n product: x_bar is copied to
  // LDS (the T1|H area is not live yet) and the product runs as 16x16 tiles on the matrix core,
  // tiles dealt round-robin to the four waves (dot products out of L2: 31 k cycles per pass; out
  // of LDS on the VALU: 17 k, LDS-bandwidth-bound; this: ~3 k).  lu is small and stays scalar.
  double* Lxu = lds + Ly::doubles;
  {
    double* Xs_ = T1;                                        // T1 and H are contiguous
    double* Us_ = F;
    // the last row tile reads up to 15 rows past step N-2: they must stay inside the staging area
    const bool staged = !SPLIT && (size_t)n * (N + 15) <= (size_t)(n + Ly::NMP) * TS && (size_t)m * (N - 1) <= (size_t)n * FS;
    if (lx_ready) {
      // the rollout of the accepted trial left lx_t, lu_t here (large_rollout)
    } else if (staged) {
     if constexpr (!SPLIT) {
      for (int e = tid; e < n * (N - 1); e += kLargeThreads) Xs_[e] = v.X[e];
      for (int e = tid; e < m * (N - 1); e += kLargeThreads) Us_[e] = v.U[e];
      __syncthreads();
      if (wave == 0) BP_TICK(5);
      constexpr int CTn = (n + 15) / 16;
      const int ntiles = ((N - 1 + 15) / 16) * CTn;
      for (int tile = wave; tile < ntiles; tile += 4) {
        const int q = tile / CTn, c = tile - q * CTn;
        TileOps<n / 4> op;                                   // A = x_bar rows, B[j][pp] = 2 Q[pp][j] = 2 Q^T[j][pp]
        op.load(Xs_ + (16 * q + lr) * n + lk, 4, QT + lk * n + 16 * c + lr, 4 * n);
#pragma unroll
        for (int ks = 0; ks < n / 4; ++ks) op.bv[ks] *= 2.0;
        const int pp = 16 * c + lr;
        const double c0 = pp < n ? -qn[pp] : 0.0;
        d4_t acc = {c0, c0, c0, c0};
        acc = op.run(acc);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int tt = 16 * q + lk + 4 * reg;
          if (tt < N - 1 && pp < n) lxu_store(v, Lxu, tt * nm + pp, acc[reg]);
        }
      }
      for (int idx = tid; idx < (N - 1) * m; idx += kLargeThreads) {
        const int tt = idx / m, a_ = idx - tt * m;
        double s_ = 0.0;
#pragma unroll
        for (int j = 0; j < m; ++j) s_ += (2.0 * R[a_ * m + j]) * Us_[tt * m + j];
        lxu_store(v, Lxu, tt * nm + n + a_, s_);
      }
     }
    } else {
      for (int idx = tid; idx < (N - 1) * nm; idx += kLargeThreads) {   // long horizons (and the split layout): straight from L2
        const int tt = idx / nm, pp = idx - tt * nm;
        double s_;
        if (pp < n) {
          const double* xg = v.X + (size_t)tt * n;
          s_ = -qn[pp];
          for (int j = 0; j < n; ++j) s_ += (2.0 * Q[pp * n + j]) * xg[j];
        } else {
          const double* ug = v.U + (size_t)tt * m;
          s_ = 0.0;
          for (int j = 0; j < m; ++j) s_ += (2.0 * R[(pp - n) * m + j]) * ug[j];
        }
        lxu_store(v, Lxu, idx, s_);
      }
    }
    if (wave == 0) BP_TICK(6);
    __syncthreads();
    if (wave == 0) BP_TICK(7);
    // the zero padding the tiles rely on (F's pad rows and columns, in both buffers; the exchange area where H used to be)
    for (int e = tid; e < NK * FS; e += kLargeThreads) { F[e] = 0.0; F[(Ly::oT1 - Ly::oF) + e] = 0.0; }   // both F buffers (the second one is the T1 area)
    for (int e = tid; e < Ly::NMP * TS; e += kLargeThreads) H[e] = 0.0;
  }
  __syncthreads();
  BP_TICK(14);
  // The solver wave runs F's pipeline: a WHOLE F = [fx_t | fu_t] (contiguous n*n and n*m blocks in HBM) is fetched as
  // 16-byte pairs into registers three steps ahead and published to one of two LDS buffers two steps ahead, always in
  // the second half of a step - the three matrix-core waves only touch global memory to store K.
  // (16-byte pairs where the rows allow it - n, m even; single doubles otherwise, e.g. n = 37)
  // fx and fu have their own width: with an odd n (37) the rows of fx do not hold whole pairs, but HBM still delivers
  // pairs - 16-byte loads that may straddle two rows, stored as two doubles (kSplitX) - and fu (m even) keeps whole pairs:
  // half the load instructions of the single-double form on the wave that binds the second half-step of n = 37.
  constexpr bool kEvenBase = FS % 2 == 0 && Ly::oF % 2 == 0 && Ly::oT1 % 2 == 0;
  constexpr int WX = (n % 2 == 0 && kEvenBase) ? 2 : 1;                  // doubles per LDS store of fx
  constexpr int WU = (m % 2 == 0 && UC % 2 == 0 && kEvenBase) ? 2 : 1;   // ... of fu
  constexpr bool kSplitX = WX == 1 && n * n >= 2;                        // fx: pair loads, two single stores
  constexpr int PFX = kSplitX ? (n * n + 1) / 2 : n * n / WX, PFU = n * m / WU;
  constexpr int NFX = (PFX + 63) / 64, NFU = (PFU + 63) / 64;
  typedef double d2_t __attribute__((ext_vector_type(2)));
  using fx_t = std::conditional_t<(WX == 2 || kSplitX), d2_t, double>;
  using fu_t = std::conditional_t<WU == 2, d2_t, double>;
  fx_t frx[NFX];
  fu_t fru[NFU];
  int fx_src[kSplitX ? NFX : 1], fx_off[NFX], fx_off1[kSplitX ? NFX : 1], fu_off[NFU];   // (split: first element in HBM;) LDS offsets (doubles)
#pragma unroll
  for (int r = 0; r < NFX; ++r) {
    int e = (kSplitX ? 2 : WX) * (lane + 64 * r);
    if constexpr (kSplitX) {
      e = e < n * n - 1 ? e : n * n - 2;                                 // (the last pair overlaps its neighbour: same values stored twice)
      fx_src[r] = e;
      fx_off1[r] = ((e + 1) / n) * FS + ((e + 1) % n);
    } else {
      e = e < n * n ? e : n * n - WX;
    }
    fx_off[r] = (e / n) * FS + (e % n);
  }
#pragma unroll
  for (int r = 0; r < NFU; ++r) { int e = WU * (lane + 64 * r); e = e < n * m ? e : n * m - WU; fu_off[r] = (e / m) * FS + UC + (e % m); }
  auto fetch = [&](int t) __attribute__((always_inline)) {
    const double* fxg = v.Fx + (size_t)t * n * n;
    const fu_t* fug = reinterpret_cast<const fu_t*>(v.Fu + (size_t)t * n * m);
#pragma unroll
    for (int r = 0; r < NFX; ++r) {
      if constexpr (kSplitX) __builtin_memcpy(&frx[r], fxg + fx_src[r], 16);   // (8-byte aligned: an unaligned 16-byte load)
      else { const int pi = lane + 64 * r; frx[r] = reinterpret_cast<const fx_t*>(fxg)[pi < PFX ? pi : PFX - 1]; }
    }
#pragma unroll
    for (int r = 0; r < NFU; ++r) { const int pi = lane + 64 * r; fru[r] = fug[pi < PFU ? pi : PFU - 1]; }
  };
  auto publish = [&](double* Fb) __attribute__((always_inline)) {       // clamped duplicates rewrite the last pair with itself
#pragma unroll
    for (int r = 0; r < NFX; ++r) {
      if constexpr (kSplitX) { Fb[fx_off[r]] = frx[r][0]; Fb[fx_off1[r]] = frx[r][1]; }
      else *reinterpret_cast<fx_t*>(Fb + fx_off[r]) = frx[r];
    }
#pragma unroll
    for (int r = 0; r < NFU; ++r) *reinterpret_cast<fu_t*>(Fb + fu_off[r]) = fru[r];
  };
  // =====================================================================================================
  // Fused chain.  The D (result) layout of one 16x16x4 product IS the B-operand layout of the next: lane
  // (lr, lk) receives D[4 reg + lk][lr] and supplies B[4 ks + lk][lr], so result register `reg` of row tile q is
  // k-step 4q + reg of a product that contracts over those rows - and, the matrix being symmetric, also the A operand
  // of a product whose ROWS are the tile's columns.  Every matrix-core wave therefore keeps ITS column tile in
  // registers from one product to the next, T1 -> H -> K -> Vxx' -> (Vxx' fu) -> its share of Quu, and only three
  // small things cross waves per step: the m rows of Qux (every wave needs all of Qux^T as an A operand), Vxx' (the
  // A operand of the next step's T1) and the three partial Quu tiles (one per matrix-core wave, summed by the solver).
  //
  //   step t, first half                                             | second half
  //   matrix-core wave w:  T1[:, w] = Vxx F_t[:, w]                  | K[:, w] = Quu^{-1} Qux[:, w]  (-> HBM, :660)
  //                        H[:, w]  = F_t^T T1[:, w]                 | Vxx'[:, w] = Qxx[:, w] + 2Q - Qux^T K[:, w]  (:667) -> LDS
  //                        Qux[:, w] -> LDS                          | T1u[w, :] = Vxx'[w, :] fu_{t-1},  P_w = fu_{t-1}[w, :]^T T1u[w, :] -> LDS
  //   solver wave:         Quu_t = 2R + P_0 + P_1 + P_2  (:654)      | kappa = Quu^{-1} Qu (:659), dV (:663),
  //                        Quu^{-1}: Gauss-Jordan, a row per lane    | Vx' = Qx - Qux^T kappa (:666)
  //                        first-order column l + F_t^T Vx (:651-652)| F_{t-2} -> LDS, prefetch of F_{t-3}
  // Two barriers per step, no T1 / H / Y round trips through LDS, no triangular solves; the solver wave's chain
  // (Quu -> inverse) runs beside the two big products instead of between them.  F is double-buffered in LDS (the
  // second buffer is the old T1 area): F_t is read until the middle of step t and F_{t-1} from there on.
  // =====================================================================================================
  constexpr int SS = 17;
  static_assert(RT == 3 && CX == 3 && CT - CX <= 1 && 2 * m <= n && m <= 16, "wave roles below: three matrix-core waves + the solver wave");
  static_assert(16 * (CT - 1) <= UC && UC + m <= 16 * CT, "Quu lies inside the last diagonal tile of H");
  constexpr int QO = UC - 16 * (CT - 1);                      // Quu's offset inside that tile
  static_assert(QO % 4 == 0, "the rows of Qux are whole result registers");
  constexpr int R0 = QO / 4, MK = m / 4;
  constexpr int QS = NP, WSS = m <= 12 ? 13 : 17, PS = 16 * SS;   // row strides (QS = 48: the four rows of a k-step on disjoint banks; WSS odd, > m)
  constexpr int FB1 = Ly::oT1 - Ly::oF;                       // F_t lives in buffer (N - 2 - t) & 1; the second buffer is the T1 area
  double* QuxS = H;                                           // [m][QS]   Qux_t, exchanged between the waves
  double* Ws = QuxS + m * QS;                                 // [16][WSS] Quu^{-1}, rows >= m zero (an A operand)
  double* Kap = Ws + 16 * WSS;                                // [m]       kappa_t
  double* Fo = Kap + m + (m & 1);                             // [NMP]     first-order column: Qx (rows < n), Qu (rows UC..)
  double* Pq = Fo + Ly::NMP;                                  // [3][16][SS] the waves' shares of Quu - luu
  double* Xt = Pq + 3 * PS;                                   // [3][16][SS] the off-diagonal Qxx tiles, transposed for their mirror's owner
  static_assert(m * QS + 16 * WSS + m + 1 + Ly::NMP + 6 * PS <= Ly::NMP * TS, "the exchange buffers live in the H area");
  static_assert(NK * FS <= NK * TS, "the second F buffer lives in the T1 area");
  constexpr int NG = 3, GS = (KN + NG - 1) / NG;              // operand loads run a group of k-steps ahead of the MFMAs
  const d4_t zero4 = {0.0, 0.0, 0.0, 0.0};
  auto wave_lds_fence = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
  };
  for (int e = tid; e < 16 * WSS; e += kLargeThreads) Ws[e] = 0.0;
  // the first two F's, fetched by two waves at once (one L2 round trip - after a clustered linearization the Jacobians
  // come from other XCDs' write-through stores - instead of two); the solver wave then has the third in flight
  if (wave == 3) {
    fetch(N - 2); publish(F);
    if (N >= 4) fetch(N - 4);
  } else if (wave == 2 && N >= 3) {
    fetch(N - 3); publish(F + FB1);
  }
  __syncthreads();
  BP_TICK(15);

  // ------------------------------------------------------------------------------------------------------------
  // Matrix-core wave W (a compile-time role: every tile offset is an immediate and every register index static).
  // Of the symmetric Qxx block of H it computes the diagonal tile (W, W) and the tile ((W+1)%3, W); the third tile of
  // its column, ((W+2)%3, W), is the mirror of what wave (W+2)%3 computes and crosses through LDS transposed.
  // ------------------------------------------------------------------------------------------------------------
  auto matrix_role = [&](auto wc) __attribute__((always_inline)) {
    constexpr int W_ = decltype(wc)::value;
    constexpr int Q1 = (W_ + 1) % 3, Q2 = (W_ + 2) % 3;       // row tiles: off-diagonal computed here / received
    constexpr bool kHasU = SPLIT;                            // u's rows of H are a tile of their own (row tile CT - 1)
    // where this wave finds the rows of Qux among its result tiles (compact layout: inside x's last row tile)
    constexpr bool kQuxDiag = !SPLIT && W_ == CT - 1, kQuxOff = !SPLIT && Q1 == CT - 1;
    const int col = 16 * W_ + lr;
    const bool col_ok = col < n;
    // This wave's share of Quu_ts - luu = fu^T Vxx fu: vc = its column tile of the (symmetric) Vxx in the D layout,
    // i.e. the A operand of  T1u[16W + r][a] = sum_k Vxx[16W + r][k] fu[k][a];  then  P_W = fu[16W.., :]^T T1u[16W.., :].
    auto quu_share = [&](const d4_t (&vc)[RT], const double (&fu_)[KN]) __attribute__((always_inline)) {
      d4_t tu = zero4;
#pragma unroll
      for (int ks = 0; ks < KN; ++ks) tu = __builtin_amdgcn_mfma_f64_16x16x4f64(vc[ks >> 2][ks & 3], fu_[ks], tu, 0, 0, 0);
      d4_t pw = zero4;
      // rows 16W + 4 reg + lk of T1u = k-step 4W + reg; k-steps past the contraction length do not exist
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
        if (4 * W_ + reg < KN) pw = __builtin_amdgcn_mfma_f64_16x16x4f64(fu_[4 * W_ + reg], tu[reg], pw, 0, 0, 0);
      double* pd = Pq + W_ * PS + lk * SS + lr;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) pd[4 * reg * SS] = pw[reg];
    };
    auto load_fu = [&](double (&fu_)[KN], const double* Fb) __attribute__((always_inline)) {
      const double* ub = Fb + lk * FS + 16 * (CT - 1) + lr;  // lane (lr, lk): F[4 ks + lk][u tile column lr]
#pragma unroll
      for (int ks = 0; ks < KN; ++ks) fu_[ks] = ub[ks * 4 * FS];
    };
    // the terminal Vxx, this wave's column tile in the D layout, and its share of the first Quu
    d4_t vc[RT];
#pragma unroll
    for (int q = 0; q < RT; ++q)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) vc[q][reg] = col_ok ? Vxx[(16 * q + 4 * reg + lk) * VS + col] : 0.0;
    {
      double fu_[KN];
      load_fu(fu_, F);
      quu_share(vc, fu_);
    }
    // 2 lxx = 2Q entries this lane adds to its Vxx tiles: constant over the sweep
    double q2[RT][4];
#pragma unroll
    for (int q = 0; q < RT; ++q)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int row = 16 * q + 4 * reg + lk;
        q2[q][reg] = (row < n && col_ok) ? 2.0 * Q[row * n + col] : 0.0;
      }
    __syncthreads();
    for (int t = N - 2; t >= 0; --t) {
      BP_TICK(0);
      const double* Fc = F + ((N - 2 - t) & 1) * FB1;         // F_t
      const double* Fn = F + ((N - 1 - t) & 1) * FB1;         // F_{t-1}, published a step ago
      // ---- T1[:, W] = Vxx F[:, W].  Row tile W of Vxx is this wave's own column tile, mirrored: already in registers
      //      (vc, from the previous step); the other row tiles and F come from LDS, a group of k-steps ahead.
      //      THE LAST ROW TILE IS THIN: of its 16 rows only NK - 32 (4 or 8) exist as k-steps of the next product, so it
      //      is computed in groups of four rows by v_mfma_f64_4x4x4_4b (four 4x4x4 blocks = 4 rows x 16 columns, 16
      //      cycles instead of 64; tools/ubench/mfma4x4.hip): block b, row i, k at lane 16k + 4b + i for A and B - with
      //      every block given the same four rows of Vxx, B is the 16x16x4 product's operand register unchanged, and the
      //      result lands lane for lane where result register g of the 16x16x4 tile would.
      constexpr int LT = RT - 1, G4 = (NK - 16 * LT) / 4;      // the thin tile and its groups of four rows
      static_assert(G4 >= 1 && G4 <= 4, "thin last row tile: groups of four rows");
      constexpr int F1 = (W_ == 1) ? Q2 : Q1;                  // full row tiles read from LDS: one (two for the wave that owns the thin tile)
      constexpr bool kTwoFull = W_ == LT;
      const double* a1 = Vxx + (16 * F1 + lr) * VS + lk;
      const double* a2 = Vxx + (16 * Q2 + lr) * VS + lk;     // (kTwoFull only)
      const double* at = Vxx + (16 * LT + (lr & 3)) * VS + lk;
      const double* b_base = Fc + lk * FS + 16 * W_ + lr;    // also A = F^T of the diagonal tile: A[p][k] = F[k][16W + p]
      const double* f1 = Fc + lk * FS + 16 * Q1 + lr;        // A = F^T, row tile Q1
      const double* f3 = Fc + lk * FS + 16 * (CT - 1) + lr;  // A = F^T, u's row tile (split layout)
      double va1[KN], va2[kTwoFull ? KN : 1], vt[G4][KN], fb[KN], fo[KN], fuu[kHasU ? KN : 1];
      auto load_group = [&](int g) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = g * GS; ks < (g + 1) * GS && ks < KN; ++ks) {
          fb[ks] = b_base[ks * 4 * FS];
          va1[ks] = a1[4 * ks];
          if constexpr (kTwoFull) va2[ks] = a2[4 * ks];
#pragma unroll
          for (int g4 = 0; g4 < G4; ++g4) vt[g4][ks] = at[4 * g4 * VS + 4 * ks];
        }
      };
      auto load_fa = [&](int g) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = g * GS; ks < (g + 1) * GS && ks < KN; ++ks) { fo[ks] = f1[ks * 4 * FS]; if constexpr (kHasU) fuu[ks] = f3[ks * 4 * FS]; }
      };
      d4_t accA[RT];
      double thin[G4];
#pragma unroll
      for (int q = 0; q < RT; ++q) accA[q] = zero4;
#pragma unroll
      for (int g4 = 0; g4 < G4; ++g4) thin[g4] = 0.0;
      load_group(0);
      load_group(1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if (g + 2 < NG) load_group(g + 2); else load_fa(g + 2 - NG);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = g * GS; ks < (g + 1) * GS && ks < KN; ++ks) {
          if constexpr (!kTwoFull) accA[W_] = __builtin_amdgcn_mfma_f64_16x16x4f64(vc[ks >> 2][ks & 3], fb[ks], accA[W_], 0, 0, 0);
          accA[F1] = __builtin_amdgcn_mfma_f64_16x16x4f64(va1[ks], fb[ks], accA[F1], 0, 0, 0);
          if constexpr (kTwoFull) accA[Q2] = __builtin_amdgcn_mfma_f64_16x16x4f64(va2[ks], fb[ks], accA[Q2], 0, 0, 0);
#pragma unroll
          for (int g4 = 0; g4 < G4; ++g4) thin[g4] = __builtin_amdgcn_mfma_f64_4x4x4f64(vt[g4][ks], fb[ks], thin[g4], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int g4 = 0; g4 < G4; ++g4) accA[LT][g4] = thin[g4];
      BP_TICK(1);
      // ---- H[:, W] = F^T T1[:, W], T1 from the accumulators: the diagonal tile, one off-diagonal tile, u's rows
      d4_t bd = zero4, bo = zero4, bu = zero4;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if (g + 2 < NG) load_fa(g + 2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = g * GS; ks < (g + 1) * GS && ks < KN; ++ks) {
          bd = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[ks], accA[ks >> 2][ks & 3], bd, 0, 0, 0);
          bo = __builtin_amdgcn_mfma_f64_16x16x4f64(fo[ks], accA[ks >> 2][ks & 3], bo, 0, 0, 0);
          if constexpr (kHasU) bu = __builtin_amdgcn_mfma_f64_16x16x4f64(fuu[ks], accA[ks >> 2][ks & 3], bu, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // the off-diagonal tile, transposed, for the owner of its mirror; the rows of Qux this wave has -> every wave
      {
        double* xd = Xt + W_ * PS + lr * SS + lk;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) xd[4 * reg] = bo[reg];
        if constexpr (kHasU || kQuxDiag || kQuxOff) {
          const d4_t& src = kHasU ? bu : (kQuxDiag ? bd : bo);
#pragma unroll
          for (int j = 0; j < MK; ++j) QuxS[(4 * j + lk) * QS + col] = src[R0 + j];
        }
        if constexpr (!SPLIT && W_ == CT - 1) {
          // compact layout: nobody computes the tile (CT-1, Q1) that holds Qux's columns of tile Q1 - they are the
          // mirror of this wave's off-diagonal tile (Q1, CT-1): rows 4 reg + lk, columns QO.. = u
          if (lr >= QO) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) QuxS[(lr - QO) * QS + 16 * Q1 + 4 * reg + lk] = bo[reg];
          }
        }
      }
      BP_TICK(2);
      lds_barrier();
      BP_TICK(3);
      // ---- K[:, W] = Quu^{-1} Qux[:, W] (:660), Vxx'[:, W] = Qxx[:, W] + 2Q - Qux^T K[:, W] (:667)
      double wa[MK], qux[MK], qa[RT][MK], fu_[KN];
      d4_t cm;
#pragma unroll
      for (int j = 0; j < MK; ++j) {
        // Quu^{-1} enters K as an EXACTLY symmetric matrix: its upper triangle, mirrored.  The elimination's W is symmetric only to
        // eps * cond(Quu), and this pass uses Vxx' both as computed and transposed (a wave's column tile is its row tile too): with a
        // nearly singular Quu that asymmetry - fed back through -Qux^T K into Vxx' - grows ~1.5 x per step and takes K, kappa with it
        // (planar quadruped, dt = 1.5e-3, N = 148: kappa_0 1.4e3 against 2.5 in extended precision, and a line search that then
        // fails where the reference's succeeds; with a symmetric W the pass stays on the extended-precision gains there - K_0 within
        // 0.2 where the fp64 reference is off by 16 x: tools/diag/n148_continue.py).  Same three LDS reads, another address.
        const int k_ = 4 * j + lk;
        wa[j] = Ws[lr < m ? (lr <= k_ ? lr * WSS + k_ : k_ * WSS + lr) : lr * WSS + k_];
        qux[j] = QuxS[(4 * j + lk) * QS + col];
      }
#pragma unroll
      for (int q = 0; q < LT; ++q)
#pragma unroll
        for (int j = 0; j < MK; ++j) qa[q][j] = -QuxS[(4 * j + lk) * QS + 16 * q + lr];
      double qt[G4][MK];                                     // the thin row tile's A operand: four rows per group, every block the same
#pragma unroll
      for (int g4 = 0; g4 < G4; ++g4)
#pragma unroll
        for (int j = 0; j < MK; ++j) qt[g4][j] = -QuxS[(4 * j + lk) * QS + 16 * LT + 4 * g4 + (lr & 3)];
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) cm[reg] = Xt[Q2 * PS + (4 * reg + lk) * SS + lr];
      if (t > 0) load_fu(fu_, Fn);
      __builtin_amdgcn_sched_barrier(0);
      d4_t kt = zero4;
#pragma unroll
      for (int j = 0; j < MK; ++j) kt = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[j], qux[j], kt, 0, 0, 0);
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        vc[W_][reg] = bd[reg] + q2[W_][reg];
        vc[Q1][reg] = bo[reg] + q2[Q1][reg];
        vc[Q2][reg] = cm[reg] + q2[Q2][reg];
      }
      double thd[G4];
#pragma unroll
      for (int g4 = 0; g4 < G4; ++g4) thd[g4] = vc[LT][g4];
#pragma unroll
      for (int j = 0; j < MK; ++j) {
#pragma unroll
        for (int q = 0; q < LT; ++q) vc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[q][j], kt[j], vc[q], 0, 0, 0);
#pragma unroll
        for (int g4 = 0; g4 < G4; ++g4) thd[g4] = __builtin_amdgcn_mfma_f64_4x4x4f64(qt[g4][j], kt[j], thd[g4], 0, 0, 0);
      }
      vc[LT] = zero4;                                        // (rows past the thin tile's groups: never k-steps, never stored)
#pragma unroll
      for (int g4 = 0; g4 < G4; ++g4) vc[LT][g4] = thd[g4];
      if (col_ok) {
        double* Kg = v.K + (size_t)t * m * n + col;          // K_t[4 reg + lk][col]
#pragma unroll
        for (int j = 0; j < MK; ++j) Kg[(4 * j + lk) * n] = kt[j];
      }
      double* d_base = Vxx + lk * VS + col;
#pragma unroll
      for (int q = 0; q < RT; ++q) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int ib = 16 * q + 4 * reg;
          const bool ok = col_ok && (ib + 3 < n || (ib < n && ib + lk < n));
          if (ok) d_base[ib * VS] = vc[q][reg];
        }
      }
      BP_TICK(4);
      // ---- this wave's share of the NEXT step's Quu, from the Vxx' it holds (rows / columns past n: the k-steps stop
      //      at the contraction length, and what lies between n and it is exact zeros - F's padding)
      if (t > 0) quu_share(vc, fu_);
      BP_TICK(12);
      lds_barrier();
      BP_TICK(11);
    }
  };

  // ------------------------------------------------------------------------------------------------------------
  // Solver wave: Quu = 2R + the three shares (:654), inverted one row per lane; the first-order terms; F's pipeline.
  // ------------------------------------------------------------------------------------------------------------
  auto solver_role = [&]() __attribute__((always_inline)) {
    // first-order column of step ts: l_{x,u} + F^T Vx (:651-652); F_ts (buffer Fb) and Vx are in LDS
    // (s0 = l_{x,u} of that step, fetched by the caller at the TOP of the step: an LDS read, or - long horizons - an L2 read
    //  whose latency then hides behind the elimination)
    auto first_order = [&](double s0, const double* Fb) __attribute__((always_inline)) {
      if (lane < nm) {
        double s = s0;
        const int hp = lane < n ? lane : UC + (lane - n);    // this entry's column of F
        // chunks of the contraction (rows >= n: zeros), software-pipelined: the next chunk's LDS reads are in flight
        // while this chunk's multiply-adds run - one exposed LDS latency per call instead of one per chunk
        constexpr int CH = (NK % 12 == 0) ? 12 : ((NK % 10 == 0) ? 10 : 4), NC = NK / CH;
        double fv[2][CH], vv[2][CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) { fv[0][k] = Fb[k * FS + hp]; vv[0][k] = Vx[k]; }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          if (c + 1 < NC) {
#pragma unroll
            for (int k = 0; k < CH; ++k) { fv[(c + 1) & 1][k] = Fb[((c + 1) * CH + k) * FS + hp]; vv[(c + 1) & 1][k] = Vx[(c + 1) * CH + k]; }
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int k = 0; k < CH; ++k) s += fv[c & 1][k] * vv[c & 1][k];
          __builtin_amdgcn_sched_barrier(0);
        }
        Fo[hp] = s;
      }
    };
    first_order(lane < nm ? lxu_load(v, Lxu, (N - 2) * nm + lane) : 0.0, F);
    // this lane's row of luu = 2R (lanes >= m of each 16-lane row shadow the last row)
    const int si = lr < m ? lr : m - 1;
    double r2[m];
#pragma unroll
    for (int j = 0; j < m; ++j) r2[j] = 2.0 * R[si * m + j];
    __syncthreads();
    for (int t = N - 2; t >= 0; --t) {
      BP_TICK(0);
      const double* Fc = F + ((N - 2 - t) & 1) * FB1;         // F_t
      const double* Fn = F + ((N - 1 - t) & 1) * FB1;         // F_{t-1}
      const double lx_next = (t > 0 && lane < nm) ? lxu_load(v, Lxu, (t - 1) * nm + lane) : 0.0;
      double arow[m];
      {
        const double* p0 = Pq + (QO + si) * SS + QO;
#pragma unroll
        for (int j = 0; j < m; ++j) arow[j] = r2[j] + ((p0[j] + p0[PS + j]) + p0[2 * PS + j]);
      }
      BP_TICK(5);
      double sc = 1.0;
      GjOuter<m, 0>::run(arow, sc, si);                      // Quu^{-1}[si][j] = sc * arow[j]
      if (!gj_row_positive(sc)) lds[Ly::oRed + kPdFlag] = 1.0;   // Quu not positive definite (read by the kernel after the pass: MI_STATUS_NOT_PD)
      // (PIV: kernels instantiated WITH the cold path - on_indefinite = 1 or cost matrices that are not symmetric.  The default
      //  kernels do not carry it: ~150 instructions on a path that never runs cost the n = 36 / 37 solves 2 - 5 % through the
      //  register allocation of the phases around them - same-box A/B)
      if (PIV && v.pd_continue && __any(!gj_row_positive(sc))) {    // cold path: the reference's inverse of an indefinite Quu
        const double* p0 = Pq + (QO + si) * SS + QO;
        if (lane < m) {
#pragma unroll
          for (int j = 0; j < m; ++j) Ws[lane * WSS + j] = r2[j] + ((p0[j] + p0[PS + j]) + p0[2 * PS + j]);
        }
        quu_inverse_pivoted<m>(Ws, WSS, lane);
#pragma unroll
        for (int j = 0; j < m; ++j) arow[j] = Ws[si * WSS + j];
        sc = 1.0;
      }
      if (lane < m) {
#pragma unroll
        for (int j = 0; j < m; ++j) Ws[lane * WSS + j] = sc * arow[j];
      }
      BP_TICK(6);
      lds_barrier();
      BP_TICK(8);
      // kappa = Quu^{-1} Qu (:659), dV = Qu^T kappa (:663), Vx' = Qx - Qux^T kappa (:666)
      double kp = 0.0;
      {
        double qu[m];
#pragma unroll
        for (int j = 0; j < m; ++j) qu[j] = Fo[UC + j];
#pragma unroll
        for (int j = 0; j < m; ++j) kp = fma(arow[j], qu[j], kp);
        kp *= sc;
        const double dv = row16_sum(lr < m ? Fo[UC + si] * kp : 0.0);
        if (lane < m) { Kap[lane] = kp; v.kap[(size_t)t * m + lane] = kp; }
        if (lane == 0) v.dV[t] = dv;
      }
      wave_lds_fence();
      if (lane < n) {
        double qc[m], kc[m];
#pragma unroll
        for (int a_ = 0; a_ < m; ++a_) { qc[a_] = QuxS[a_ * QS + lane]; kc[a_] = Kap[a_]; }
        double s = Fo[lane];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a_ = 0; a_ < m; ++a_) s -= qc[a_] * kc[a_];
        Vx[lane] = s;
      }
      BP_TICK(9);
      if (t > 1) {
        publish(const_cast<double*>(Fc));                    // F_{t-2} replaces F_t, which nobody reads any more
        if (t > 2) fetch(t - 3);
      }
      BP_TICK(10);
      if (t > 0) {
        wave_lds_fence();
        first_order(lx_next, Fn);                            // the next step's, from the Vx' just formed
      }
      BP_TICK(7);
      lds_barrier();
      BP_TICK(11);
    }
  };

  if (wave == 0) matrix_role(std::integral_constant<int, 0>{});
  else if (wave == 1) matrix_role(std::integral_constant<int, 1>{});
  else if (wave == 2) matrix_role(std::integral_constant<int, 2>{});
  else solver_role();
}

// Backward Riccati pass (ilqr.py:623-667, cost expansion :161-206 fused) for MID-SIZE models: 1 <= n <= 32 (one or two 16-row
// tiles of Vxx), any 1 <= m <= 16 - a quadrotor (12, 4), a 7-joint arm (14, 7), the arm + free body of kinova_gen3.py /
// panda_fr3.py (27, 7).  The same matrix-core formulation as large_backward, cut differently: u ALWAYS has a column tile of
// its own (zero-padded to 16 - the padding never reaches a result: pad columns of F are zero, so the pad rows of Qux, the pad
// rows / columns of Quu - luu and of Quu^{-1} are exact zeros), and every wave owns one COLUMN TILE of F = [fx | fu]:
//
//   step t, first half                                                    | second half
//   x-wave w (w < RT):  T1[:, w] = Vxx F_t[:, w]            (accumulators)  | K[:, w] = Quu^{-1} Qux[:, w]  -> HBM        (:660)
//                       Qxx[:, w] - lxx = F_t[:, x]^T T1[:, w]             | Vxx'[:, w] = Qxx[:, w] + 2Q - Qux^T K[:, w] -> LDS (:667)
//                       Qux[:, w] = fu_t^T T1[:, w]  -> LDS                |
//                                                                          | its share of Quu_{t-1}: fu_{t-1}^T Vxx' fu_{t-1} -> LDS
//   u-wave:             Quu = 2R + the x-waves' shares            (:654)   | Vx' = Qx - Qux^T kappa (:666); first-order column of
//                       Quu^{-1}: Gauss-Jordan, one row per lane (GjOuter) | step t-1: l + F_{t-1}^T Vx' (:651-652)
//                       -> LDS; kappa = Quu^{-1} Qu (:659), dV (:663)      |
//   pipeline wave:      (F_{t-3} in flight from HBM)                       | F_{t-2} -> the LDS buffer F_t leaves, fetch F_{t-3}
//
// The D layout of one 16x16x4 product is the B-operand layout of the next (large_backward: "Fused chain"), so T1, the H
// column and K stay in registers; Vxx is read and written in full (every entry of Vxx' is computed once, by the wave that owns
// its column, like the reference's dense update); only Quu is formed from the transposed column tiles (= the transpose of a
// matrix that is symmetric up to round-off), which takes it off the step's critical path.  Two barriers per step.
template <class M, bool PIV = true>
__device__ MI_BP_INLINE void mid_backward(const LView<M::n, M::m>& v, double* lds, bool lx_ready = false, bool xu_staged = false) {
  constexpr int n = M::n, m = M::m, nm = n + m;
  using Ly = LLay<n, m>;
  static_assert(Ly::kMid && Ly::kSplit && m >= 1 && m <= 16, "mid-size family: n <= 32, m <= 16");
  constexpr int VS = Ly::VS, FS = Ly::NMP, NP = Ly::NP, RT = NP / 16, UC = Ly::UC, KN = Ly::KN, NK = Ly::NK;
  constexpr int MK = (m + 3) / 4;                            // k-steps over the controls
  constexpr int WU = RT, WP = RT + 1;                        // the u-wave and the pipeline wave
  static_assert(RT >= 1 && RT <= 2 && WP <= 3, "one or two x-waves + the u-wave + the pipeline wave");
  const int tid = stage_tid(), N = v.N, wave = tid >> 6, lane = tid & 63;
  const int lr = lane & 15, lk = lane >> 4;
  const double* Q = lds + Ly::oQ;
  const double* R = lds + Ly::oR;
  const double* Qf = lds + Ly::oQf;
  const double* qn = lds + Ly::oQn;
  const double* qfn = lds + Ly::oQfn;
  double* Vxx = lds + Ly::oVxx;      // [NP][VS], rows / columns >= n zero
  double* Vx = lds + Ly::oVx;        // [NK], entries >= n zero
  double* F = lds + Ly::oF;          // [NK][FS] = [fx | 0 | fu | 0], two buffers (the second one in the T1 area)
  double* H = lds + Ly::oH;          // exchange buffers
  double* Lxu = lds + Ly::doubles;   // [N-1][n+m] cost gradients lx_t | lu_t
  constexpr int FB1 = Ly::oT1 - Ly::oF;
  static_assert(NK * FS <= NK * Ly::TS, "the second F buffer lives in the T1 area");
  constexpr int QS = 48, WSS = 17, SS = 17;
  double* QuxS = H;                  // [16][QS]  Qux_t (rows >= m zero)
  double* Ws = QuxS + 16 * QS;       // [16][WSS] Quu^{-1} (rows / columns >= m zero)
  double* Kap = Ws + 16 * WSS;       // [16]      kappa_t
  double* Fo = Kap + 16;             // [FS]      first-order column: Qx (entries < n), Qu (entries UC..UC+m)
  double* Pq = Fo + FS;              // [RT][16][SS] the x-waves' shares of Quu - luu (accumulator layout -> one row per lane)
  double* KapT = Pq + RT * 16 * SS;  // [16]      Quu^{-T} Qu (cost matrices that are not symmetric: Vx' = Qx - Qux^T Quu^{-T} Qu, :666)
  constexpr int kExch = 16 * QS + 16 * WSS + 16 + FS + RT * 16 * SS + 16;
  static_assert(kExch <= Ly::NMP * Ly::TS, "the exchange buffers live in the H area");
  const d4_t zero4 = {0.0, 0.0, 0.0, 0.0};
  auto wave_lds_fence = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
  };

  // terminal: Vx = 2 Qf x_T - 2 x_nom^T Qf ; Vxx = 2 Qf   (ilqr.py:203-204, :638); pads = 0
  if (tid == 0) lds[Ly::oRed + kPdFlag] = 0.0;
  for (int e = tid; e < NP * VS; e += kLargeThreads) {
    const int i = e / VS, j = e - i * VS;
    Vxx[e] = (i < n && j < n) ? 2.0 * Qf[i * n + j] : 0.0;
  }
  if (tid >= n && tid < NK) Vx[tid] = 0.0;
  if (tid < n) {
    const double* xT = v.X + (size_t)(N - 1) * n;
    double s = 0.0;
    for (int j = 0; j < n; ++j) s += (2.0 * Qf[tid * n + j]) * xT[j];
    Vx[tid] = s - qfn[tid];
  }
  // cost gradients for all steps (ilqr.py:180-181), unless the accepted trial's rollout left them (large_rollout)
  // (a thread keeps ITS row of 2Q / 2R in registers and walks the time steps: kLargeThreads / (n + m) of them in flight;
  //  xu_staged: the linearization's LDS copy of x_bar / u_bar - rows of n / m doubles in the T1 / F areas - is still there)
  if (!lx_ready) {
    const double* const Xs_ = xu_staged ? lds + Ly::oT1 : v.X;
    const double* const Us_ = xu_staged ? lds + Ly::oF : v.U;
    constexpr int TG = kLargeThreads / nm;
    const int pp = tid % nm, g0 = tid / nm;
    if (g0 < TG) {
      if (pp < n) {
        double q2[n];
#pragma unroll
        for (int j = 0; j < n; ++j) q2[j] = 2.0 * Q[pp * n + j];
        const double qnp = qn[pp];
        for (int tt = g0; tt < N - 1; tt += TG) {
          const double* xg = Xs_ + (size_t)tt * n;
          double s_ = -qnp;
#pragma unroll
          for (int j = 0; j < n; ++j) s_ += q2[j] * xg[j];
          lxu_store(v, Lxu, tt * nm + pp, s_);
        }
      } else {
        double r2_[m];
#pragma unroll
        for (int j = 0; j < m; ++j) r2_[j] = 2.0 * R[(pp - n) * m + j];
        for (int tt = g0; tt < N - 1; tt += TG) {
          const double* ug = Us_ + (size_t)tt * m;
          double s_ = 0.0;
#pragma unroll
          for (int j = 0; j < m; ++j) s_ += r2_[j] * ug[j];
          lxu_store(v, Lxu, tt * nm + pp, s_);
        }
      }
    }
  }
  __syncthreads();                                           // (the rollout's scratch inside the T1 area is dead now)
  for (int e = tid; e < NK * FS; e += kLargeThreads) { F[e] = 0.0; F[FB1 + e] = 0.0; }
  for (int e = tid; e < kExch; e += kLargeThreads) H[e] = 0.0;
  __syncthreads();

  // F's pipeline (the pipeline wave): a whole F_t = [fx_t | fu_t] (contiguous n*n and n*m blocks in HBM) goes to registers
  // three steps ahead and to the LDS buffer that F_{t+2} leaves two steps ahead.
  constexpr int W = (n % 2 == 0 && m % 2 == 0 && FS % 2 == 0 && Ly::oF % 2 == 0 && Ly::oT1 % 2 == 0 && UC % 2 == 0) ? 2 : 1;
  constexpr int PFX = n * n / W, PFU = n * m / W;
  constexpr int NFX = (PFX + 63) / 64, NFU = (PFU + 63) / 64;
  typedef double d2_t __attribute__((ext_vector_type(2)));
  using fw_t = std::conditional_t<W == 2, d2_t, double>;
  auto pipeline_role = [&]() __attribute__((always_inline)) {
    fw_t frx[NFX], fru[NFU];
    int fx_off[NFX], fu_off[NFU];
#pragma unroll
    for (int r = 0; r < NFX; ++r) { int e = W * (lane + 64 * r); e = e < n * n ? e : n * n - W; fx_off[r] = (e / n) * FS + (e % n); }
#pragma unroll
    for (int r = 0; r < NFU; ++r) { int e = W * (lane + 64 * r); e = e < n * m ? e : n * m - W; fu_off[r] = (e / m) * FS + UC + (e % m); }
    auto fetch = [&](int t) __attribute__((always_inline)) {
      const fw_t* fxg = reinterpret_cast<const fw_t*>(v.Fx + (size_t)t * n * n);
      const fw_t* fug = reinterpret_cast<const fw_t*>(v.Fu + (size_t)t * n * m);
#pragma unroll
      for (int r = 0; r < NFX; ++r) { const int pi = lane + 64 * r; frx[r] = fxg[pi < PFX ? pi : PFX - 1]; }
#pragma unroll
      for (int r = 0; r < NFU; ++r) { const int pi = lane + 64 * r; fru[r] = fug[pi < PFU ? pi : PFU - 1]; }
    };
    auto publish = [&](double* Fb) __attribute__((always_inline)) {     // clamped duplicates rewrite the last element with itself
#pragma unroll
      for (int r = 0; r < NFX; ++r) *reinterpret_cast<fw_t*>(Fb + fx_off[r]) = frx[r];
#pragma unroll
      for (int r = 0; r < NFU; ++r) *reinterpret_cast<fw_t*>(Fb + fu_off[r]) = fru[r];
    };
    fetch(N - 2); publish(F);
    if (N >= 3) { fetch(N - 3); publish(F + FB1); }
    if (N >= 4) fetch(N - 4);
    __syncthreads();                                         // (A) F_{N-2}, F_{N-3} in LDS
    __syncthreads();                                         // (B)
    for (int t = N - 2; t >= 0; --t) {
      lds_barrier();
      if (t >= 2) {
        publish(F + ((N - 2 - t) & 1) * FB1);                // F_{t-2} replaces F_t, which nobody reads any more
        if (t >= 3) fetch(t - 3);
      }
      lds_barrier();
    }
  };

  // x-wave W_: column tile W_ of F (columns 16 W_ .. of x)
  auto x_role = [&](auto wc) __attribute__((always_inline)) {
    constexpr int W_ = decltype(wc)::value;
    const int col = 16 * W_ + lr;
    const bool col_ok = col < n;
    double q2[RT][4];                                        // lxx = 2Q entries of this lane's Vxx' results
#pragma unroll
    for (int q = 0; q < RT; ++q)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int row = 16 * q + 4 * reg + lk;
        q2[q][reg] = (row < n && col_ok) ? 2.0 * Q[row * n + col] : 0.0;
      }
    // This wave's share of the next Quu - luu = fu^T Vxx fu, from the column tile of Vxx it holds in the accumulator layout:
    // read as an A operand that tile is the ROW tile of Vxx^T (large_backward: "Fused chain"), so
    //   T1u[16W + r][a] = sum_k Vxx^T[16W + r][k] fu[k][a],   P_W = fu[16W.., :]^T T1u[16W.., :],   sum_W P_W = (fu^T Vxx fu)^T -
    // the transpose of a matrix that is symmetric up to round-off.  Off the u-wave's critical path: it only adds the shares.
    auto quu_share = [&](const d4_t (&vc)[RT], const double* Fb) __attribute__((always_inline)) {
      double fun[KN];
      const double* ub = Fb + lk * FS + UC + lr;             // B[k][c] = fu[k][c]; also A = fu^T: A[p][k] = fu[k][p]
#pragma unroll
      for (int ks = 0; ks < KN; ++ks) fun[ks] = ub[ks * 4 * FS];
      d4_t tu = zero4;
#pragma unroll
      for (int ks = 0; ks < KN; ++ks) tu = __builtin_amdgcn_mfma_f64_16x16x4f64(vc[ks >> 2][ks & 3], fun[ks], tu, 0, 0, 0);
      d4_t pw = zero4;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
        if (4 * W_ + reg < KN) pw = __builtin_amdgcn_mfma_f64_16x16x4f64(fun[4 * W_ + reg], tu[reg], pw, 0, 0, 0);
      double* pd = Pq + W_ * 16 * SS + lk * SS + lr;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) pd[4 * reg * SS] = pw[reg];
    };
    __syncthreads();                                         // (A)
    {
      d4_t vterm[RT];                                        // the terminal Vxx = 2 Qf: this wave's column tile
#pragma unroll
      for (int q = 0; q < RT; ++q)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) vterm[q][reg] = col_ok ? Vxx[(16 * q + 4 * reg + lk) * VS + col] : 0.0;
      quu_share(vterm, F);
    }
    __syncthreads();                                         // (B)
    for (int t = N - 2; t >= 0; --t) {
      const double* Fc = F + ((N - 2 - t) & 1) * FB1;         // F_t
      const double* Fn = F + ((N - 1 - t) & 1) * FB1;         // F_{t-1}, published a step ago
      // ---- T1[:, W] = Vxx F[:, W]
      double fb[KN];
      const double* b_base = Fc + lk * FS + col;
#pragma unroll
      for (int ks = 0; ks < KN; ++ks) fb[ks] = b_base[ks * 4 * FS];
      d4_t accA[RT];
#pragma unroll
      for (int q = 0; q < RT; ++q) {
        const double* ap = Vxx + (16 * q + lr) * VS + lk;
        double va[KN];
#pragma unroll
        for (int ks = 0; ks < KN; ++ks) va[ks] = ap[4 * ks];
        d4_t acc = zero4;
#pragma unroll
        for (int ks = 0; ks < KN; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(va[ks], fb[ks], acc, 0, 0, 0);
        accA[q] = acc;
      }
      // ---- H[:, W] = F^T T1[:, W]: the x row tiles (Qxx - lxx) and u's row tile (Qux)
      d4_t hx[RT], hu = zero4;
#pragma unroll
      for (int q = 0; q < RT; ++q) {
        const double* ap = Fc + lk * FS + 16 * q + lr;       // A = F^T: A[p][k] = F[k][16 q + p]
        double fa[KN];
#pragma unroll
        for (int ks = 0; ks < KN; ++ks) fa[ks] = ap[ks * 4 * FS];
        d4_t acc = zero4;
#pragma unroll
        for (int ks = 0; ks < KN; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[ks], accA[ks >> 2][ks & 3], acc, 0, 0, 0);
        hx[q] = acc;
      }
      {
        const double* ap = Fc + lk * FS + UC + lr;
        double fa[KN];
#pragma unroll
        for (int ks = 0; ks < KN; ++ks) fa[ks] = ap[ks * 4 * FS];
#pragma unroll
        for (int ks = 0; ks < KN; ++ks) hu = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[ks], accA[ks >> 2][ks & 3], hu, 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < MK; ++j) QuxS[(4 * j + lk) * QS + col] = hu[j];
      lds_barrier();
      // ---- K[:, W] = Quu^{-1} Qux[:, W] (:660)
      double wa[MK];
#pragma unroll
      for (int j = 0; j < MK; ++j) {                         // symmetric costs: W's upper triangle, mirrored (large_backward, same place)
        const int k_ = 4 * j + lk;
        wa[j] = Ws[(v.asym || lr <= k_) ? lr * WSS + k_ : k_ * WSS + lr];
      }
      d4_t kt = zero4;
#pragma unroll
      for (int j = 0; j < MK; ++j) kt = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[j], hu[j], kt, 0, 0, 0);
      if (col_ok) {
        double* Kg = v.K + (size_t)t * m * n + col;          // K_t[4 j + lk][col]
#pragma unroll
        for (int j = 0; j < MK; ++j) if (4 * j + lk < m) Kg[(4 * j + lk) * n] = kt[j];
      }
      // ---- Vxx'[:, W] = Qxx[:, W] - Qux^T K[:, W] (:667)
      d4_t vnew[RT];
#pragma unroll
      for (int q = 0; q < RT; ++q) {
        double qa[MK];
#pragma unroll
        for (int j = 0; j < MK; ++j) qa[j] = -QuxS[(4 * j + lk) * QS + 16 * q + lr];
        d4_t vq;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) vq[reg] = hx[q][reg] + q2[q][reg];
#pragma unroll
        for (int j = 0; j < MK; ++j) vq = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[j], kt[j], vq, 0, 0, 0);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int row = 16 * q + 4 * reg + lk;
          const bool ok = row < n && col_ok;
          if (ok) Vxx[row * VS + col] = vq[reg];
          vq[reg] = ok ? vq[reg] : 0.0;                      // (pad rows / columns of Vxx are exact zeros)
        }
        vnew[q] = vq;
      }
      if (t > 0) quu_share(vnew, Fn);                        // this wave's share of the NEXT step's Quu
      lds_barrier();
    }
  };

  // u-wave: Quu and its inverse, the first-order terms
  auto u_role = [&]() __attribute__((always_inline)) {
    // (the gradient is read HERE, not at the top of the step as in large_backward: measured on the mid-size kernels the early
    //  read costs the u-wave's elimination more than it saves - (12, 4): 2.46 k -> 2.59 k cycles per step)
    auto first_order = [&](int ts, const double* Fb) __attribute__((always_inline)) {   // l_{x,u} + F^T Vx (:651-652)
      if (lane < nm) {
        double s = lxu_load(v, Lxu, ts * nm + lane);
        const int hp = lane < n ? lane : UC + (lane - n);    // this entry's column of F
#pragma unroll
        for (int k = 0; k < NK; ++k) s += Fb[k * FS + hp] * Vx[k];
        Fo[hp] = s;
      }
    };
    const int si = lr < m ? lr : m - 1;                      // lanes >= m of each 16-lane row shadow the last row
    double r2[m];
#pragma unroll
    for (int j = 0; j < m; ++j) r2[j] = 2.0 * R[si * m + j];
    __syncthreads();                                         // (A)
    first_order(N - 2, F);
    __syncthreads();                                         // (B)
    for (int t = N - 2; t >= 0; --t) {
      const double* Fn = F + ((N - 1 - t) & 1) * FB1;         // F_{t-1}
      // ---- Quu = luu + the x-waves' shares of fu^T Vxx fu (:654)
      double arow[m];
      // (the shares add up to the TRANSPOSE of fu^T Vxx fu - x_role - which only matters when the cost matrices are not symmetric)
      auto load_quu = [&](double (&ar)[m]) __attribute__((always_inline)) {
        if (v.asym) {
          const double* p0 = Pq + si;
#pragma unroll
          for (int j = 0; j < m; ++j) ar[j] = r2[j] + (RT == 2 ? p0[j * SS] + p0[(16 + j) * SS] : p0[j * SS]);
        } else {
          const double* p0 = Pq + si * SS;
#pragma unroll
          for (int j = 0; j < m; ++j) ar[j] = r2[j] + (RT == 2 ? p0[j] + p0[16 * SS + j] : p0[j]);
        }
      };
      load_quu(arow);
      double sc = 1.0;
      GjOuter<m, 0>::run(arow, sc, si);                      // Quu^{-1}[si][j] = sc * arow[j] (:655)
      if (!gj_row_positive(sc)) lds[Ly::oRed + kPdFlag] = 1.0;   // Quu not positive definite (read by the kernel after the pass: MI_STATUS_NOT_PD)
      // cold path: the reference's inverse (LU with partial pivoting) of an indefinite Quu - and of EVERY Quu when the cost matrices
      // are not symmetric: positive pivots say nothing about the growth of an unpivoted elimination of a matrix that is not
      // symmetric (measured: 5e-8 / 4e-5 from the oracle on the (27, 7) plugin / the arm at N = 24 next to steps that tripped the check)
      if (PIV && (v.asym || (v.pd_continue && __any(!gj_row_positive(sc))))) {
        load_quu(arow);
        if (lane < m) {
#pragma unroll
          for (int j = 0; j < m; ++j) Ws[lane * WSS + j] = arow[j];
        }
        quu_inverse_pivoted<m>(Ws, WSS, lane);
#pragma unroll
        for (int j = 0; j < m; ++j) arow[j] = Ws[si * WSS + j];
        sc = 1.0;
      }
      if (lane < m) {
#pragma unroll
        for (int j = 0; j < m; ++j) Ws[lane * WSS + j] = sc * arow[j];
      }
      // ---- kappa = Quu^{-1} Qu (:659), dV = Qu^T kappa (:663): Qu is the first-order column this wave formed a step ago
      {
        double kp = 0.0;
#pragma unroll
        for (int j = 0; j < m; ++j) kp = fma(arow[j], Fo[UC + j], kp);
        kp *= sc;
        const double dv = row16_sum(lr < m ? Fo[UC + si] * kp : 0.0);
        if (lane < m) { Kap[lane] = kp; v.kap[(size_t)t * m + lane] = kp; }
        if (lane == 0) v.dV[t] = dv;
      }
      if (v.asym) {
        // Vx' = Qx - Qu^T Quu^{-1} Qux (:666) = Qx - Qux^T (Quu^{-T} Qu): with a Quu that is not symmetric that is NOT Qux^T kappa
        wave_lds_fence();
        if (lane < m) {
          double kt_ = 0.0;
#pragma unroll
          for (int i_ = 0; i_ < m; ++i_) kt_ = fma(Ws[i_ * WSS + lane], Fo[UC + i_], kt_);
          KapT[lane] = kt_;
        }
      }
      lds_barrier();
      // ---- Vx' = Qx - Qux^T kappa (:666)
      if (lane < n) {
        const double* const kv_ = v.asym ? KapT : Kap;
        double s = Fo[lane];
#pragma unroll
        for (int a_ = 0; a_ < m; ++a_) s -= QuxS[a_ * QS + lane] * kv_[a_];
        Vx[lane] = s;
      }
      if (t > 0) {
        wave_lds_fence();
        first_order(t - 1, Fn);                              // the next step's, from the Vx' just formed
      }
      lds_barrier();
    }
  };

  if (wave == WU) u_role();
  else if (wave == WP) pipeline_role();
  else if (wave == 0) x_role(std::integral_constant<int, 0>{});
  else {
    if constexpr (RT == 2) {
      x_role(std::integral_constant<int, 1>{});
    } else {                                                 // (n <= 16: the fourth wave only keeps the barriers' count)
      __syncthreads();
      __syncthreads();
      for (int t = N - 2; t >= 0; --t) { lds_barrier(); lds_barrier(); }
    }
  }
}

// The backward pass for cost matrices that are NOT symmetric, 32 < n <= 40 (round 6).  The reference takes any Q, R, Qf and never
// symmetrizes anything (ilqr.py:130-146, 180-184, 651-667); large_backward cannot follow it there - its fused chain keeps a wave's
// column tile of Vxx as that wave's ROW tile too and mirrors three tiles of Qxx, i.e. it computes with Vxx = Vxx^T - and the
// library refused such matrices for these sizes until round 6.  This is the recursion as the reference writes it, with no use of
// symmetry anywhere, in plain fp64 multiply-adds out of LDS:
//     A = F^T Vxx ((n+m) x n), F = [fx | fu];   Qx | Qu = lx | lu + F^T Vx;   [Qxx - lxx ; Qux] = A fx;   Quu = luu + A_u fu;
//     Quu^{-1} by LU with partial pivoting (np.linalg.inv's algorithm; positive pivots of an unpivoted elimination say nothing
//     about a matrix that is not symmetric - they only feed the MI_STATUS_NOT_PD / FLAG_INDEFINITE report, as in mid_backward);
//     kappa = Quu^{-1} Qu, K = Quu^{-1} Qux, dV = Qu^T kappa, Vx' = Qx - Qux^T (Quu^{-T} Qu), Vxx' = Qxx - Qux^T K   (:659-667)
// lx = 2 Q x - 2 x_nom^T Q, lu = 2 R u are formed here, step by step (a rollout's cost rows hold 2 Q (x - x_nom): not lx when Q is
// not symmetric).  A cold path: ~4 x the cycles of the matrix-core pass per step (LDS-bound dot products, six barriers), taken
// only by handles whose cost matrices are not symmetric - the kernels of every other handle never enter it.
template <class M>
__device__ inline void large_backward_asym(const LView<M::n, M::m>& v, double* lds) {
  constexpr int n = M::n, m = M::m, nm = n + m;
  using Ly = LLay<n, m>;
  static_assert(!Ly::kMid && m <= 16, "32 < n <= 40, m <= 16");
  constexpr int WS = 17;
  const int tid = stage_tid(), N = v.N, lane = tid & 63, wave = tid >> 6;
  const double* Q = lds + Ly::oQ;
  const double* R = lds + Ly::oR;
  const double* Qf = lds + Ly::oQf;
  const double* qn = lds + Ly::oQn;
  const double* qfn = lds + Ly::oQfn;
  double* Vxx = lds + Ly::oVxx;      // [n][n] dense (the matrix-core layout's area: NP x VS)
  double* Vx = lds + Ly::oVx;        // [n]
  double* Fb = lds + Ly::oF;         // [n][nm] = [fx_t | fu_t]; after the products: scratch of the pivot check [m][WS]
  double* A = lds + Ly::oT1;         // [nm][n] = F^T Vxx; after the products: K_t [m][n]
  double* Hb = lds + Ly::oH;         // [nm][n]: rows < n  fx^T Vxx fx, rows >= n  Qux
  double* Gr = Hb + nm * n;          // [nm] lx_t | lu_t
  double* q1 = Gr + nm;              // [nm] Qx | Qu
  double* kap = q1 + nm;             // [16] kappa_t
  double* kapT = kap + 16;           // [16] Quu^{-T} Qu
  double* W = lds + Ly::oS;          // [m][WS] Quu, then its inverse
  static_assert(n * n <= Ly::NP * Ly::VS && n * nm <= Ly::NK * Ly::NMP && nm * n <= Ly::NK * Ly::TS, "dense operands inside the matrix-core layout's areas");
  static_assert(nm * n + 2 * nm + 32 <= Ly::NMP * Ly::TS && m * WS <= 16 * 17 + 1 && m * WS <= n * nm, "products and small vectors inside the H area, Quu in the tile scratch");
  auto wave_fence = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
  };
  // terminal: Vx = 2 Qf x_T - 2 x_nom^T Qf ; Vxx = 2 Qf   (ilqr.py:203-204, :638)
  __syncthreads();
  if (tid == 0) lds[Ly::oRed + kPdFlag] = 0.0;
  for (int e = tid; e < n * n; e += kLargeThreads) Vxx[e] = 2.0 * Qf[e];
  if (tid < n) {
    const double* xT = v.X + (size_t)(N - 1) * n;
    double s_ = 0.0;
    for (int j = 0; j < n; ++j) s_ += (2.0 * Qf[tid * n + j]) * xT[j];
    Vx[tid] = s_ - qfn[tid];
  }
  __syncthreads();
#pragma unroll 1
  for (int t = N - 2; t >= 0; --t) {
    // ---- F_t and the cost gradients (:180-181)
    {
      const double* fxg = v.Fx + (size_t)t * n * n;
      const double* fug = v.Fu + (size_t)t * n * m;
      for (int e = tid; e < n * nm; e += kLargeThreads) {
        const int i = e / nm, j = e - i * nm;
        Fb[e] = j < n ? fxg[i * n + j] : fug[i * m + (j - n)];
      }
      if (tid < n) {
        const double* xg = v.X + (size_t)t * n;
        double s_ = -qn[tid];
        for (int j = 0; j < n; ++j) s_ += (2.0 * Q[tid * n + j]) * xg[j];
        Gr[tid] = s_;
      } else if (tid < nm) {
        const double* ug = v.U + (size_t)t * m;
        double s_ = 0.0;
        for (int j = 0; j < m; ++j) s_ += (2.0 * R[(tid - n) * m + j]) * ug[j];
        Gr[tid] = s_;
      }
    }
    __syncthreads();
    // ---- A = F^T Vxx ; Qx | Qu = l + F^T Vx (:651-652)
    for (int e = tid; e < nm * n; e += kLargeThreads) {
      const int r = e / n, j = e - r * n;
      double s_ = 0.0;
#pragma unroll 4
      for (int i = 0; i < n; ++i) s_ = fma(Fb[i * nm + r], Vxx[i * n + j], s_);
      A[e] = s_;
    }
    if (tid >= 192 && tid - 192 < nm) {
      const int r = tid - 192;
      double s_ = Gr[r];
      for (int i = 0; i < n; ++i) s_ = fma(Fb[i * nm + r], Vx[i], s_);
      q1[r] = s_;
    }
    __syncthreads();
    // ---- [Qxx - lxx ; Qux] = A fx ; Quu = luu + A_u fu (:653-656)
    for (int e = tid; e < nm * n; e += kLargeThreads) {
      const int r = e / n, j = e - r * n;
      double s_ = 0.0;
#pragma unroll 4
      for (int k = 0; k < n; ++k) s_ = fma(A[r * n + k], Fb[k * nm + j], s_);
      Hb[e] = s_;
    }
    for (int e = tid; e < m * m; e += kLargeThreads) {
      const int a_ = e / m, b_ = e - a_ * m;
      double s_ = 0.0;
      for (int k = 0; k < n; ++k) s_ = fma(A[(n + a_) * n + k], Fb[k * nm + n + b_], s_);
      W[a_ * WS + b_] = 2.0 * R[a_ * m + b_] + s_;
    }
    __syncthreads();
    // ---- Quu^{-1} (:655) on one wave; the pivots of an UNPIVOTED elimination of a copy only feed the status report
    if (wave == 0) {
      double* S = Fb;
      const bool on = lane < m;
      const int i = on ? lane : 0;
      if (on) { for (int j = 0; j < m; ++j) S[i * WS + j] = W[i * WS + j]; }
      bool bad = false;
#pragma unroll 1
      for (int k = 0; k < m; ++k) {
        wave_fence();
        const double piv = S[k * WS + k];
        bad = bad || !(piv > 0.0);
        double l_ = 0.0;
        if (on && i > k) l_ = S[i * WS + k] / piv;
        wave_fence();
        if (on && i > k) { for (int j = k + 1; j < m; ++j) S[i * WS + j] = fma(-l_, S[k * WS + j], S[i * WS + j]); }
      }
      if (bad && lane == 0) lds[Ly::oRed + kPdFlag] = 1.0;
      wave_fence();
      quu_inverse_pivoted<m>(W, WS, lane);
    }
    __syncthreads();
    // ---- kappa = Quu^{-1} Qu, Quu^{-T} Qu, K = Quu^{-1} Qux (:659-660)
    if (tid < m) {
      double s_ = 0.0;
      for (int b_ = 0; b_ < m; ++b_) s_ = fma(W[tid * WS + b_], q1[n + b_], s_);
      kap[tid] = s_;
      v.kap[(size_t)t * m + tid] = s_;
    } else if (tid >= 64 && tid - 64 < m) {
      const int a_ = tid - 64;
      double s_ = 0.0;
      for (int b_ = 0; b_ < m; ++b_) s_ = fma(W[b_ * WS + a_], q1[n + b_], s_);
      kapT[a_] = s_;
    }
    {
      double* Kg = v.K + (size_t)t * m * n;
      for (int e = tid; e < m * n; e += kLargeThreads) {
        const int a_ = e / n, j = e - a_ * n;
        double s_ = 0.0;
        for (int b_ = 0; b_ < m; ++b_) s_ = fma(W[a_ * WS + b_], Hb[(n + b_) * n + j], s_);
        A[e] = s_;
        Kg[e] = s_;
      }
    }
    __syncthreads();
    // ---- dV = Qu^T kappa (:663), Vx' = Qx - Qux^T Quu^{-T} Qu, Vxx' = Qxx - Qux^T K (:666-667)
    if (tid == 0) {
      double s_ = 0.0;
      for (int a_ = 0; a_ < m; ++a_) s_ = fma(q1[n + a_], kap[a_], s_);
      v.dV[t] = s_;
    }
    if (tid >= 64 && tid - 64 < n) {
      const int j = tid - 64;
      double s_ = q1[j];
      for (int b_ = 0; b_ < m; ++b_) s_ -= Hb[(n + b_) * n + j] * kapT[b_];
      Vx[j] = s_;
    }
    for (int e = tid; e < n * n; e += kLargeThreads) {
      const int i = e / n, j = e - i * n;
      double s_ = 0.0;
      for (int b_ = 0; b_ < m; ++b_) s_ = fma(Hb[(n + b_) * n + i], A[b_ * n + j], s_);
      Vxx[e] = (2.0 * Q[e] + Hb[e]) - s_;
    }
    __syncthreads();
  }
}

// The backward pass of a model's size class.
template <class M, bool PIV>
__device__ __forceinline__ void backward_pass(const LView<M::n, M::m>& v, double* lds, long long* bp_acc, bool lx_ready, bool xu_staged = false) {
  if constexpr (LLay<M::n, M::m>::kMid) mid_backward<M, PIV>(v, lds, lx_ready, xu_staged);
  else {
    if constexpr (PIV) {
      if (v.asym) { large_backward_asym<M>(v, lds); return; }   // (cost matrices that are not symmetric: launched as the PIV form, launch_large.hpp)
    }
    large_backward<M, PIV>(v, lds, bp_acc, lx_ready);
  }
}

#ifndef MI_MID_MINBLOCKS
#define MI_MID_MINBLOCKS 1
#endif
template <class M>
constexpr int kMinBlocks = LLay<M::n, M::m>::kMid ? MI_MID_MINBLOCKS : 1;
// PIV: the backward passes carry the pivoted-inverse cold path (quu_inverse_pivoted) - launched for on_indefinite = 1 and for cost
// matrices that are not symmetric; modes without a backward pass exist as PIV = false only.
template <class M, int JAC, int MODE, bool PIV = false>
__global__ void __launch_bounds__(kLargeThreads, kMinBlocks<M>) ilqr_large_kernel(const KArgs a) {
  constexpr int n = M::n, m = M::m;
  using Ly = LLay<n, m>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* lds = reinterpret_cast<double*>(smem);
  int* ilds = reinterpret_cast<int*>(lds + Ly::doubles + (a.lxu ? (size_t)0 : (size_t)a.N * (n + m)));
  // `cluster` workgroups per problem (MODE_SOLVE / MODE_MPC): workgroup 0 of a cluster is the leader and runs the
  // solve, the others only help with its linearizations (cluster handshake below)
  const int G = ((MODE == MODE_SOLVE || MODE == MODE_MPC) && (a.cluster & 0xff) > 1) ? (a.cluster & 0xff) : 1;
  const int corder = (a.cluster >> 8) & 3;                  // placement of a cluster's members (mi_ilqr.hip: launch arguments)
  const bool early_lin = ((a.cluster >> 10) & 1) != 0;      // early linearization (below)
  const bool ls_groups = ((a.cluster >> 11) & 1) != 0;      // candidate groups on the helper workgroups (mid_linesearch4)
  constexpr int kEarlyBlock = (kLargeThreads / (n + m)) > 0 ? kLargeThreads / (n + m) : 1;   // steps per block: one pass of the workgroup
  const int early_blocks = (a.N - 1 + kEarlyBlock - 1) / kEarlyBlock;                          // blocks of an early round ...
  const int early_helper_blocks = early_blocks - (EarlyLeaderBlocks<M>::value < early_blocks ? EarlyLeaderBlocks<M>::value : early_blocks - 1);   // ... the first of them the helpers'
  // XCD-aware placement of a cluster: the dispatcher deals workgroups to the 8 XCDs round-robin by blockIdx.x, so the members of
  // one cluster take block indices that are congruent mod 8 - one XCD, ONE L2: the helpers' Jacobians reach the leader through the
  // cache they share, and the leaders spread over all eight L2s (with b = blockIdx.x / G the leaders of clusters of 4 all sat on
  // XCDs 0 and 4).  The grid is 8 G ceil(B / 8) workgroups (launch_large.hpp); those beyond the batch leave at once.  Nothing
  // below RELIES on the placement: every member reads its XCC id from the hardware register and the leader takes the same-L2
  // path of the handshake only when all the helpers it counts on reported its own id.
  const int xslot = (int)blockIdx.x >> 3, xP = (a.B + 7) >> 3;       // slot on its XCD; problems per XCD
  const int tid = threadIdx.x, N = a.N;
  int b = (int)blockIdx.x, role = 0;
  if (G > 1) {
    if (corder == 0) { b = (int)blockIdx.x / G; role = (int)blockIdx.x - b * G; }                     // consecutive blocks: a cluster spans G XCDs
    else if (corder == 1) { b = ((int)blockIdx.x & 7) + 8 * (xslot / G); role = xslot % G; }         // one XCD, members in consecutive slots
    else { b = ((int)blockIdx.x & 7) + 8 * (xslot % xP); role = xslot / xP; }                         // one XCD, the XCD's leaders first, then its helpers
    if (b >= a.B) return;
  }
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
  v.LxG = a.lxu ? a.lxu + (size_t)b * (N - 1) * (n + m) : nullptr;
  v.pd_continue = a.pd_continue;
  v.asym = a.cost_asym;                     // (n >= 33: large_backward_asym, the PIV form of the kernel)
  LargeAcc<n, m> acc;
  acc.X = v.X; acc.Fx = v.Fx; acc.Fu = v.Fu; acc.N = N;
  { const int Nr = int_row(N); acc.kp = ilds; acc.aux = ilds + Nr; acc.need = ilds + 2 * Nr; acc.binA = ilds + 3 * Nr; acc.binB = ilds + 5 * Nr; }
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
  // lazily-zero persistent state / pending initial guess (the cluster's leader only: helpers never write solver state)
  if (a.cold && role == 0) {
    for (int e = tid; e < n * N; e += kLargeThreads) v.X[e] = 0.0;
    for (int e = tid; e < m * n * (N - 1); e += kLargeThreads) v.K[e] = 0.0;
    for (int e = tid; e < m * (N - 1); e += kLargeThreads) v.kap[e] = 0.0;
    for (int e = tid; e < N - 1; e += kLargeThreads) v.dV[e] = 0.0;
    // (a clustered launch: no dirty line of fx / fu may stay in this XCD's L2 - see open_early)
    if (G > 1) {
      for (int e = tid; e < n * n * (N - 1); e += kLargeThreads) st_shared<true>(v.Fx + e, 0.0);
      for (int e = tid; e < n * m * (N - 1); e += kLargeThreads) st_shared<true>(v.Fu + e, 0.0);
    } else {
      for (int e = tid; e < n * n * (N - 1); e += kLargeThreads) v.Fx[e] = 0.0;
      for (int e = tid; e < n * m * (N - 1); e += kLargeThreads) v.Fu[e] = 0.0;
    }
  }
  if (a.u_pending && role == 0) {
    const double* ug = a.u_guess + (size_t)b * m * (N - 1);
    for (int e = tid; e < m * (N - 1); e += kLargeThreads) v.U[e] = ug[e];
  }
  __syncthreads();

  // The sparse Jacobian code reads a handful of x/u entries per evaluation: it takes them from an
  // LDS copy of the nominal trajectory (the backward pass's T1|H and F areas are idle during the
  // linearization) instead of paying an L2 round trip per dependent access.
  const bool lin_staged = (size_t)(n + 1) * N <= (size_t)(Ly::NK + Ly::NMP) * Ly::TS && (size_t)(m + 1) * (N - 1) <= (size_t)Ly::NK * Ly::NMP;
  const double* lin_X = lin_staged ? lds + Ly::oT1 : v.X;
  const double* lin_U = lin_staged ? lds + Ly::oF : v.U;
  // Row strides of the LDS copy.  Chain models deal their items key-point fastest: the lanes of a wave read the
  // same entry of DIFFERENT time steps, and strides of 36 / 12 doubles would put them 16 deep on four bank
  // groups - odd strides spread them over all banks.
  constexpr int kXS = IsChainModel<M>::value ? (n | 1) : n, kUS = IsChainModel<M>::value ? (m | 1) : m;
  const int lin_xs = lin_staged ? kXS : n, lin_us = lin_staged ? kUS : m;
  auto jac = [&](const int* list, int count) __attribute__((always_inline)) {
    if constexpr (HasSparsity<M>::value) large_jac_at_sparse<M, JAC>(v, a, list, count, lin_X, lin_U);
    else if constexpr (IsChainModel<M>::value) {
      // the per-key-point cache lives where the backward pass keeps its cost gradients (idle here): LDS, or - long horizons - HBM
      if (v.LxG) large_jac_at_tree<M, JAC>(v, a, list, count, lin_X, lin_U, lin_xs, lin_us, v.LxG);
      else large_jac_at_tree<M, JAC>(v, a, list, count, lin_X, lin_U, lin_xs, lin_us, lds + Ly::doubles);
    }
    else if constexpr (IsLegModel<M>::value) large_jac_at_legs<M, JAC>(v, a, list, count, lin_X, lin_U, lin_xs, lin_us);
    else large_jac_at<M, JAC>(v, a, list, count, lin_X, lin_U, lin_xs, lin_us);
  };
  // ---- cluster handshake (G > 1; every step a key-point, models with an LDS-staged linearization) -------------
  // The linearization is the one stage of an iteration whose (step, column) items are independent, and with few
  // problems per GPU most CUs idle: the leader of a problem's cluster shares them with its helper workgroups.
  //   sync words (global, zero at launch, kSyncWords per problem): [0] command = round << 32 | early << 16 | participants,
  //   [1] done (shares finished, monotonic), [2] alive (helpers that have started; per XCC id above bit 16), [3] exit,
  //   [4] progress of the trial being rolled out (early linearization: round << 32 | steps out, or | kPubAbort),
  //   [5] early rounds opened << 32 | early rounds whose trial was accepted (written at exit, diagnostic).
  // A helper registers when it starts running (role by arrival); the leader snapshots `alive` when it publishes a
  // round and only counts on those helpers - a workgroup that is not resident yet is never waited for, so the
  // scheme cannot deadlock on an oversubscribed device.  Items are dealt in chunks of 256 to the participants
  // round-robin (static: every Jacobian entry is written by exactly one workgroup, with the arithmetic of the
  // single-workgroup path - bitwise the same fx, fu).  Trajectory and Jacobians cross workgroups through global
  // memory under device-scope release / acquire fences.
  //
  // EARLY LINEARIZATION (round 5).  The first trial of a line search is accepted almost always (the receding-horizon
  // configs: 1.00 - 1.03 trials per iteration), and while the leader rolls it out - N - 1 dependent steps on a handful of lanes -
  // its helpers idle.  So the leader opens an `early` round before the line search: the rollout publishes its progress
  // (large_rollout), the helpers - the leader takes no share - linearize the trial block by block of kEarlyBlock steps as the
  // steps come out, and an accepted trial finds its Jacobians all but finished: what is left of the linearization stage is the
  // wait for the last block.  A rejected first trial is called off (the helpers stop at their next block boundary and report
  // in), and the accepted trial is linearized the usual way.  Same items, same arithmetic: bitwise the same fx, fu.
  unsigned long long* csync = a.cluster_sync + (size_t)kSyncWords * b;
  const bool clustered = G > 1 && lin_staged && a.kp_method == MI_KP_SET_INTERVAL && a.minN == 1;
  // a share of a clustered linearization - the leader's own as well as a helper's: a list of key-points (a block of time steps),
  // agent-scope write-through stores.  (The leader's own share too, although only the leader reads it back: plain stores would
  // leave dirty lines of fx / fu in this XCD's L2, and in a later round - the shares move when a helper arrives late, and an early
  // round's helpers write everything - a helper behind ANOTHER L2 may own those entries.)
  auto jac_list = [&](const int* list, int count, int first, int stride) __attribute__((always_inline)) {
    if constexpr (HasSparsity<M>::value) large_jac_at_sparse<M, JAC, true>(v, a, list, count, lin_X, lin_U, first, stride);
    else if constexpr (IsChainModel<M>::value) large_jac_at_tree<M, JAC, true>(v, a, list, count, lin_X, lin_U, lin_xs, lin_us, lds + Ly::doubles, first, stride);
    else if constexpr (IsLegModel<M>::value) large_jac_at_legs<M, JAC, true>(v, a, list, count, lin_X, lin_U, lin_xs, lin_us, first, stride);
    else large_jac_at<M, JAC, true>(v, a, list, count, lin_X, lin_U, lin_xs, lin_us, first, stride);
  };
#ifndef MI_SPIN_CAP_SHIFT
#define MI_SPIN_CAP_SHIFT 22
#endif
  constexpr long long kSpinCap = 1ll << MI_SPIN_CAP_SHIFT;                 // x s_sleep(4) ~ 1 s: a lost partner ends the wait, not the device
  if (role > 0) {
    // ---- helper workgroup
    if (!clustered) return;
    if (tid == 0) acc.aux[1] = (int)(__hip_atomic_fetch_add(csync + 2, 1ull | (1ull << (16 + 6 * xcc_id())), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xffffull) + 1;
    __syncthreads();
    const int my = acc.aux[1];                              // 1, 2, ...: order of arrival
    // The last round served lives in LDS, not in a register of the loop: this hipcc placed the register's spill copy
    // (v_accvgpr_write_b32) of ilqr_large_kernel<Arm27C, 0, MODE_MPC> at the top of a join block BEFORE the s_or_b64 that
    // re-activates the lanes - executed with EXEC = 0 it saved nothing, the reload returned the round before, and the helpers
    // repeated a finished round for ever.  tools/check_exec_spill.py looks for that pattern in every built kernel (CPU test).
    int* const last_round_p = acc.need;
    if (tid == 0) *last_round_p = 0;
    for (;;) {
      if (tid == 0) {
        unsigned long long cmd = 0;
        long long spins = 0;
        int go = -1;
        const unsigned last_round = (unsigned)*last_round_p;
        for (; spins < 16 * kSpinCap; ++spins) {           // (a helper may be resident long before its leader speaks)
          cmd = __hip_atomic_load(csync + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((unsigned)(cmd >> 32) != last_round) { go = 1; break; }
          if (__hip_atomic_load(csync + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) break;
          __builtin_amdgcn_s_sleep(4);
        }
        acc.aux[0] = go;
        acc.aux[2] = (int)(unsigned)(cmd >> 32);
        acc.aux[3] = (int)(unsigned)(cmd & 0xffffffffull);
        if (go > 0) *last_round_p = (int)(unsigned)(cmd >> 32);
        // a candidate-group round reads the leader's gains, nominal trajectory and x0 with plain loads (mid_rollout4): form (B),
        // the reader's side - this thread has seen the flag; the barrier below hands the acquire to the workgroup
        if (go > 0 && ((cmd >> 18) & 1ull) != 0ull) cluster_acquire();
      }
      __syncthreads();
      const int go = acc.aux[0], parts = acc.aux[3] & 0xffff;
      const bool early = ((acc.aux[3] >> 16) & 1) != 0;
      const unsigned last_round = (unsigned)acc.aux[2];     // (the round being served: the tag of its progress word)
      __syncthreads();
      if (go < 0) return;                                   // exit flag (or nobody spoke for a second)
      if (my >= parts) continue;                            // arrived after this round's snapshot: not counted on
      if constexpr (kSpecRollout<M>) {
        if (((acc.aux[3] >> 18) & 1) != 0) {
          // a candidate-group round (acquired above): roll out the candidates 4 my .. 4 my + 3 of the pass
          double e4[kSpec], L4[kSpec], dvs;
          double e = 1.0;
          for (int q = 0; q < kSpec * my; ++q) e *= a.beta;              // (the sequence eps *= beta produces, ilqr.py:335)
          e4[0] = e;
#pragma unroll
          for (int c = 1; c < kSpec; ++c) e4[c] = e4[c - 1] * a.beta;
          const size_t sxh = (size_t)a.B * n * N, suh = (size_t)a.B * m * (N - 1);
          double* const xh = a.x_spec + (size_t)b * n * N + (size_t)(kSpec * my - 1) * sxh;     // slot of candidate 4 my; the next three follow
          double* const uh = a.u_spec + (size_t)b * m * (N - 1) + (size_t)(kSpec * my - 1) * suh;
          LView<n, m> vh = v;
          vh.Xn = xh; vh.Un = uh;
          mid_rollout4<M>(vh, xh + sxh, uh + suh, sxh, suh, lds, a, x0g, e4, L4, dvs);
          if (tid == 0) {
#pragma unroll
            for (int c = 0; c < kSpec; ++c)
              __hip_atomic_store(csync + 8 + kSpec * (my - 1) + c, (unsigned long long)__double_as_longlong(L4[c]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          cluster_publish_barrier();                        // form (B): the trial trajectories are mid_rollout4's plain stores
          if (tid == 0) { cluster_release(); __hip_atomic_fetch_add(csync + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
          continue;
        }
      }
      // a regular round is ONE block: every key-point, this workgroup's chunks of it; an early round: the blocks of kEarlyBlock
      // steps dealt to the parts - 1 helpers in turn, each linearized whole once the rollout has put its steps out
      const int nblk = early ? early_helper_blocks : 1;
      for (int j = early ? my - 1 : 0; j < nblk; j += early ? parts - 1 : 1) {
        const int t0 = early ? j * kEarlyBlock : 0, t1 = early ? (t0 + kEarlyBlock < N - 1 ? t0 + kEarlyBlock : N - 1) : N - 1;
        if (early) {
          if (tid == 0) {
            int st = -1;                                    // -1: called off (or the leader is gone); 1: steps t0 .. t1-1 are out
            for (long long spins = 0; spins < 16 * kSpinCap; ++spins) {
              const unsigned long long pw = __hip_atomic_load(csync + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if ((unsigned)(pw >> 32) == last_round) {
                if (pw & kPubAbort) break;
                if ((int)(pw & 0x7fffffffull) >= t1) { st = 1; break; }
              }
              if (__hip_atomic_load(csync + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) break;
              __builtin_amdgcn_s_sleep(2);
            }
            acc.aux[0] = st;
          }
          __syncthreads();
          const int st = acc.aux[0];
          __syncthreads();
          if (st < 0) break;
        }
        {                                                   // the leader's trajectory -> LDS: form (A), agent-scope loads of its agent-scope stores
          const double* Xg = early ? v.Xn : v.X;            // (early: the trial as the rollout publishes it, large_rollout)
          const double* Ug = early ? v.Un : v.U;
          double* xs_ = lds + Ly::oT1;
          double* us_ = lds + Ly::oF;
          const int xe = early ? t1 * n : n * N;            // (a regular round stages x_{N-1} too, as it always has)
          for (int e = t0 * n + tid; e < xe; e += kLargeThreads) xs_[(e / n) * kXS + e % n] = ld_shared(Xg + e);
          for (int e = t0 * m + tid; e < t1 * m; e += kLargeThreads) us_[(e / m) * kUS + e % m] = ld_shared(Ug + e);
          for (int i = t0 + tid; i < t1; i += kLargeThreads) acc.kp[i] = i;      // keypoints_set_interval(minN = 1)
          __syncthreads();
        }
        jac_list(acc.kp + t0, t1 - t0, early ? 0 : my, early ? 1 : parts);       // this share of fx / fu: write-through stores
        __syncthreads();                                    // (the next block's staging reuses what this one read)
      }
      cluster_publish_barrier();                            // form (A): this share of fx / fu is agent-scope stores, complete now
      if (tid == 0) __hip_atomic_fetch_add(csync + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // The leader's handshake state lives in LDS (the integer scratch of the key-point methods, idle when every step is a key-point),
  // thread 0 only: [0] rounds, [1] rounds on the same-L2 path, [2] early rounds opened, [3] ... accepted, [4] done-counter value
  // after the helpers of all rounds so far.  (As loop-carried registers of the solve loop they were what tipped the coupled arm's
  // receding-horizon kernel into scratch spills.)
  int* const cst = acc.need;
  if (G > 1 && tid == 0) { cst[0] = 0; cst[1] = 0; cst[2] = 0; cst[3] = 0; cst[4] = 0; cst[5] = 0; }   // ([5]: candidate-group rounds)
  bool groups_round = false;                                 // the round close_early is closing is a candidate-group round
  __syncthreads();
  // leader: one clustered linearization of the committed trajectory (LDS copy in place).  Returns false on a lost helper.
  // (own_only: just the key-points own_t0 .. N-2, every item of them, no handshake - the tail of an accepted early round)
  auto linearize_clustered = [&](bool own_only, int own_t0) __attribute__((always_inline)) -> bool {
    for (int i = tid; i < N - 1; i += kLargeThreads) acc.kp[i] = i;
    int parts = 1;
    if (!own_only) {
      if (tid == 0) {
        const unsigned long long word = __hip_atomic_load(csync + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long alive = word & 0xffffull; // (at most G - 1 <= 7 helpers exist; bits 16 + 6 x ..: how many of them per XCC id)
        const unsigned parts_ = 1u + (unsigned)(alive < (unsigned long long)(G - 1) ? alive : (unsigned long long)(G - 1));
        acc.aux[3] = (int)parts_;
        acc.aux[1] = ((word >> (16 + 6 * xcc_id())) & 63ull) == alive ? 1 : 0;   // every helper counted on shares this workgroup's L2
      }
      __syncthreads();
      parts = acc.aux[3];
      cluster_publish_barrier();                            // form (A): the agent-scope x_bar / u_bar stores of the commit are complete
      if (tid == 0) {
        const unsigned cl_round = (unsigned)(cst[0] += 1);
        __hip_atomic_store(csync + 0, ((unsigned long long)cl_round << 32) | ((unsigned long long)(acc.aux[1] != 0 ? 1 : 0) << 17) | (unsigned)parts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      __syncthreads();
    }
    const int t_first = own_only ? own_t0 : 0;
    jac_list(acc.kp + t_first, N - 1 - t_first, 0, parts);
    cluster_publish_barrier();                              // (this workgroup's own share is complete before anybody's acquire)
    if (own_only) return true;                              // (the tail of an accepted early round: close_early does the rest)
    if (tid == 0) {
      const unsigned long long cl_expected = (unsigned long long)(cst[4] += parts - 1);
      long long spins = 0;
      int ok = 1;
      while (__hip_atomic_load(csync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < cl_expected) {
        if (++spins >= kSpinCap) { ok = 0; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      acc.aux[0] = ok;
      // Form (A), the reader's side: the helpers' shares of fx / fu are complete at agent scope; the backward pass reads them with
      // plain loads, so this CU's vector L1 - which may hold last iteration's lines of fx / fu - is dropped here, once, by the
      // thread that saw the counter (tools/ubench/l1_probe.hip: ~1.7 us; `buffer_inv sc0`, which stood here in round 5, drops nothing)
      cluster_acquire();
      if (acc.aux[1] != 0) cst[1] += 1;                     // (diagnostic: rounds whose helpers all sat on this workgroup's XCD)
    }
    __syncthreads();
    return acc.aux[0] != 0;
  };
  // leader, early linearization: open a round for the trial the line search is about to roll out (false: no helper is there yet)
  auto open_early = [&]() __attribute__((always_inline)) -> bool {
    __syncthreads();
    if (tid == 0) {
      const unsigned long long word = __hip_atomic_load(csync + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long alive = word & 0xffffull;
      const unsigned parts = 1u + (unsigned)(alive < (unsigned long long)(G - 1) ? alive : (unsigned long long)(G - 1));
      const bool same = ((word >> (16 + 6 * xcc_id())) & 63ull) == alive;
      // Any placement (round 6).  The helpers of an early round write ALL of fx, fu, so no byte of fx / fu may sit dirty in this
      // XCD's L2 when a helper behind another L2 writes it: every writer of fx / fu in a clustered launch - the helpers, the leader's
      // own share (jac_own), the lazy zero-fill - uses agent-scope write-through stores, which leave no dirty line behind.  (Round 5
      // had the leader's share in plain stores and therefore refused early rounds across XCDs: "the stale line wins" was seen.)
      acc.aux[3] = (int)parts;
      acc.aux[1] = same ? 1 : 0;                            // (diagnostic only)
      if (parts > 1) {
        const unsigned cl_round = (unsigned)(cst[0] += 1);
        cst[4] += (int)parts - 1;
        cst[2] += 1;
        // the progress word first: a helper that sees the command must not take last round's final count for this round's
        // (both are agent-scope stores of ONE thread to different addresses: drained in between)
        __hip_atomic_store(csync + 4, (unsigned long long)cl_round << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        drain_stores();
        __hip_atomic_store(csync + 0, ((unsigned long long)cl_round << 32) | (1ull << 16) | ((unsigned long long)(same ? 1 : 0) << 17) | parts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      acc.aux[2] = cst[0];
    }
    __syncthreads();
    return acc.aux[3] > 1;
  };
  // leader, candidate groups: the helpers roll out the candidates 4 .. 4 parts - 1 of the line search's first pass (false: no helper
  // there).  Form (B): the gains (K, kappa, dV: the backward pass's plain stores), the nominal trajectory and x0 are read by the
  // helpers with plain loads - every wave drains, ONE release fence, then the command.
  auto open_groups = [&]() __attribute__((always_inline)) -> bool {
    cluster_publish_barrier();
    if (tid == 0) {
      const unsigned long long word = __hip_atomic_load(csync + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long alive = word & 0xffffull;
      const unsigned parts = 1u + (unsigned)(alive < (unsigned long long)(G - 1) ? alive : (unsigned long long)(G - 1));
      const bool same = ((word >> (16 + 6 * xcc_id())) & 63ull) == alive;
      acc.aux[3] = (int)parts;
      acc.aux[1] = same ? 1 : 0;                            // (diagnostic only)
      if (parts > 1) {
        const unsigned cl_round = (unsigned)(cst[0] += 1);
        cst[4] += (int)parts - 1;
        cst[5] += 1;
        cluster_release();
        __hip_atomic_store(csync + 0, ((unsigned long long)cl_round << 32) | (4ull << 16) | ((unsigned long long)(same ? 1 : 0) << 17) | parts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();
    return acc.aux[3] > 1;
  };
  // ... and close it: every helper of the round has reported in (its blocks finished, or called off).  False on a lost helper.
  auto close_early = [&](bool use) __attribute__((always_inline)) -> bool {
    __syncthreads();
    const bool same_l2 = acc.aux[1] != 0;
    if (tid == 0) {
      const unsigned long long cl_expected = (unsigned long long)cst[4];
      long long spins = 0;
      int ok = 1;
      while (__hip_atomic_load(csync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < cl_expected) {
        if (++spins >= kSpinCap) { ok = 0; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      acc.aux[0] = ok;
      if (use && !groups_round) { cst[3] += 1; cst[1] += same_l2 ? 1 : 0; }
      // The reader's side of form (A) (an early round: fx, fu are the helpers' agent-scope stores) and of form (B) (a candidate-group
      // round: the winner's trial trajectory is a helper's plain stores behind its release fence): ONE agent-scope acquire by the
      // thread that saw the counter, the barrier below hands it to the workgroup.
      if (use) cluster_acquire();
    }
    __syncthreads();
    return acc.aux[0] != 0;
  };
  auto do_linearize = [&](bool have_copy) __attribute__((always_inline)) {
    if (lin_staged && !have_copy) {
      double* xs_ = lds + Ly::oT1;
      double* us_ = lds + Ly::oF;
      for (int e = tid; e < n * N; e += kLargeThreads) xs_[(e / n) * kXS + e % n] = v.X[e];
      for (int e = tid; e < m * (N - 1); e += kLargeThreads) us_[(e / m) * kUS + e % m] = v.U[e];
      __syncthreads();
    }
    return linearize_generic(acc, a.kp_method, a.minN, a.maxN, a.jerk_thr, a.err_thr, jac);
  };

  if (MODE == MODE_ROLLOUT) {
    double ex;
    const double L = large_rollout<M>(v, lds, a, x0g, a.stage_in[b], ex);
    if (tid == 0) { a.trial_cost[2 * b] = L; a.trial_cost[2 * b + 1] = ex; }
    return;
  }
  if (MODE == MODE_LINEARIZE) {
    const int nk = do_linearize(false);
    for (int i = tid; i < nk; i += kLargeThreads) a.kp_list[(size_t)b * (N - 1) + i] = acc.kp[i];
    if (tid == 0) a.kp_count[b] = nk;
    return;
  }
  if (MODE == MODE_BACKWARD) {
    backward_pass<M, PIV>(v, lds, nullptr, false);
    __syncthreads();
    if (tid == 0) a.status[b] = lds[Ly::oRed + kPdFlag] == 0.0 ? MI_STATUS_CONVERGED : (a.pd_continue ? MI_STATUS_FLAG_INDEFINITE : MI_STATUS_NOT_PD);
    return;
  }

  // MODE_SOLVE / MODE_FORWARD / MODE_MPC: the Solve loop (ilqr.py:680-708); MODE_MPC wraps it in the
  // receding-horizon loop of the callers (mini_cheetah.py:190-201) so 100 re-solves are one launch.
  double L = (MODE == MODE_FORWARD) ? a.stage_in[b] : __builtin_inf();
  int iters = 0, ls_total = 0, nk = 0;
  int status = MI_STATUS_CONVERGED;
  double* hist = a.hist + (size_t)b * a.hist_cap * 4;
  long long c_ls = 0, c_lin = 0, c_bp = 0;
  const long long c_begin = clock64();
  const int n_solves = (MODE == MODE_MPC) ? a.mpc_resolves : 1;
  // mid-size kernels: the trial buffers of line-search candidates 1..3 (mid_rollout4), and whether this problem has backtracked
  const size_t sx_ = (size_t)a.B * n * N, su_ = (size_t)a.B * m * (N - 1);
  double* const xsp = kSpecRollout<M> ? a.x_spec + (size_t)b * n * N : nullptr;
  double* const usp = kSpecRollout<M> ? a.u_spec + (size_t)b * m * (N - 1) : nullptr;
  bool backtracked = false;
  bool met_indefinite = false;                               // on_indefinite = continue: a backward pass of this launch inverted a Quu that is not positive definite
  for (int rs = 0; rs < n_solves; ++rs) {
    if (MODE == MODE_MPC) {
      // warm start (mini_cheetah.py:193-198): x0 <- x_bar[:, replan]; u_bar <- [u_bar[:, replan:], repeat(last)]
      const int r = a.mpc_replan;
      double* x0w = const_cast<double*>(x0g);
      constexpr int UPT = 8;                               // (m*(N-1) + 255) / 256 <= 8 for every admissible N
      double ush[UPT];
#pragma unroll
      for (int q = 0; q < UPT; ++q) {
        const int e = tid + kLargeThreads * q;             // element (t, k) of the time-major u_bar
        const int t = e / m, k = e - t * m;
        const int src = (t + r < N - 1) ? t + r : N - 2;
        ush[q] = (e < m * (N - 1)) ? v.U[(size_t)src * m + k] : 0.0;
      }
      double x0n = (tid < n) ? v.X[(size_t)r * n + tid] : 0.0;
      __syncthreads();
#pragma unroll
      for (int q = 0; q < UPT; ++q) {
        const int e = tid + kLargeThreads * q;
        if (e < m * (N - 1)) v.U[e] = ush[q];
      }
      if (tid < n) {
        x0w[tid] = x0n;
        // moving target (mini_cheetah.py:151-156) and the constants derived from it (ilqr.py:180,203)
        lds[Ly::oXnom + tid] += a.mpc_target_step[tid];
      }
      __syncthreads();
      if (tid < n) {
        double s_ = 0.0, sf_ = 0.0;
        for (int i = 0; i < n; ++i) {
          s_ += (2.0 * lds[Ly::oXnom + i]) * lds[Ly::oQ + i * n + tid];
          sf_ += (2.0 * lds[Ly::oXnom + i]) * lds[Ly::oQf + i * n + tid];
        }
        lds[Ly::oQn + tid] = s_; lds[Ly::oQfn + tid] = sf_;
      }
      __syncthreads();
      L = __builtin_inf();
      status = MI_STATUS_CONVERGED;                          // per re-solve, as a host loop of Solve() calls would leave it
    }
    double improvement = __builtin_inf();
    int it_this = 0;
    while (improvement > a.delta) {
      if (it_this >= a.max_iters) { status = MI_STATUS_MAX_ITERS; break; }
      double L_new, eps; int trials;
      const long long c0 = clock64();
      bool ok, used_spec = false, early = false, lost = false;
      int win = 0;
      if constexpr (kSpecRollout<M>) {
        if ((a.spec_policy == 2 || (a.spec_policy == 1 && backtracked)) && L < __builtin_inf()) {
          used_spec = true;
          bool par = false, collected = false;
          if (clustered && ls_groups) par = open_groups();
          const int groups = par ? acc.aux[3] : 1;
          auto collect = [&]() __attribute__((always_inline)) { groups_round = true; lost = !close_early(true); groups_round = false; collected = true; };
          ok = mid_linesearch4<M>(v, xsp, usp, sx_, su_, lds, a, x0g, L, L_new, eps, trials, win, groups, csync + 8, collect);
          if (par && !collected) collect();                                           // (an accepted candidate of the leader's own four)
          if (lost) { status = MI_STATUS_INTERNAL; break; }
        } else {
          if constexpr (kEarlyLin<M>) { if (clustered && early_lin) early = open_early(); }
          ok = large_linesearch<M>(v, lds, a, x0g, L, L_new, eps, trials, early ? csync + 4 : nullptr, (unsigned long long)(unsigned)acc.aux[2] << 32);
        }
        backtracked = backtracked || trials > 1;
      } else {
        if constexpr (kEarlyLin<M>) { if (clustered && early_lin) early = open_early(); }
        ok = large_linesearch<M>(v, lds, a, x0g, L, L_new, eps, trials, early ? csync + 4 : nullptr, (unsigned long long)(unsigned)acc.aux[2] << 32);
      }
      ls_total += trials;
      // (a failed search after an early round: its helpers have written a rejected trial's Jacobians over fx, fu - the stage below
      //  runs once more, on the nominal trajectory, before the solve stops; the reference's fx, fu are the last linearization's)
      if (!ok && !early) { status = MI_STATUS_LINESEARCH_FAILED; break; }
      __syncthreads();
      const long long c1 = clock64();
      const double* const Xw = !ok ? v.X : (win == 0 ? v.Xn : xsp + (size_t)(win - 1) * sx_);      // the accepted trial
      const double* const Uw = !ok ? v.U : (win == 0 ? v.Un : usp + (size_t)(win - 1) * su_);
      {                                                                              // :375-376 (+ the LDS copy the
        double* xs_ = lds + Ly::oT1;                                                 //  linearization reads)
        double* us_ = lds + Ly::oF;
        if (clustered) {                                                             // (the helpers read x_bar / u_bar: write-through)
          for (int e = tid; e < n * N; e += kLargeThreads) { const double x_ = Xw[e]; st_shared<true>(v.X + e, x_); xs_[(e / n) * kXS + e % n] = x_; }
          for (int e = tid; e < m * (N - 1); e += kLargeThreads) { const double u_ = Uw[e]; st_shared<true>(v.U + e, u_); us_[(e / m) * kUS + e % m] = u_; }
        } else {
          for (int e = tid; e < n * N; e += kLargeThreads) { const double x_ = Xw[e]; v.X[e] = x_; if (lin_staged) xs_[(e / n) * kXS + e % n] = x_; }
          for (int e = tid; e < m * (N - 1); e += kLargeThreads) { const double u_ = Uw[e]; v.U[e] = u_; if (lin_staged) us_[(e / m) * kUS + e % m] = u_; }
        }
      }
      __syncthreads();
      const bool early_hit = early && ok && trials == 1;                             // the helpers have linearized THIS trajectory
      if constexpr (EarlyLeaderBlocks<M>::value > 0) {                               // ... but for the last block(s), which are the leader's
        if (early_hit && early_helper_blocks < early_blocks) linearize_clustered(true, early_helper_blocks * kEarlyBlock);
      }
      if (early && !close_early(early_hit)) { status = MI_STATUS_INTERNAL; break; }
      if (early_hit) {
        nk = N - 1;
      } else if (clustered) {                                                        // :370, shared with the helper workgroups
        nk = N - 1;
        if (!linearize_clustered(false, 0)) { status = MI_STATUS_INTERNAL; break; }
      } else {
        nk = do_linearize(true);                                                     // :370
      }
      if (!ok) { status = MI_STATUS_LINESEARCH_FAILED; break; }
      __syncthreads();
      const long long c2 = clock64();
#ifdef MI_PROF_BACKWARD
      // profiling build only: 16 phase accumulators of thread 0 (a matrix-core wave) and of thread
      // 192 (the spare wave) land in the last 8 rows of the history buffer (tools/bp_prof.py)
      long long bpa[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      if (MODE != MODE_FORWARD) { backward_pass<M, PIV>(v, lds, bpa, kLxFromRollout<M> && !v.asym); __syncthreads(); }
      if ((tid == 0 || tid == 192) && iters == 0 && it_this == 0) {
        double* hp = a.hist + (size_t)b * a.hist_cap * 4 + 4 * (a.hist_cap - (tid == 0 ? 4 : 8));
        for (int q_ = 0; q_ < 16; ++q_) hp[q_] = (double)bpa[q_];
      }
#else
      // (a four-candidate pass leaves no cost gradients behind: the backward pass forms them itself)
      // (nor does any rollout when Q is not symmetric: its cost rows hold 2 Q (x - x_nom), the reference's lx is 2 Q x - 2 Q^T x_nom, :180)
      if (MODE != MODE_FORWARD) { backward_pass<M, PIV>(v, lds, nullptr, kLxFromRollout<M> && !used_spec && !v.asym, lin_staged && !IsChainModel<M>::value); __syncthreads(); }      // :697
#endif
      const long long c3 = clock64();
      const bool not_pd = MODE != MODE_FORWARD && lds[Ly::oRed + kPdFlag] != 0.0;     // a Quu of this backward pass was not positive definite
      c_ls += c1 - c0; c_lin += c2 - c1; c_bp += c3 - c2;
      if (tid == 0 && it_this < a.hist_cap) {                                        // history of the LAST solve
        hist[4 * it_this + 0] = L_new; hist[4 * it_this + 1] = eps;
        hist[4 * it_this + 2] = (double)trials; hist[4 * it_this + 3] = (double)nk / (double)(N - 1) * 100.0;
        double* ic = a.iter_cyc + ((size_t)b * a.hist_cap + it_this) * 4;             // per-iteration stopwatches (ilqr.py:364-372,696-702)
        ic[0] = (double)(c1 - c0); ic[1] = (double)(c2 - c1); ic[2] = (double)(c3 - c2); ic[3] = (double)(c3 - c0);
      }
      improvement = L - L_new;
      L = L_new;
      it_this += 1;
      if (MODE == MODE_FORWARD) break;
      if (not_pd && !a.pd_continue) { status = MI_STATUS_NOT_PD; break; }      // the gains of that pass are not to be used (unless asked to: on_indefinite)
      met_indefinite = met_indefinite || not_pd;
    }
    iters += it_this;
    if (MODE == MODE_MPC) {
      double* lg = a.mpc_log + ((size_t)b * a.mpc_resolves + rs) * (n + 2);
      if (tid < n) lg[tid] = x0g[tid];
      if (tid == 0) { lg[n] = L; lg[n + 1] = (double)it_this; }
      if (status >= MI_STATUS_LINESEARCH_FAILED) break;
    }
  }
  if (G > 1 && tid == 0) __hip_atomic_store(csync + 5, ((unsigned long long)(unsigned)cst[2] << 32) | (unsigned)cst[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (G > 1 && tid == 0) __hip_atomic_store(csync + 7, (unsigned long long)(unsigned)cst[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // helpers: go home (any non-zero value; the rest of the word is for MI_I64_CLUSTER_WORDS: rounds << 32 | same-L2 rounds << 8 | 1)
  if (G > 1 && tid == 0) __hip_atomic_store(csync + 3, ((unsigned long long)(unsigned)cst[0] << 32) | ((unsigned long long)((unsigned)cst[1] & 0xffffffu) << 8) | 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  for (int i = tid; i < nk; i += kLargeThreads) a.kp_list[(size_t)b * (N - 1) + i] = acc.kp[i];
  if (tid == 0) {
    a.cost[b] = L; a.iters[b] = iters; a.status[b] = status | (met_indefinite ? MI_STATUS_FLAG_INDEFINITE : 0); a.ls_trials[b] = ls_total; a.kp_count[b] = nk;
#ifndef MI_PROF_BACKWARD
    a.prof[4 * b + 0] = c_ls; a.prof[4 * b + 1] = c_lin; a.prof[4 * b + 2] = c_bp; a.prof[4 * b + 3] = clock64() - c_begin;
#endif
  }
}

}  // namespace mi
