// Host-side pieces shared by the translation units of libmi_ilqr.so: the handle, the error macro and the
// kernel-launch templates.  mi_ilqr.hip holds the C ABI; every model's kernels are instantiated in their own
// k_<model>.hip (built in parallel, drake_ddp_amd/build.py) behind one `launch_<model>(handle, mode, args)` each.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/mi_ilqr.h"
#include "ilqr_small.hpp"


struct mi_ilqr {
  mi_ilqr_desc d;
  int n, m, N, B;
  hipStream_t stream = nullptr;
  // Per-launch records (kernel start/stop events + aggregate statistics) live in a ring, so that up to
  // kStatsRing solves can be enqueued back to back (mi_ilqr_solve_async) before anything is collected;
  // ev0/ev1/h_stats/d_stats alias the slot of the most recent launch.
  static constexpr int kStatsRing = 32;
  hipEvent_t ring_ev0[kStatsRing] = {}, ring_ev1[kStatsRing] = {};
  mi::DevStats* h_ring = nullptr;    // pinned host memory, device-mapped
  mi::DevStats* d_ring = nullptr;    // its device alias
  long long seq = 0;             // solves enqueued so far
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  unsigned long long seq_timed = 0;
  int time_every = 1;              // mi_ilqr_set_timing: events on one solve in `time_every` (0 = never)
  bool timed_launch = true;        // this launch carries the events
  bool last_timed = true;          // ... and so did the most recent launch
  int cur_slot = 0;
  bool ring_timed[kStatsRing] = {};
  // double fields
  double *x_bar = nullptr, *u_bar = nullptr, *K = nullptr, *kappa = nullptr, *dV = nullptr, *fx = nullptr, *fu = nullptr;
  double *x0 = nullptr, *u_guess = nullptr, *cost = nullptr, *hist = nullptr, *iter_cyc = nullptr;
  double *x_trial = nullptr, *u_trial = nullptr, *trial_cost = nullptr, *stage_in = nullptr, *costmat = nullptr;
  int32_t *iters = nullptr, *status = nullptr, *ls_trials = nullptr, *kp_count = nullptr, *kp_list = nullptr;
  // cost / iters / status / ls_trials above alias the CURRENT slot of these rings (kStatsRing x B each): every
  // pipelined solve leaves its per-problem results in its own slot, so their reduction to a DevStats record can
  // wait until somebody collects - one stats_kernel launch over all pending slots - instead of one dispatch
  // (6 us + its gap) behind every solve.
  double* cost_ring = nullptr;
  int32_t *iters_ring = nullptr, *status_ring = nullptr, *ls_ring = nullptr;
  long long stats_done = 0;      // solves with sequence number < stats_done have their DevStats record
  bool in_async_solve = false;
  double* u_one = nullptr;         // device copy of a shared (m, N-1) initial guess
  char* pin_in = nullptr;          // page-locked staging ring of small host -> device inputs (stage_h2d)
  size_t pin_off = 0;
  hipEvent_t pin_ev = nullptr;
  long long* prof = nullptr;
  int32_t* done_counter = nullptr;   // wave-per-problem kernels: tickets of the in-kernel statistics epilogue
  mi::DevStats* h_stats = nullptr;   // pinned host memory, device-mapped
  mi::DevStats* d_stats = nullptr;   // its device alias
  double* mpc_log = nullptr;     // (B, mpc_log_resolves, n+2)
  int mpc_log_resolves = 0;
  int mpc_resolves = 0, mpc_replan = 0;
  double mpc_target_step[mi::kMaxStateDim] = {};
  bool cold = true;        // persistent state is known to be all zero (fresh object / after reset)
  bool u_pending = false;  // SetInitialGuess input waiting in u_guess
  bool u_zero = false;     // u_bar is to read as all zero (after reset, until a guess is set or re-armed): ilqr.py:71
  int exact_backward = 0;  // cost matrices the fast backward forms do not cover (asymmetric / indefinite): reference recursion
  // tiny batches (B <= 4): the per-solve records - iteration log, stopwatches, iterations, status - live in page-locked host memory the
  // kernels write directly (hist, iter_cyc, prof, iters_ring, status_ring point into it): mi_ilqr_solve_into reads them with a memcpy
  // after the solve's one synchronization instead of five device-to-host copies queued behind the kernel (~5 us each)
  char* host_records = nullptr;       // host address of the block (hipHostMalloc, mapped)
  char* host_records_dev = nullptr;   // the same block as the device sees it
  size_t host_records_bytes = 0;
  bool host_inputs = false;           // x0 and u_guess live in the block too (B <= 4, wave-per-problem kernels)
  int q_diag = 0;          // Q has no off-diagonal entry (KArgs::q_diag)
  int cost_asym = 0;       // workgroup-per-problem kernels, n <= 32: Q, R or Qf is not symmetric (mid_backward then uses no symmetry at all)
  std::vector<double> h_costmat;   // host mirror of costmat (Q | R | Qf | x_nom)
  bool costmat_synced = false;     // the device copy equals the mirror
  unsigned long long* cluster_sync = nullptr;   // workgroup-per-problem kernels: kSyncWords handshake words per problem
  bool last_clustered = false;     // the last MODE_SOLVE / MODE_MPC launch shared its linearizations among clusters (else MI_I64_CLUSTER_WORDS reads as zeros)
  int n_cus = 0;                   // compute units of the device
  double* scratch = nullptr;       // device staging area of the boundary's layout conversions (grow-only)
  size_t scratch_bytes = 0;
  size_t lds = 0;
  bool large = false;      // workgroup-per-problem path: state arrays are TIME-MAJOR in HBM
  int n_store = 1;         // line-search candidate trajectories kept in LDS
  bool batch_minor = false; // lane-per-problem path: state arrays are [t][row][b] in HBM
  double* lxu = nullptr;             // workgroup-per-problem kernels, long horizons: cost gradients in HBM
  int spec_slots = 0;              // trial trajectories x_spec / u_spec hold per problem (3: one workgroup's four candidates; 31: candidate groups)
  double *x_spec = nullptr, *u_spec = nullptr;   // mid-size kernels: trial trajectories of three more line-search candidates
  int32_t* bm_scratch = nullptr;   // lane-per-problem kernels with key-points: integer scratch (ilqr_batch.hpp)
  double *sink_x = nullptr, *sink_u = nullptr, *sink_cost = nullptr;   // result sink (device aliases of host arrays), optional
};

// Small batches of the wave-per-problem kernels aggregate the batch statistics in the solve kernel
// itself (last workgroup to finish): a blocking single-problem solve saves a kernel launch, 5 us of
// 98.  Large batches keep the separate stats_kernel: with pipelined solves the two cost the same per
// step (measured at B = 1024: 0.166 ms either way), and the solve kernel stays 3.6 us shorter.
// MI_ILQR_STATS_KERNEL=1 / =0 forces the separate kernel / the in-kernel epilogue (A/B runs).
static inline bool stats_in_kernel(const mi_ilqr* h) {
  static const int forced = [] { const char* e = std::getenv("MI_ILQR_STATS_KERNEL"); return !e ? -1 : (e[0] == '1' ? 1 : 0); }();
  if (h->large || h->batch_minor) return false;
  if (forced >= 0) return forced == 0;
  return h->B <= 64;
}
static inline void select_stats_slot(mi_ilqr* h, int slot) {
  h->ev0 = h->ring_ev0[slot]; h->ev1 = h->ring_ev1[slot];
  h->h_stats = h->h_ring + slot; h->d_stats = h->d_ring + slot;
  h->cur_slot = slot;
  const size_t o = (size_t)slot * h->B;
  h->cost = h->cost_ring + o; h->iters = h->iters_ring + o; h->status = h->status_ring + o; h->ls_trials = h->ls_ring + o;
}

namespace mi_host {
using namespace mi;
#define HIPCHK(expr)                                                                      \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) {                                                               \
      std::fprintf(stderr, "mi_ilqr: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return MI_ILQR_E_HIP;                                                               \
    }                                                                                     \
  } while (0)

constexpr size_t kMaxLds = 160 * 1024;
constexpr int kMaxBatchPluginN = 6;      // family-0 plugin models up to this n also get the lane-per-problem kernels

// The dynamic-LDS ceiling of a kernel is raised once per (kernel, device), to the hardware maximum - not
// on every launch.
constexpr int kMaxDevices = 64;
template <class Kern>
int allow_max_lds(Kern kern, bool (&done)[kMaxDevices], int device) {
  if (device >= 0 && device < kMaxDevices && done[device]) return MI_ILQR_OK;
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds));
  if (device >= 0 && device < kMaxDevices) done[device] = true;
  return MI_ILQR_OK;
}

// One dispatch packet per solve: the launch carries the handle's start/stop events itself (the kernel's own
// begin/end timestamps, what rocprofv3 reports for it) instead of two hipEventRecord marker packets around it.
// A profiled dispatch still costs the stream ~5 us of serialization in a pipelined sequence
// (tools/ubench/gap.hip), so mi_ilqr_set_timing can restrict the events to one launch in k.
// `part`: 0 = a launch on its own (start and stop events); 1 / 2 = first / last kernel of a solve made of two launches
// (the start event rides on the first, the stop event on the last: the elapsed time covers both and the gap).
template <class Kern>
int launch_timed(mi_ilqr* h, Kern kern, dim3 grid, dim3 block, size_t lds, const KArgs& a, int part = 0) {
  KArgs args = a;
  void* argv[] = {&args};
  h->ring_timed[h->cur_slot] = h->last_timed = h->timed_launch;
  if (!h->timed_launch) { HIPCHK(hipLaunchKernel(reinterpret_cast<const void*>(kern), grid, block, argv, lds, h->stream)); return MI_ILQR_OK; }
  HIPCHK(hipExtLaunchKernel(reinterpret_cast<const void*>(kern), grid, block, argv, lds, h->stream, part == 2 ? nullptr : h->ev0,
                            part == 1 ? nullptr : h->ev1, 0));
  return MI_ILQR_OK;
}

}  // namespace mi_host

// one per kernel translation unit (library-internal: hidden from the dynamic symbol table, the C ABI is mi_ilqr.h)
#define MI_INTERNAL __attribute__((visibility("hidden")))
MI_INTERNAL int launch_pendulum(mi_ilqr* h, int mode, const mi::KArgs& a);
MI_INTERNAL int launch_acrobot(mi_ilqr* h, int mode, const mi::KArgs& a);
MI_INTERNAL int launch_cartpole(mi_ilqr* h, int mode, const mi::KArgs& a);
MI_INTERNAL int launch_cartpole_wall(mi_ilqr* h, int mode, const mi::KArgs& a);
MI_INTERNAL int launch_synth36(mi_ilqr* h, int mode, const mi::KArgs& a);
MI_INTERNAL int launch_planar_quad(mi_ilqr* h, int mode, const mi::KArgs& a);
MI_INTERNAL int launch_quad3d(mi_ilqr* h, int mode, const mi::KArgs& a);
MI_INTERNAL int launch_arm27(mi_ilqr* h, int mode, const mi::KArgs& a);
MI_INTERNAL int launch_arm27c(mi_ilqr* h, int mode, const mi::KArgs& a);
MI_INTERNAL int launch_batch_minor(mi_ilqr* h, int mode, const mi::KArgs& a);
