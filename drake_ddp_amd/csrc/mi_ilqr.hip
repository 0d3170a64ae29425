// libmi_ilqr.so — host side of the C ABI declared in include/mi_ilqr.h.
//
// Owns the device-resident solver state of a batch (the persistent attributes of
// the reference class, /root/reference/ilqr.py:61-91, with a leading batch axis),
// dispatches the gfx950 kernels and moves data across the boundary.  No compute
// happens on the host: without a usable device every compute entry fails with
// MI_ILQR_E_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <dlfcn.h>
#include <rccl/rccl.h>     // types and enum VALUES only (ncclFloat64, ncclMin): the symbols are bound at run time (dlopen below)

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/mi_ilqr.h"
#include "ilqr_batch.hpp"
#include "ilqr_large.hpp"
#include "ilqr_small.hpp"

using namespace mi;


#include "host.hpp"

using namespace mi_host;

namespace {


struct ModelInfo { int n, m, n_params; double defaults[MI_ILQR_MAX_PARAMS]; };

// models registered at run time (mi_ilqr_register_model): id = MI_MODEL_PLUGIN_BASE + slot
// Registrations take the mutex; a slot is filled once and read lock-free afterwards: `used` is published LAST with release
// ordering and read with acquire ordering, so a thread that sees it set also sees the slot's contents (a create or a launch on
// one thread may race a registration on another).
struct PluginSlot { ModelInfo info; mi_ilqr_model_plugin p; std::atomic<bool> used{false}; };
PluginSlot g_plugins[MI_ILQR_MAX_PLUGINS];
std::mutex g_plugins_mutex;
const PluginSlot* plugin_of(int id) {
  const int s_ = id - MI_MODEL_PLUGIN_BASE;
  return (s_ >= 0 && s_ < MI_ILQR_MAX_PLUGINS && g_plugins[s_].used.load(std::memory_order_acquire)) ? &g_plugins[s_] : nullptr;
}

const ModelInfo* model_info(int id) {
  if (const PluginSlot* ps = plugin_of(id)) return &ps->info;
  static const ModelInfo table[] = {
      {2, 1, 3, {0.25, 0.1, 4.905}},
      {4, 1, 10, {1.0, 1.0, 1.0, 0.5, 1.0, 0.083, 0.33, 0.1, 0.1, 9.81}},
      {4, 1, 4, {10.0, 1.0, 0.5, 9.81}},
      {4, 1, 8, {10.0, 1.0, 0.5, 9.81, -0.45, 0.05, 2000.0, 0.01}},
      {36, 12, 4, {4.0, 0.5, 6.0, 0.1}},
      {36, 12, 9, {9.81, 4000.0, 0.004, 0.3, 0.15, 0.05, 0.02, 2.0, 60.0}},
      {37, 12, 14, {9.81, 4000.0, 0.004, 0.3, 0.15, 0.3, 60.0, 9.0, 0.07, 0.26, 0.28, 0.06, 0.06, 0.04}},
      {27, 7, 15, {9.81, 1500.0, 0.005, 0.5, 1.0, 0.5, 0.2, 0.1, 0.05, 1.0, 0.8, 0.6, 0.3, 0.1, 0.04}},
      {27, 7, 16, {9.81, 1500.0, 0.005, 0.5, 1.0, 0.5, 0.2, 0.1, 0.05, 1.0, 0.8, 0.3, 0.15, 0.05, 0.04, 0.6}},
  };
  if (id < 0 || id > 8) return nullptr;
  return &table[id];
}

size_t small_lds_bytes(int model_id, int N, int n_store = 1) {
  if (const PluginSlot* ps = plugin_of(model_id)) return ps->p.family == 0 ? ps->p.lds_bytes(N, n_store) : 0;
  switch (model_id) {
    case MI_MODEL_PENDULUM: return ws_bytes<2, 1>(N, n_store);
    case MI_MODEL_ACROBOT:
    case MI_MODEL_CARTPOLE:
    case MI_MODEL_CARTPOLE_WALL: return ws_bytes<4, 1>(N, n_store);
    default: return 0;
  }
}

size_t large_lds(int model_id, int N) {
  if (const PluginSlot* ps = plugin_of(model_id)) return ps->p.family == 1 ? ps->p.lds_bytes(N, 1) : 0;
  switch (model_id) {
    case MI_MODEL_SYNTH36: return large_lds_bytes<Synth36::n, Synth36::m>(N);
    case MI_MODEL_PLANAR_QUAD: return large_lds_bytes<PlanarQuad::n, PlanarQuad::m>(N);
    case MI_MODEL_QUAD3D: return large_lds_bytes<Quad3D::n, Quad3D::m>(N);
    case MI_MODEL_ARM27: case MI_MODEL_ARM27C: return large_lds_bytes<Arm27::n, Arm27::m>(N);
    default: return 0;
  }
}

// LDS of the workgroup-per-problem kernels with the cost gradients in HBM (long horizons)
size_t large_lds_hbm(int model_id, int N) {
  if (const PluginSlot* ps = plugin_of(model_id)) return ps->p.family == 1 ? ps->p.lds_bytes(N, -1) : 0;
  switch (model_id) {
    case MI_MODEL_SYNTH36: return large_lds_bytes_hbm<Synth36::n, Synth36::m>(N);
    case MI_MODEL_PLANAR_QUAD: return large_lds_bytes_hbm<PlanarQuad::n, PlanarQuad::m>(N);
    case MI_MODEL_QUAD3D: return large_lds_bytes_hbm<Quad3D::n, Quad3D::m>(N);
    case MI_MODEL_ARM27: case MI_MODEL_ARM27C: return large_lds_bytes_hbm<Arm27::n, Arm27::m>(N);
    default: return 0;
  }
}

KArgs make_args(const mi_ilqr* h) {
  KArgs a;
  std::memset(&a, 0, sizeof(a));
  a.x_bar = h->x_bar; a.u_bar = h->u_bar; a.K = h->K; a.kappa = h->kappa; a.dV = h->dV; a.fx = h->fx; a.fu = h->fu;
  a.x0 = h->x0; a.u_guess = h->u_guess; a.cost = h->cost; a.hist = h->hist; a.iter_cyc = h->iter_cyc;
  a.x_trial = h->x_trial; a.u_trial = h->u_trial; a.trial_cost = h->trial_cost; a.stage_in = h->stage_in;
  a.costmat = h->costmat;
  a.iters = h->iters; a.status = h->status; a.ls_trials = h->ls_trials; a.kp_count = h->kp_count; a.kp_list = h->kp_list;
  a.prof = h->prof;
  for (int i = 0; i < MI_ILQR_MAX_PARAMS; ++i) a.params[i] = h->d.model_params[i];
  a.dt = h->d.dt; a.delta = h->d.delta; a.beta = h->d.beta; a.gamma = h->d.gamma;
  a.jerk_thr = h->d.jerk_threshold; a.err_thr = h->d.iterative_error_threshold; a.fd_h = h->d.fd_step;
  a.N = h->N; a.B = h->B; a.kp_method = h->d.keypoint_method; a.minN = h->d.minN; a.maxN = h->d.maxN;
  a.max_iters = h->d.max_iters; a.hist_cap = h->d.hist_cap;
  a.n_store = h->n_store;
  a.cold = h->cold ? 1 : 0;
  a.u_pending = h->u_pending ? 1 : 0;
  a.mpc_resolves = h->mpc_resolves; a.mpc_replan = h->mpc_replan; a.mpc_log = h->mpc_log;
  for (int i = 0; i < kMaxStateDim; ++i) a.mpc_target_step[i] = h->mpc_target_step[i];
  // wave-per-problem kernels: helper wavefronts share the linearization when every step is a key-point
  // (the default) - as many as keep the whole batch resident at ONE wave per SIMD (1024 slots): every
  // wave of the kernel carries the main wave's register allocation (> 256 VGPRs with the time-parallel
  // rollout and sweep), so a second resident wave per SIMD does not fit and extra waves would queue.
  // MI_ILQR_NO_HELPER=1 turns them off (A/B measurements).
  static const bool no_helper = [] { const char* e = std::getenv("MI_ILQR_NO_HELPER"); return e && e[0] == '1'; }();
  static const bool seq_bp = [] { const char* e = std::getenv("MI_ILQR_SEQ_BACKWARD"); return e && e[0] == '1'; }();
  a.seq_backward = h->exact_backward ? 2 : (seq_bp ? 1 : 0);
  static const bool seq_ro = [] { const char* e = std::getenv("MI_ILQR_SEQ_ROLLOUT"); return e && e[0] == '1'; }();
  a.newton_rollout = seq_ro ? 0 : 1;
  // (not for n = 2 with the time-parallel rollout: its final pass differentiates the steps it holds in registers,
  //  which beats sharing them - C2's batch at B = 512: 0.141 ms without helpers, 0.149 ms with one; tools/helper_ab.py)
  const bool fused_linearization = h->n == 2 && h->m == 1 && h->N - 1 <= 256 && a.newton_rollout != 0;
  a.helpers = 0;
  if (!no_helper && !h->large && !h->batch_minor && h->d.keypoint_method == MI_KP_SET_INTERVAL && h->d.minN == 1 &&
      (h->N - 1) * (h->n + h->m) > 128 && !fused_linearization)
    a.helpers = h->B <= 256 ? 3 : (h->B <= 512 ? 1 : 0);
  // wave-per-problem kernels aggregate the batch statistics themselves (MODE_SOLVE / MODE_MPC)
  const bool own_stats = stats_in_kernel(h);
  a.stats_out = own_stats ? h->d_stats : nullptr;
  a.done_counter = own_stats ? h->done_counter : nullptr;
  // workgroup-per-problem kernels: with few problems per GPU most CUs idle - up to 8 workgroups per problem share the
  // linearization (ilqr_large.hpp: cluster handshake), as many as keep every workgroup of the launch on its own CU.
  // MI_ILQR_CLUSTER=k forces k (1 = off) for A/B runs.
  a.sink_x = h->sink_x; a.sink_u = h->sink_u; a.sink_cost = h->sink_cost;
  a.bm_scratch = h->bm_scratch;
  a.x_spec = h->x_spec; a.u_spec = h->u_spec;
  a.lxu = h->lxu;
  a.pd_continue = h->d.on_indefinite == 1 ? 1 : 0;
  a.cost_asym = h->cost_asym ? 1 : 0;
  a.q_diag = h->q_diag ? 1 : 0;
  static const int spec = [] { const char* e = std::getenv("MI_ILQR_SPEC"); return e ? std::atoi(e) : 1; }();
  a.spec_policy = (h->x_spec && spec >= 0 && spec <= 2) ? spec : 0;
  a.cluster = 1;
  a.cluster_sync = h->cluster_sync;
  if (h->large && h->cluster_sync && h->d.keypoint_method == MI_KP_SET_INTERVAL && h->d.minN == 1) {
    static const int forced = [] { const char* e = std::getenv("MI_ILQR_CLUSTER"); return e ? std::atoi(e) : 0; }();
    int g = forced > 0 ? forced : (h->n_cus > 0 ? h->n_cus / h->B : 1);
    // Which models: those whose linearization is worth a handshake (round 5: ~20 k cycles - six memory round trips - since the
    // cache-wide write-backs are gone; with the helpers linearizing the line search's first trial WHILE it is rolled out - early
    // linearization, ilqr_large.hpp - the stage shrinks to the wait for the last block: 36-state chain 74 k -> 12 k cycles at
    // B = 64, 3-D quadruped 92 k -> 30 k, planar quadruped 164 k -> 132 k (its helpers cannot keep up with the rollout)).  The arm
    // + ball's dense linearization (100 k single) up to B = 64; plugin models: the library cannot know what their step costs - a
    // cheap one loses to the handshake, so they are not clustered unless forced.
    if (forced <= 0) {
      const int id = h->d.model_id;
      if (id == MI_MODEL_PLANAR_QUAD || id == MI_MODEL_QUAD3D) {}
      else if (id == MI_MODEL_ARM27 || id == MI_MODEL_ARM27C || id == MI_MODEL_SYNTH36) { if (h->B > 64) g = 1; }
      else if (plugin_of(id)) g = 1;
      else if (h->B > 16) g = 1;
    }
    // Forward-mode duals (a parity / testing mode - the benchmarked path is central differences) are not clustered unless forced.
    // Round 5: the helper path of ilqr_large_kernel<PlanarQuad, JAC = 1, MODE_SOLVE> faulted (HSA memory aperture violation in a
    // flat load of the item loop of large_jac_at_tree) after two unrelated edits inside large_backward, each alone enough, and
    // stopped faulting with a bounds check added next to it.  Under rocgdb the faulting helper wavefront sits at the top of the item
    // loop with an EXEC mask that is no prefix of the lanes (0x316eaa6bf7995fc5) and garbage in the lanes' time index - a state no
    // path of the source produces (the loop's lanes leave in order); the spilled scalars it restores there (361 - 932 SGPRs of these
    // kernels live in VGPR lanes) were written correctly at kernel start.  Root cause not established beyond that - it points at
    // the compiler's handling of this kernel's spills, not at the handshake - so the instantiation is kept off the default path
    // and the full GPU suite stays the safety net for the clustered central-difference kernels, which every bench config runs.
    if (forced <= 0 && h->d.jacobian_mode == MI_JAC_AUTODIFF) g = 1;
    if (g > 8) g = 8;
    if (g < 1) g = 1;
    // placement (MI_ILQR_CLUSTER_ORDER, A/B runs): 2 = a cluster on ONE XCD, the XCD's leaders in its first slots (default: measured
    // best or level for every model once early linearization is on); 1 = one XCD, members in consecutive slots (the quadrupeds'
    // helpers run 40 - 70 % slower next to leaders: neighbouring CUs share an instruction cache, and their linearization loops are
    // 50 - 90 KB of code); 0 = consecutive blocks, a cluster spans XCDs (what rounds 2 - 4 did; no early linearization there)
    static const int order = [] { const char* e = std::getenv("MI_ILQR_CLUSTER_ORDER"); return e ? std::atoi(e) : 2; }();
    // Early linearization (MI_ILQR_EARLY=0|1): every model, built-in or plugin (round 6).  Round 5 opened it for the built-in models
    // only: forced onto plugin chains with steps of a few hundred cycles the helpers linearized rows of the previous trial.  The cause
    // was not the progress word's lag but the helpers' side of the hand-shake - `buffer_inv sc0` does not drop another CU's lines
    // from the vector L1 (tools/ubench/l1_probe.hip), and a small trajectory survives there from one iteration to the next; the
    // helpers now read the trial with agent-scope loads and the progress word follows a full drain (ilqr_large.hpp).
    static const int early_env = [] { const char* e = std::getenv("MI_ILQR_EARLY"); return e ? std::atoi(e) : 1; }();
    const int early = early_env;
    // candidate groups (MI_ILQR_LS_GROUPS=0|1): the helpers need trial buffers of their own, and - they keep their own LDS copy of
    // the cost constants - a target that does not move inside the launch
    static const int groups = [] { const char* e = std::getenv("MI_ILQR_LS_GROUPS"); return e ? std::atoi(e) : 1; }();
    bool still = true;
    for (int i = 0; i < h->n; ++i) still = still && h->mpc_target_step[i] == 0.0;
    const bool lsg = groups && still && h->spec_slots >= 4 * g - 1 && g > 1;
    a.cluster = g | ((order & 3) << 8) | ((early ? 1 : 0) << 10) | ((lsg ? 1 : 0) << 11);
  }
  return a;
}

// Launch `mode` for the handle's model; afterwards the persistent state is no
// longer known-zero for the fields the mode writes, and u_bar is materialized.
int reduce_pending_stats(mi_ilqr* h);

int launch(mi_ilqr* h, int mode) {
  HIPCHK(hipSetDevice(h->d.device_id));
  // any launch other than a pipelined solve reuses the current ring slot: settle the statistics still owed first
  if (!h->in_async_solve) { const int rc = reduce_pending_stats(h); if (rc != MI_ILQR_OK) return rc; }
  if (h->u_zero && !h->u_pending) HIPCHK(hipMemsetAsync(h->u_bar, 0, (size_t)h->B * h->m * (h->N - 1) * 8, h->stream));
  h->u_zero = false;
  const KArgs a = make_args(h);
  int rc;
  if (h->batch_minor) {
    if (const PluginSlot* ps = plugin_of(h->d.model_id)) return ps->p.launch(h, mode, &a);   // (family-0 plugins carry the lane-per-problem kernels too)
    return launch_batch_minor(h, mode, a);
  }
  switch (h->d.model_id) {
    case MI_MODEL_PENDULUM: rc = launch_pendulum(h, mode, a); break;
    case MI_MODEL_ACROBOT: rc = launch_acrobot(h, mode, a); break;
    case MI_MODEL_CARTPOLE: rc = launch_cartpole(h, mode, a); break;
    case MI_MODEL_CARTPOLE_WALL: rc = launch_cartpole_wall(h, mode, a); break;
    case MI_MODEL_SYNTH36: rc = launch_synth36(h, mode, a); break;
    case MI_MODEL_PLANAR_QUAD: rc = launch_planar_quad(h, mode, a); break;
    case MI_MODEL_QUAD3D: rc = launch_quad3d(h, mode, a); break;
    case MI_MODEL_ARM27: rc = launch_arm27(h, mode, a); break;
    case MI_MODEL_ARM27C: rc = launch_arm27c(h, mode, a); break;
    default:
      if (const PluginSlot* ps = plugin_of(h->d.model_id)) { rc = ps->p.launch(h, mode, &a); break; }
      return MI_ILQR_E_UNSUPPORTED;
  }
  return rc;
}

// When the state is lazily-zero (cold) but a kernel is about to write only part
// of it, materialize the zeros first.
int materialize_zero_state(mi_ilqr* h) {
  if (!h->cold) return MI_ILQR_OK;
  const size_t n = h->n, m = h->m, N = h->N, B = h->B;
  HIPCHK(hipMemsetAsync(h->x_bar, 0, B * n * N * 8, h->stream));
  HIPCHK(hipMemsetAsync(h->K, 0, B * m * n * (N - 1) * 8, h->stream));
  HIPCHK(hipMemsetAsync(h->kappa, 0, B * m * (N - 1) * 8, h->stream));
  HIPCHK(hipMemsetAsync(h->dV, 0, B * (N - 1) * 8, h->stream));
  HIPCHK(hipMemsetAsync(h->fx, 0, B * n * n * (N - 1) * 8, h->stream));
  HIPCHK(hipMemsetAsync(h->fu, 0, B * n * m * (N - 1) * 8, h->stream));
  h->cold = false;
  return MI_ILQR_OK;
}

int materialize_u(mi_ilqr* h) {
  if (h->u_zero && !h->u_pending) HIPCHK(hipMemsetAsync(h->u_bar, 0, (size_t)h->B * h->m * (h->N - 1) * 8, h->stream));
  h->u_zero = false;
  if (!h->u_pending) return MI_ILQR_OK;
  HIPCHK(hipMemcpyAsync(h->u_bar, h->u_guess, (size_t)h->B * h->m * (h->N - 1) * 8, h->host_inputs ? hipMemcpyDefault : hipMemcpyDeviceToDevice, h->stream));
  h->u_pending = false;
  return MI_ILQR_OK;
}

__global__ void mpc_shift_kernel(const double* x_bar, const double* u_bar, double* x0, double* u_guess,
                                 int B, int n, int m, int N, int r) {
  const int b = blockIdx.x;
  if (b >= B) return;
  for (int i = threadIdx.x; i < n; i += blockDim.x) x0[(size_t)b * n + i] = x_bar[((size_t)b * n + i) * N + r];
  const int M1 = N - 1;
  for (int idx = threadIdx.x; idx < m * M1; idx += blockDim.x) {
    const int k = idx / M1, t = idx - k * M1;
    const int src = (t + r < M1) ? t + r : M1 - 1;
    u_guess[((size_t)b * m + k) * M1 + t] = u_bar[((size_t)b * m + k) * M1 + src];
  }
}

__global__ void mpc_shift_kernel_tm(const double* x_bar, const double* u_bar, double* x0, double* u_guess,
                                    int B, int n, int m, int N, int r) {
  const int b = blockIdx.x;
  if (b >= B) return;
  for (int i = threadIdx.x; i < n; i += blockDim.x) x0[(size_t)b * n + i] = x_bar[((size_t)b * N + r) * n + i];
  const int M1 = N - 1;
  for (int idx = threadIdx.x; idx < m * M1; idx += blockDim.x) {
    const int t = idx / m, k = idx - t * m;
    const int src = (t + r < M1) ? t + r : M1 - 1;
    u_guess[((size_t)b * M1 + t) * m + k] = u_bar[((size_t)b * M1 + src) * m + k];
  }
}

__global__ void mpc_shift_kernel_bm(const double* x_bar, const double* u_bar, double* x0, double* u_guess,
                                    int B, int n, int m, int N, int r) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  for (int i = 0; i < n; ++i) x0[(size_t)b * n + i] = x_bar[((size_t)r * n + i) * B + b];
  const int M1 = N - 1;
  for (int t = 0; t < M1; ++t) {
    const int src = (t + r < M1) ? t + r : M1 - 1;
    for (int k = 0; k < m; ++k) u_guess[((size_t)t * m + k) * B + b] = u_bar[((size_t)src * m + k) * B + b];
  }
}

// Layouts of a (B, rows, len) trajectory array: TL time-last [b][row][t] (the reference's, SURVEY F5 - what
// crosses the C ABI); TM time-major [b][t][row] (workgroup-per-problem kernels); BM batch-minor
// [t][row][b] (lane-per-problem kernels).  The conversions run on the DEVICE, between the field and a
// staging buffer; the host copy is then one linear transfer.
enum { LAYOUT_TL = 0, LAYOUT_TM = 1, LAYOUT_BM = 2 };

// TL <-> TM: a per-problem (rows x len) transpose; both sides of a problem fit the caches, a gather is enough.
__global__ void __launch_bounds__(256) relayout_tm_kernel(const double* __restrict__ src, double* __restrict__ dst,
                                                            int rows, int len, int to_tm) {
  const size_t base = (size_t)blockIdx.x * rows * len;
  for (int e = threadIdx.x; e < rows * len; e += 256) {
    int r, t;
    if (to_tm) { t = e / rows; r = e - t * rows; dst[base + e] = src[base + (size_t)r * len + t]; }
    else { r = e / len; t = e - r * len; dst[base + e] = src[base + (size_t)t * rows + r]; }
  }
}

// TL <-> BM: for every row r a (B x len) <-> (len x B) transpose with pitches; 32 x 32 tiles through LDS
// so that both the reads and the writes are coalesced.  grid = (ceil(len/32), ceil(B/32), rows).
__global__ void __launch_bounds__(256) relayout_bm_kernel(const double* __restrict__ src, double* __restrict__ dst,
                                                            int B, int rows, int len, int to_bm) {
  __shared__ double tile[32][33];
  const int r = blockIdx.z, t0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8 threads
  if (to_bm) {
    for (int j = ty; j < 32; j += 8) {                             // read TL: t fastest
      const int b = b0 + j, t = t0 + tx;
      if (b < B && t < len) tile[j][tx] = src[((size_t)b * rows + r) * len + t];
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {                             // write BM: b fastest
      const int t = t0 + j, b = b0 + tx;
      if (b < B && t < len) dst[((size_t)t * rows + r) * B + b] = tile[tx][j];
    }
  } else {
    for (int j = ty; j < 32; j += 8) {                             // read BM: b fastest
      const int t = t0 + j, b = b0 + tx;
      if (b < B && t < len) tile[j][tx] = src[((size_t)t * rows + r) * B + b];
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {                             // write TL: t fastest
      const int b = b0 + j, t = t0 + tx;
      if (b < B && t < len) dst[((size_t)b * rows + r) * len + t] = tile[tx][j];
    }
  }
}

int ensure_scratch(mi_ilqr* h, size_t bytes) {
  if (h->scratch_bytes >= bytes) return MI_ILQR_OK;
  HIPCHK(hipStreamSynchronize(h->stream));
  if (h->scratch) HIPCHK(hipFree(h->scratch));
  h->scratch = nullptr; h->scratch_bytes = 0;
  HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->scratch), bytes));
  h->scratch_bytes = bytes;
  return MI_ILQR_OK;
}

// Convert between the handle's kernel layout and time-last, on the handle's stream.
int relayout(mi_ilqr* h, const double* src, double* dst, int rows, int len, bool to_kernel_layout) {
  if (h->batch_minor) {
    const dim3 grid((len + 31) / 32, (h->B + 31) / 32, rows);
    hipLaunchKernelGGL(relayout_bm_kernel, grid, dim3(256), 0, h->stream, src, dst, h->B, rows, len, to_kernel_layout ? 1 : 0);
  } else {
    hipLaunchKernelGGL(relayout_tm_kernel, dim3(h->B), dim3(256), 0, h->stream, src, dst, rows, len, to_kernel_layout ? 1 : 0);
  }
  HIPCHK(hipGetLastError());
  return MI_ILQR_OK;
}

// One control sequence (m, len), time last, written for every problem in the handle's layout:
// layout 0 = [b][k][t] (wave-per-problem), 1 = [b][t][k] (workgroup-per-problem), 2 = [t][k][b] (lane-per-problem).
__global__ void __launch_bounds__(256) broadcast_u_kernel(const double* __restrict__ src, double* __restrict__ dst,
                                                            int B, int m, int len, int layout) {
  const size_t total = (size_t)B * m * len;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    int k, t;
    if (layout == 0) { const size_t r = e % ((size_t)m * len); k = (int)(r / len); t = (int)(r - (size_t)k * len); }
    else if (layout == 1) { const size_t r = e % ((size_t)m * len); t = (int)(r / m); k = (int)(r - (size_t)t * m); }
    else { const size_t r = e / B; t = (int)(r / m); k = (int)(r - (size_t)t * m); }
    dst[e] = src[(size_t)k * len + t];
  }
}

// rows of the (rows,len) time-last view of a double field; 0 = not a trajectory array
int traj_rows(const mi_ilqr* h, int which, int* len) {
  const int n = h->n, m = h->m, N = h->N;
  switch (which) {
    case MI_F_X_BAR: case MI_F_X_TRIAL: *len = N; return n;
    case MI_F_U_BAR: case MI_F_U_TRIAL: case MI_F_KAPPA: *len = N - 1; return m;
    case MI_F_K: *len = N - 1; return m * n;
    case MI_F_FX: *len = N - 1; return n * n;
    case MI_F_FU: *len = N - 1; return n * m;
    case MI_F_DV: *len = N - 1; return 1;
  }
  *len = 0;
  return 0;
}

// Aggregate per-problem results on the device so a blocking solve costs ONE small host read
// (pinned, device-mapped) instead of four D2H copies.

// One workgroup per ring slot: blockIdx.x-th of `first, first+1, ...` (mod ring); the arrays are the slot-0 bases.
__global__ void __launch_bounds__(256) stats_kernel(const int32_t* iters, const int32_t* status, const int32_t* ls,
                                                    const double* cost, int B, DevStats* out, int first, int ring) {
  {
    const int slot = (first + (int)blockIdx.x) % ring;
    const size_t o = (size_t)slot * B;
    iters += o; status += o; ls += o; cost += o; out += slot;
  }
  __shared__ long long s_it[256], s_ls[256];
  __shared__ int s_c[256], s_m[256], s_f[256], s_mx[256], s_bi[256], s_x[256], s_p[256];
  __shared__ double s_bc[256];
  const int tid = threadIdx.x;
  long long it = 0, l = 0; int c = 0, m = 0, f = 0, mx = 0, bi = -1, xi = 0, npd = 0; double bc = INFINITY;
  for (int b = tid; b < B; b += 256) {
    it += iters[b]; l += ls[b];
    if (iters[b] > mx) mx = iters[b];
    const int st = status[b] & ~MI_STATUS_FLAG_INDEFINITE;      // (the flag rides on the solve's own outcome; counted in n_not_pd as well)
    if (status[b] & MI_STATUS_FLAG_INDEFINITE) npd++;
    if (st == MI_STATUS_CONVERGED) { c++; if (cost[b] < bc) { bc = cost[b]; bi = b; } }
    else if (st == MI_STATUS_MAX_ITERS) m++;
    else if (st == MI_STATUS_INTERNAL) xi++;
    else if (st == MI_STATUS_NOT_PD) npd++;
    else f++;
  }
  s_x[tid] = xi; s_p[tid] = npd;
  s_it[tid] = it; s_ls[tid] = l; s_c[tid] = c; s_m[tid] = m; s_f[tid] = f; s_mx[tid] = mx; s_bi[tid] = bi; s_bc[tid] = bc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) {
      s_it[tid] += s_it[tid + o]; s_ls[tid] += s_ls[tid + o]; s_c[tid] += s_c[tid + o]; s_m[tid] += s_m[tid + o]; s_f[tid] += s_f[tid + o];
      s_x[tid] += s_x[tid + o]; s_p[tid] += s_p[tid + o];
      if (s_mx[tid + o] > s_mx[tid]) s_mx[tid] = s_mx[tid + o];
      // ties resolve to the lower problem index, like a sequential scan
      if (s_bc[tid + o] < s_bc[tid] || (s_bc[tid + o] == s_bc[tid] && s_bi[tid + o] >= 0 && (s_bi[tid] < 0 || s_bi[tid + o] < s_bi[tid]))) { s_bc[tid] = s_bc[tid + o]; s_bi[tid] = s_bi[tid + o]; }
    }
    __syncthreads();
  }
  if (tid == 0) {
    out->total_iters = s_it[0]; out->total_ls = s_ls[0]; out->n_conv = s_c[0]; out->n_max = s_m[0]; out->n_fail = s_f[0];
    out->max_iters_seen = s_mx[0]; out->best_index = s_bi[0]; out->best_cost = s_bc[0]; out->n_internal = s_x[0]; out->n_not_pd = s_p[0];
  }
}

// DevStats records of every enqueued solve that has none yet: ONE launch, a workgroup per pending ring slot.
int reduce_pending_stats(mi_ilqr* h) {
  long long first = h->stats_done;
  if (h->seq - first > mi_ilqr::kStatsRing) first = h->seq - mi_ilqr::kStatsRing;    // older slots were overwritten
  const int count = (int)(h->seq - first);
  if (count > 0) {
    hipLaunchKernelGGL(stats_kernel, dim3(count), dim3(256), 0, h->stream, h->iters_ring, h->status_ring, h->ls_ring, h->cost_ring,
                       h->B, h->d_ring, (int)(first % mi_ilqr::kStatsRing), (int)mi_ilqr::kStatsRing);
    HIPCHK(hipGetLastError());
  }
  h->stats_done = h->seq;
  return MI_ILQR_OK;
}

// Host -> device copy of a caller's (pageable) buffer, ordered on the handle's stream.  Small inputs go through
// the handle's page-locked staging block and an ASYNCHRONOUS copy: the call returns after a host memcpy, the
// kernels that follow on the stream see the data (a blocking hipMemcpy costs ~15-20 us each whatever its size).
// Large ones keep the runtime's own pipelined staging.
int stage_h2d(mi_ilqr* h, void* dst, const void* src, size_t bytes) {
  constexpr size_t kSmall = 256 * 1024;
  if (bytes > kSmall) {
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return MI_ILQR_OK;
  }
  if (!h->pin_in) {
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&h->pin_in), 4 * kSmall, hipHostMallocDefault));
    HIPCHK(hipEventCreateWithFlags(&h->pin_ev, hipEventDisableTiming));
    h->pin_off = 0;
  }
  if (h->pin_off + bytes > 4 * kSmall) {          // the block is used as a ring; wrap once the copies in flight are done
    HIPCHK(hipEventSynchronize(h->pin_ev));
    h->pin_off = 0;
  }
  char* stage = h->pin_in + h->pin_off;
  std::memcpy(stage, src, bytes);
  HIPCHK(hipMemcpyAsync(dst, stage, bytes, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipEventRecord(h->pin_ev, h->stream));
  h->pin_off += (bytes + 255) & ~(size_t)255;
  return MI_ILQR_OK;
}

struct Field { void* ptr; size_t bytes; bool is_int; };

Field field_of(mi_ilqr* h, int which) {
  const size_t n = h->n, m = h->m, N = h->N, B = h->B;
  switch (which) {
    case MI_F_X_BAR: return {h->x_bar, B * n * N * 8, false};
    case MI_F_U_BAR: return {h->u_bar, B * m * (N - 1) * 8, false};
    case MI_F_K: return {h->K, B * m * n * (N - 1) * 8, false};
    case MI_F_KAPPA: return {h->kappa, B * m * (N - 1) * 8, false};
    case MI_F_DV: return {h->dV, B * (N - 1) * 8, false};
    case MI_F_FX: return {h->fx, B * n * n * (N - 1) * 8, false};
    case MI_F_FU: return {h->fu, B * n * m * (N - 1) * 8, false};
    case MI_F_COST: return {h->cost, B * 8, false};
    case MI_F_X0: return {h->x0, B * n * 8, false};
    case MI_F_HIST: return {h->hist, B * (size_t)h->d.hist_cap * 4 * 8, false};
    case MI_F_ITER_CYCLES: return {h->iter_cyc, B * (size_t)h->d.hist_cap * 4 * 8, false};
    case MI_F_X_TRIAL: return {h->x_trial, B * n * N * 8, false};
    case MI_F_U_TRIAL: return {h->u_trial, B * m * (N - 1) * 8, false};
    case MI_F_TRIAL_COST: return {h->trial_cost, B * 2 * 8, false};
    case MI_I_ITERS: return {h->iters, B * 4, true};
    case MI_I_STATUS: return {h->status, B * 4, true};
    case MI_I_LS_TRIALS: return {h->ls_trials, B * 4, true};
    case MI_I_KP_COUNT: return {h->kp_count, B * 4, true};
    case MI_I_KP_LIST: return {h->kp_list, B * (N - 1) * 4, true};
    case MI_I64_STAGE_CYCLES: return {h->prof, B * 4 * 8, true};
    case MI_I64_CLUSTER_WORDS: return {h->cluster_sync, B * kSyncWords * 8, true};
  }
  return {nullptr, 0, false};
}

// The per-iteration records (MI_F_HIST, MI_F_ITER_CYCLES: hist_cap rows of 4 doubles per problem) may be read in part: any whole
// number of LEADING rows of the first problem - what a caller with one problem and a long log capacity wants after a short solve
// (the drop-in class keeps 4096 rows and reads the 64 first with the solve, the rest only when a solve took more iterations).
bool prefix_ok(int which, size_t bytes, size_t field_bytes) {
  return (which == MI_F_HIST || which == MI_F_ITER_CYCLES) && bytes > 0 && bytes < field_bytes && bytes % 32 == 0;
}

bool is_state_field(int which) {
  return which == MI_F_X_BAR || which == MI_F_K || which == MI_F_KAPPA || which == MI_F_DV || which == MI_F_FX || which == MI_F_FU;
}

double bytes_per_iteration(int n, int m, int N, int ls) {
  const double roll = 2.0 * n * N + (3.0 * m + (double)m * n) * (N - 1) + n + 1;
  const double deriv = (N - 1.0) * ((n + m) + ((double)n * n + (double)n * m));
  const double back = (N - 1.0) * ((n + m) + ((double)n * n + (double)n * m) + ((double)m * n + m + 1)) + n;
  return 8.0 * (ls * roll + deriv + back);
}

}  // namespace

extern "C" {

int mi_ilqr_abi_version(void) { return MI_ILQR_ABI_VERSION; }
void mi_ilqr_struct_sizes(int32_t* desc_bytes, int32_t* stats_bytes, int32_t* plugin_bytes) {
  if (desc_bytes) *desc_bytes = (int32_t)sizeof(mi_ilqr_desc);
  if (stats_bytes) *stats_bytes = (int32_t)sizeof(mi_ilqr_stats);
  if (plugin_bytes) *plugin_bytes = (int32_t)sizeof(mi_ilqr_model_plugin);
}

const char* mi_ilqr_strerror(int code) {
  switch (code) {
    case MI_ILQR_OK: return "ok";
    case MI_ILQR_E_BAD_SHAPE: return "bad shape";
    case MI_ILQR_E_BAD_METHOD: return "unknown interpolation method";
    case MI_ILQR_E_LINESEARCH: return "linesearch failed";
    case MI_ILQR_E_HIP: return "HIP runtime error";
    case MI_ILQR_E_NO_DEVICE: return "no usable gfx950 device (there is no CPU fallback)";
    case MI_ILQR_E_BAD_ARG: return "bad argument";
    case MI_ILQR_E_UNSUPPORTED: return "model/size/cost combination not supported by any kernel";
    case MI_ILQR_E_RCCL: return "RCCL error (librccl missing, or a communicator / collective call failed)";
  }
  return "unknown error";
}

int mi_ilqr_register_model(const mi_ilqr_model_plugin* p, int32_t* model_id_out) {
  if (!p || !model_id_out || !p->launch || !p->lds_bytes) return MI_ILQR_E_BAD_ARG;
  if (p->abi_version != MI_ILQR_ABI_VERSION || p->kernel_args_bytes != (int32_t)sizeof(KArgs) || p->handle_bytes != (int32_t)sizeof(mi_ilqr)) {
    std::fprintf(stderr, "mi_ilqr_register_model: plugin built against other headers (ABI %d, %d-byte kernel arguments, %d-byte handle; the library: %d, %d, %d)\n",
                 p->abi_version, p->kernel_args_bytes, p->handle_bytes, MI_ILQR_ABI_VERSION, (int)sizeof(KArgs), (int)sizeof(mi_ilqr));
    return MI_ILQR_E_BAD_ARG;
  }
  if (p->n < 1 || p->m < 1 || p->n > kMaxStateDim || p->n_params < 0 || p->n_params > MI_ILQR_MAX_PARAMS) return MI_ILQR_E_BAD_SHAPE;
  if (p->m_user < 0 || p->m_user > p->m) return MI_ILQR_E_BAD_SHAPE;     // (0: no padding controls)
  // family 0: wave-per-problem kernels (m <= 2); family 1: workgroup-per-problem kernels - n <= 32 with any m <= 16
  // (mid_backward), 32 < n <= 40 with m % 4 == 0 and 2 m <= n (large_backward's wave roles)
  if (p->family == 0 ? p->m > 2
                     : (p->family != 1 || p->m > 16 || (p->n > 32 && (2 * p->m > p->n || p->m % 4 != 0)))) return MI_ILQR_E_UNSUPPORTED;
  std::lock_guard<std::mutex> lock(g_plugins_mutex);                     // (readers: plugin_of, lock-free)
  for (int s_ = 0; s_ < MI_ILQR_MAX_PLUGINS; ++s_) {
    if (g_plugins[s_].used.load(std::memory_order_relaxed)) continue;
    PluginSlot& ps = g_plugins[s_];
    ps.p = *p;
    ps.info.n = p->n; ps.info.m = p->m; ps.info.n_params = p->n_params;
    for (int i = 0; i < MI_ILQR_MAX_PARAMS; ++i) ps.info.defaults[i] = p->default_params[i];
    ps.used.store(true, std::memory_order_release);
    *model_id_out = MI_MODEL_PLUGIN_BASE + s_;
    return MI_ILQR_OK;
  }
  return MI_ILQR_E_UNSUPPORTED;                                       // registry full
}

int mi_ilqr_model_info(int model_id, int32_t* n, int32_t* m, int32_t* n_params, double* default_params) {
  const ModelInfo* mi_ = model_info(model_id);
  if (!mi_) return MI_ILQR_E_BAD_ARG;
  if (n) *n = mi_->n;
  if (m) *m = mi_->m;
  if (n_params) *n_params = mi_->n_params;
  if (default_params) for (int i = 0; i < MI_ILQR_MAX_PARAMS; ++i) default_params[i] = i < mi_->n_params ? mi_->defaults[i] : 0.0;
  return MI_ILQR_OK;
}

double mi_ilqr_bytes_per_iteration(int32_t n, int32_t m, int32_t N, int32_t ls) { return bytes_per_iteration(n, m, N, ls); }

size_t mi_ilqr_lds_bytes(const mi_ilqr_desc* d) {
  if (!d) return 0;
  const size_t s_ = small_lds_bytes(d->model_id, d->N);
  return s_ ? s_ : large_lds(d->model_id, d->N);
}

int mi_ilqr_create(const mi_ilqr_desc* desc, mi_ilqr_t** out) {
  if (!desc || !out) return MI_ILQR_E_BAD_ARG;
  *out = nullptr;
  const ModelInfo* info = model_info(desc->model_id);
  if (!info) return MI_ILQR_E_BAD_ARG;
  if (desc->n != info->n || desc->m != info->m) return MI_ILQR_E_BAD_SHAPE;
  if (desc->N < 2 || desc->B < 1) return MI_ILQR_E_BAD_SHAPE;
  if (desc->keypoint_method < MI_KP_SET_INTERVAL || desc->keypoint_method > MI_KP_ITERATIVE_ERROR) return MI_ILQR_E_BAD_METHOD;
  if (desc->minN < 1) return MI_ILQR_E_BAD_ARG;
  if (desc->jacobian_mode != MI_JAC_FD_CENTRAL && desc->jacobian_mode != MI_JAC_AUTODIFF) return MI_ILQR_E_BAD_ARG;
  if (desc->jacobian_mode == MI_JAC_FD_CENTRAL && !(desc->fd_step > 0.0)) return MI_ILQR_E_BAD_ARG;

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return MI_ILQR_E_NO_DEVICE;
  if (desc->device_id < 0 || desc->device_id >= ndev) return MI_ILQR_E_NO_DEVICE;
  HIPCHK(hipSetDevice(desc->device_id));

  size_t lds = small_lds_bytes(desc->model_id, desc->N);
  bool large = false;
  int n_store = 1;
  if (lds == 0) { lds = large_lds(desc->model_id, desc->N); large = true; }
  if (lds == 0) return MI_ILQR_E_UNSUPPORTED;
  bool batch_minor = false;
  if (desc->kernel_mode < MI_KERNEL_AUTO || desc->kernel_mode > MI_KERNEL_THROUGHPUT) return MI_ILQR_E_BAD_ARG;
  if (desc->on_indefinite != 0 && desc->on_indefinite != 1) return MI_ILQR_E_BAD_ARG;
  {
    // (every key-point configuration since round 4: the KP instantiation of the lane-per-problem kernels)
    // plugin models: family 0 with n <= 6 (their units instantiate the lane-per-problem kernels as well - per-lane register
    // arrays of n x n doubles set the limit)
    const PluginSlot* const plug = plugin_of(desc->model_id);
    const bool can = !large && (!plug || (plug->p.family == 0 && plug->p.n <= kMaxBatchPluginN));
    if (desc->kernel_mode == MI_KERNEL_THROUGHPUT && !can) return MI_ILQR_E_UNSUPPORTED;
    // n = 2 within the time-parallel passes' horizon: the wave-per-problem kernel is the faster one at
    // every batch size (B = 65536: 68 M vs 42 M it/s, profiles/r01n_c2_modes_batch_sweep.txt)
    const bool time_parallel = info->n == 2 && info->m == 1 && desc->N - 1 <= 256;
    batch_minor = can && (desc->kernel_mode == MI_KERNEL_THROUGHPUT ||
                          (desc->kernel_mode == MI_KERNEL_AUTO && desc->B >= 8192 && !time_parallel));
    // horizons whose per-problem state exceeds the 160 KB of LDS (e.g. acrobot.py's literal N = 750) are
    // served by the HBM-streaming kernel, which has no such limit
    if (!batch_minor && lds > kMaxLds && can && desc->kernel_mode == MI_KERNEL_AUTO) batch_minor = true;
  }
  bool lxu_hbm = false;
  if (!batch_minor && lds > kMaxLds && large) {
    // the horizon's cost gradients do not fit beside the fixed block: keep them in HBM (ilqr_large.hpp: large_lds_bytes_hbm)
    const size_t l = large_lds_hbm(desc->model_id, desc->N);
    if (l != 0 && l <= kMaxLds) { lds = l; lxu_hbm = true; }
  }
  if (!batch_minor && lds > kMaxLds) return MI_ILQR_E_UNSUPPORTED;
  if (batch_minor) lds = 0;
  if (!large && !batch_minor && desc->beta <= 0.75) {
    // Coarse backtracking (beta <= 0.75) accepts one of the first few eps values: keep up to 6
    // candidate trajectories in LDS as long as that does not lower the problems-per-CU this
    // batch needs (256 CUs) — it removes the second rollout of a backtracking iteration.
    const size_t per_cu_needed = ((size_t)desc->B + 255) / 256;
    for (int ns = 6; ns > 1; --ns) {
      const size_t l = small_lds_bytes(desc->model_id, desc->N, ns);
      if (l <= kMaxLds && kMaxLds / l >= per_cu_needed) { n_store = ns; lds = l; break; }
    }
  }

  mi_ilqr* h = new (std::nothrow) mi_ilqr();
  if (!h) return MI_ILQR_E_BAD_ARG;
  h->d = *desc;
  if (h->d.max_iters <= 0) h->d.max_iters = 1000;
  if (h->d.hist_cap <= 0) h->d.hist_cap = 64;
  h->n = desc->n; h->m = desc->m; h->N = desc->N; h->B = desc->B;
  h->lds = lds;
  h->large = large;
  h->n_store = n_store;
  h->batch_minor = batch_minor;
  const size_t n = h->n, m = h->m, N = h->N, B = h->B;

#define ALLOC(p, count, T)                                             \
  do {                                                                 \
    if (hipMalloc(reinterpret_cast<void**>(&(p)), (count) * sizeof(T)) != hipSuccess) { mi_ilqr_destroy(h); return MI_ILQR_E_HIP; } \
    if (hipMemset((p), 0, (count) * sizeof(T)) != hipSuccess) { mi_ilqr_destroy(h); return MI_ILQR_E_HIP; } \
  } while (0)
  ALLOC(h->x_bar, B * n * N, double);
  ALLOC(h->u_bar, B * m * (N - 1), double);
  ALLOC(h->K, B * m * n * (N - 1), double);
  ALLOC(h->kappa, B * m * (N - 1), double);
  ALLOC(h->dV, B * (N - 1), double);
  ALLOC(h->fx, B * n * n * (N - 1), double);
  ALLOC(h->fu, B * n * m * (N - 1), double);
  const bool host_rec = B <= 4;              // (host.hpp: host_records)
  const bool host_in = host_rec && !large && !batch_minor;   // ... and the inputs x0, u_guess (wave-per-problem kernels: the boundary's layout)
  if (!host_in) {
    ALLOC(h->x0, B * n, double);
    ALLOC(h->u_guess, B * m * (N - 1), double);
  }
  ALLOC(h->cost_ring, B * mi_ilqr::kStatsRing, double);
  if (host_rec) {
    auto up = [](size_t v) { return (v + 63) & ~(size_t)63; };
    const size_t s_hist = up(B * (size_t)h->d.hist_cap * 4 * 8), s_prof = up(B * 4 * 8), s_ring = up(B * mi_ilqr::kStatsRing * 4);
    const size_t s_x0 = host_in ? up(B * n * 8) : 0, s_ug = host_in ? up(B * m * (N - 1) * 8) : 0;
    h->host_records_bytes = 2 * s_hist + s_prof + 2 * s_ring + s_x0 + s_ug;
    void* dv = nullptr;
    if (hipHostMalloc(reinterpret_cast<void**>(&h->host_records), h->host_records_bytes, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer(&dv, h->host_records, 0) != hipSuccess) { mi_ilqr_destroy(h); return MI_ILQR_E_HIP; }
    std::memset(h->host_records, 0, h->host_records_bytes);
    h->host_records_dev = static_cast<char*>(dv);
    char* q = h->host_records_dev;
    h->hist = reinterpret_cast<double*>(q); q += s_hist;
    h->iter_cyc = reinterpret_cast<double*>(q); q += s_hist;
    h->prof = reinterpret_cast<long long*>(q); q += s_prof;
    h->iters_ring = reinterpret_cast<int32_t*>(q); q += s_ring;
    h->status_ring = reinterpret_cast<int32_t*>(q); q += s_ring;
    if (host_in) { h->x0 = reinterpret_cast<double*>(q); q += s_x0; h->u_guess = reinterpret_cast<double*>(q); h->host_inputs = true; }
  } else {
    ALLOC(h->hist, B * (size_t)h->d.hist_cap * 4, double);
    ALLOC(h->iter_cyc, B * (size_t)h->d.hist_cap * 4, double);
  }
  ALLOC(h->x_trial, B * n * N, double);
  ALLOC(h->u_trial, B * m * (N - 1), double);
  ALLOC(h->trial_cost, B * 2, double);
  ALLOC(h->stage_in, B, double);
  ALLOC(h->costmat, 2 * n * n + m * m + n, double);
  if (!host_rec) {
    ALLOC(h->iters_ring, B * mi_ilqr::kStatsRing, int32_t);
    ALLOC(h->status_ring, B * mi_ilqr::kStatsRing, int32_t);
    ALLOC(h->prof, B * 4, long long);
  }
  ALLOC(h->ls_ring, B * mi_ilqr::kStatsRing, int32_t);
  ALLOC(h->kp_count, B, int32_t);
  ALLOC(h->kp_list, B * (N - 1), int32_t);
  ALLOC(h->done_counter, 1, int32_t);
  if (large) ALLOC(h->cluster_sync, B * kSyncWords, unsigned long long);
  if (lxu_hbm) ALLOC(h->lxu, B * (N - 1) * (n + m), double);
  if (large && n <= 32) {
    // mid-size kernels: four line-search candidates per pass - an optimization, so a batch too large for three more
    // trial buffers simply searches one candidate at a time (make_args: spec_policy = 0 without them)
    // (batches small enough for clusters: 31 slots - the candidates 1 .. 31 of a first pass that the leader and up to seven helper
    //  workgroups roll out together, ilqr_large.hpp: candidate groups)
    // ... as many as the cluster size make_args will pick asks for: 4 g - 1 (15 at B = 64 on 256 CUs), 3 when the launch will not be
    // clustered - plugin models unless MI_ILQR_CLUSTER forces it, batches beyond 64, fewer than two CUs per problem.  (Round 5
    // allocated 31 for every batch up to 64: 1 GB of HBM for nothing on an n = 32, N = 2000 plugin.)
    {
      int cus = 0;
      (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, desc->device_id);
      const char* fe = std::getenv("MI_ILQR_CLUSTER");
      const int forced = fe ? std::atoi(fe) : 0;
      int g = forced > 0 ? forced : (cus > 0 ? cus / B : 1);
      if (forced <= 0 && (plugin_of(desc->model_id) || B > 64)) g = 1;
      if (g > 8) g = 8;
      h->spec_slots = g > 1 ? 4 * g - 1 : 3;
    }
    for (;;) {
      const size_t xb = (size_t)h->spec_slots * B * n * N * sizeof(double), ub = (size_t)h->spec_slots * B * m * (N - 1) * sizeof(double);
      if (hipMalloc(reinterpret_cast<void**>(&h->x_spec), xb) == hipSuccess && hipMalloc(reinterpret_cast<void**>(&h->u_spec), ub) == hipSuccess) break;
      (void)hipGetLastError();
      if (h->x_spec) (void)hipFree(h->x_spec);
      h->x_spec = nullptr; h->u_spec = nullptr;
      if (h->spec_slots == 3) { h->spec_slots = 0; break; }
      h->spec_slots = 3;
    }
  }
  if (batch_minor && !(desc->keypoint_method == MI_KP_SET_INTERVAL && desc->minN == 1)) ALLOC(h->bm_scratch, B * 6 * (N - 1), int32_t);
#undef ALLOC
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, desc->device_id) == hipSuccess) h->n_cus = cus;
  }
  // defaults Q=I, R=I, Qf=I, x_nom=0 (ilqr.py:61-67)
  {
    std::vector<double> cm(2 * n * n + m * m + n, 0.0);
    for (size_t i = 0; i < n; ++i) { cm[i * n + i] = 1.0; cm[n * n + m * m + i * n + i] = 1.0; }
    for (size_t i = 0; i < m; ++i) cm[n * n + i * m + i] = 1.0;
    if (hipMemcpy(h->costmat, cm.data(), cm.size() * 8, hipMemcpyHostToDevice) != hipSuccess) { mi_ilqr_destroy(h); return MI_ILQR_E_HIP; }
    h->h_costmat = cm;
    h->costmat_synced = true;
  }
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
    mi_ilqr_destroy(h);
    return MI_ILQR_E_HIP;
  }
  for (int i = 0; i < mi_ilqr::kStatsRing; ++i) {
    if (hipEventCreate(&h->ring_ev0[i]) != hipSuccess || hipEventCreate(&h->ring_ev1[i]) != hipSuccess) {
      mi_ilqr_destroy(h);
      return MI_ILQR_E_HIP;
    }
  }
  if (hipHostMalloc(reinterpret_cast<void**>(&h->h_ring), sizeof(DevStats) * mi_ilqr::kStatsRing, hipHostMallocMapped) != hipSuccess ||
      hipHostGetDevicePointer(reinterpret_cast<void**>(&h->d_ring), h->h_ring, 0) != hipSuccess) {
    mi_ilqr_destroy(h);
    return MI_ILQR_E_HIP;
  }
  std::memset(h->h_ring, 0, sizeof(DevStats) * mi_ilqr::kStatsRing);
  select_stats_slot(h, 0);
  h->cold = true;
  h->u_pending = false;
  *out = h;
  return MI_ILQR_OK;
}

void mi_ilqr_destroy(mi_ilqr_t* h) {
  if (!h) return;
  (void)hipSetDevice(h->d.device_id);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->host_records) {                               // (the five record buffers point into this block)
    (void)hipHostFree(h->host_records);
    h->hist = h->iter_cyc = nullptr; h->prof = nullptr; h->iters_ring = h->status_ring = nullptr;
    if (h->host_inputs) h->x0 = h->u_guess = nullptr;
  }
  void* ptrs[] = {h->x_bar, h->u_bar, h->K, h->kappa, h->dV, h->fx, h->fu, h->x0, h->u_guess, h->cost_ring, h->hist, h->iter_cyc,
                  h->x_trial, h->u_trial, h->trial_cost, h->stage_in, h->costmat, h->iters_ring, h->status_ring, h->ls_ring,
                  h->kp_count, h->kp_list, h->prof, h->done_counter, h->cluster_sync, h->bm_scratch, h->x_spec, h->u_spec, h->lxu};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  if (h->h_ring) (void)hipHostFree(h->h_ring);
  if (h->mpc_log) (void)hipFree(h->mpc_log);
  if (h->scratch) (void)hipFree(h->scratch);
  for (int i = 0; i < mi_ilqr::kStatsRing; ++i) {
    if (h->ring_ev0[i]) (void)hipEventDestroy(h->ring_ev0[i]);
    if (h->ring_ev1[i]) (void)hipEventDestroy(h->ring_ev1[i]);
  }
  if (h->u_one) (void)hipFree(h->u_one);
  if (h->pin_in) (void)hipHostFree(h->pin_in);
  if (h->pin_ev) (void)hipEventDestroy(h->pin_ev);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

// Symmetric positive semi-definite (definite with `strict`)?  Cholesky of A + eps*I; the matrices are tiny.
static bool is_sym_psd(const double* A, int k, bool strict) {
  double scale = 0.0;
  for (int i = 0; i < k * k; ++i) { if (!(std::fabs(A[i]) < INFINITY)) return false; scale = std::fmax(scale, std::fabs(A[i])); }
  for (int i = 0; i < k; ++i)
    for (int j = 0; j < i; ++j) if (A[i * k + j] != A[j * k + i]) return false;
  std::vector<double> L((size_t)k * k, 0.0);
  const double eps = strict ? 0.0 : 1e-12 * scale * k;
  for (int j = 0; j < k; ++j) {
    double d = A[j * k + j] + eps;
    for (int q = 0; q < j; ++q) d -= L[j * k + q] * L[j * k + q];
    if (strict ? !(d > 0.0) : !(d >= 0.0)) return false;
    const double ld = std::sqrt(d);
    L[j * k + j] = ld;
    for (int i = j + 1; i < k; ++i) {
      double v = A[i * k + j];
      for (int q = 0; q < j; ++q) v -= L[i * k + q] * L[j * k + q];
      if (ld > 0.0) L[i * k + j] = v / ld;
      else if (std::fabs(v) > 1e-12 * scale) return false;      // zero pivot with a non-zero column: indefinite
    }
  }
  return true;
}

int mi_ilqr_set_cost(mi_ilqr_t* h, const double* Q, const double* R, const double* Qf, const double* x_nom) {
  if (!h) return MI_ILQR_E_BAD_ARG;
  HIPCHK(hipSetDevice(h->d.device_id));
  const size_t n = h->n, m = h->m;
  {
    // The time-parallel / matrix-core backward passes use Vxx = Vxx^T and (scan) PSD second-order terms; the
    // reference accepts ANY Q, R, Qf and never symmetrizes (ilqr.py:182,653-667).  Matrices outside that
    // class are served by the reference's recursion verbatim (wave- and lane-per-problem kernels), by the mid-size
    // matrix-core pass with every use of symmetry switched off (n <= 32), or refused (n >= 33).
    std::vector<double> cm = h->h_costmat;
    if (Q) std::memcpy(cm.data(), Q, n * n * 8);
    if (R) std::memcpy(cm.data() + n * n, R, m * m * 8);
    if (Qf) std::memcpy(cm.data() + n * n + m * m, Qf, n * n * 8);
    if (x_nom) std::memcpy(cm.data() + 2 * n * n + m * m, x_nom, n * 8);
    if (h->large && n > 32) {
      // The n >= 33 workgroup-per-problem kernels' matrix-core pass has no form without symmetry (the plain-arithmetic pass that
      // takes over - large_backward_asym - is ~4 x slower per step), and matrices built as A^T A or by float arithmetic are often
      // symmetric only to round-off: asymmetries up to a few ulps of the largest entry are averaged away here
      // (|A - A^T| <= 8 eps max|A|) so that they keep the fast pass; anything larger is followed as given.
      // (n <= 32: matrices are taken as given, like the reference does.)
      auto symmetrize = [](double* A, size_t k) {
        double scale = 0.0;
        for (size_t i = 0; i < k * k; ++i) scale = std::fmax(scale, std::fabs(A[i]));
        for (size_t i = 0; i < k; ++i)
          for (size_t j = 0; j < i; ++j) if (std::fabs(A[i * k + j] - A[j * k + i]) > 8 * 2.220446049250313e-16 * scale) return;
        for (size_t i = 0; i < k; ++i)
          for (size_t j = 0; j < i; ++j) A[i * k + j] = A[j * k + i] = 0.5 * (A[i * k + j] + A[j * k + i]);
      };
      symmetrize(cm.data(), n); symmetrize(cm.data() + n * n, m); symmetrize(cm.data() + n * n + m * m, n);
    }
    const bool regular = is_sym_psd(cm.data(), (int)n, false) && is_sym_psd(cm.data() + n * n + m * m, (int)n, false) &&
                         is_sym_psd(cm.data() + n * n, (int)m, true);
    bool asym = false;
    if (!regular && h->large) {
      // Definiteness is not required of the workgroup-per-problem kernels - they check it where it matters: every
      // Quu = 2R + fu^T Vxx fu of every backward pass (MI_STATUS_NOT_PD, or mi_ilqr_desc.on_indefinite = 1: inverted with partial
      // pivoting like the reference's np.linalg.inv, ilqr.py:655).  SYMMETRY: the mid-size kernels (n <= 32, mid_backward) follow
      // the reference on any matrices - lxx = 2Q and luu = 2R as given, lx = 2Qx - 2 x_nom^T Q, Vx' = Qx - Qu^T Quu^{-1} Qux with
      // the inverse of a Quu that is not symmetric, Vxx stored in full (ilqr.py:180-184,651-667); the n >= 33 kernels, whose
      // matrix-core pass mirrors tiles of the symmetric products, take a plain-arithmetic pass for such matrices (round 6:
      // large_backward_asym; they refused them before).
      auto finite = [](const double* A, size_t k) {
        for (size_t i = 0; i < k * k; ++i) if (!(std::fabs(A[i]) < INFINITY)) return false;
        return true;
      };
      auto symmetric = [](const double* A, size_t k) {
        for (size_t i = 0; i < k; ++i)
          for (size_t j = 0; j < i; ++j) if (A[i * k + j] != A[j * k + i]) return false;
        return true;
      };
      if (!(finite(cm.data(), n) && finite(cm.data() + n * n, m) && finite(cm.data() + n * n + m * m, n))) {
        std::fprintf(stderr, "mi_ilqr_set_cost: Q, R, Qf must be finite\n");
        return MI_ILQR_E_UNSUPPORTED;
      }
      asym = !(symmetric(cm.data(), n) && symmetric(cm.data() + n * n, m) && symmetric(cm.data() + n * n + m * m, n));
    }
    h->cost_asym = asym ? 1 : 0;
    {
      bool diag = true;
      for (size_t i = 0; i < n; ++i)
        for (size_t j = 0; j < n; ++j) if (i != j && cm[i * n + j] != 0.0) diag = false;
      h->q_diag = diag ? 1 : 0;
    }
    h->exact_backward = regular ? 0 : 1;
    // the device copy mirrors h_costmat: nothing to send when the caller repeats the matrices it set before
    // (Solve() pushes them on every call, like the reference reads its attributes on every call)
    if (h->costmat_synced && std::memcmp(cm.data(), h->h_costmat.data(), cm.size() * 8) == 0) return MI_ILQR_OK;
    h->h_costmat.swap(cm);
  }
  h->costmat_synced = false;                                  // (a failed copy must not leave the mirror believed)
  const int rc = stage_h2d(h, h->costmat, h->h_costmat.data(), h->h_costmat.size() * 8);    // Q | R | Qf | x_nom: one copy
  h->costmat_synced = rc == MI_ILQR_OK;
  return rc;
}

// tiny batches of the wave-per-problem kernels keep x0 / u_guess in page-locked host memory the kernels read directly (host.hpp:
// host_inputs): setting them is a memcpy once nothing on the stream can still be reading the old values - no copy engine, no
// broadcast kernel in front of the solve (C1: two ~5 us copies + one launch off the critical path of every Solve())
static int host_inputs_write(mi_ilqr* h, const double* x0, const double* u_guess, bool shared) {
  const hipError_t q = hipStreamQuery(h->stream);
  if (q == hipErrorNotReady) HIPCHK(hipStreamSynchronize(h->stream));
  else if (q != hipSuccess) return MI_ILQR_E_HIP;
  char* const base = h->host_records;
  if (x0) std::memcpy(base + (reinterpret_cast<char*>(h->x0) - h->host_records_dev), x0, (size_t)h->B * h->n * 8);
  if (u_guess) {
    const size_t one = (size_t)h->m * (h->N - 1) * 8;
    char* dst = base + (reinterpret_cast<char*>(h->u_guess) - h->host_records_dev);
    for (int b = 0; b < h->B; ++b) std::memcpy(dst + b * one, reinterpret_cast<const char*>(u_guess) + (shared ? 0 : b * one), one);
    h->u_pending = true;
    h->u_zero = false;
  }
  return MI_ILQR_OK;
}

int mi_ilqr_set_initial(mi_ilqr_t* h, const double* x0, const double* u_guess) {
  if (!h) return MI_ILQR_E_BAD_ARG;
  HIPCHK(hipSetDevice(h->d.device_id));
  if (h->host_inputs) return host_inputs_write(h, x0, u_guess, false);
  if (x0) { const int rc = stage_h2d(h, h->x0, x0, (size_t)h->B * h->n * 8); if (rc != MI_ILQR_OK) return rc; }
  if (u_guess) {
    const size_t cnt = (size_t)h->B * h->m * (h->N - 1);
    if ((h->large || h->batch_minor) && (h->m > 1 || h->batch_minor)) {
      int rc = ensure_scratch(h, cnt * 8);
      if (rc != MI_ILQR_OK) return rc;
      HIPCHK(hipStreamSynchronize(h->stream));
      HIPCHK(hipMemcpy(h->scratch, u_guess, cnt * 8, hipMemcpyHostToDevice));
      if ((rc = relayout(h, h->scratch, h->u_guess, h->m, h->N - 1, true)) != MI_ILQR_OK) return rc;
      HIPCHK(hipStreamSynchronize(h->stream));
    } else {
      const int rc = stage_h2d(h, h->u_guess, u_guess, cnt * 8);
      if (rc != MI_ILQR_OK) return rc;
    }
    h->u_pending = true;
    h->u_zero = false;
  }
  return MI_ILQR_OK;
}

int mi_ilqr_set_initial_shared(mi_ilqr_t* h, const double* x0, const double* u_guess_one) {
  if (!h) return MI_ILQR_E_BAD_ARG;
  HIPCHK(hipSetDevice(h->d.device_id));
  if (h->host_inputs) return host_inputs_write(h, x0, u_guess_one, true);
  if (x0) { const int rc = stage_h2d(h, h->x0, x0, (size_t)h->B * h->n * 8); if (rc != MI_ILQR_OK) return rc; }
  if (u_guess_one) {
    const size_t one = (size_t)h->m * (h->N - 1);
    if (!h->u_one) HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->u_one), one * 8));
    int rc = stage_h2d(h, h->u_one, u_guess_one, one * 8);
    if (rc != MI_ILQR_OK) return rc;
    const int layout = h->batch_minor ? 2 : (h->large ? 1 : 0);
    const size_t total = one * h->B;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(broadcast_u_kernel, dim3(blocks), dim3(256), 0, h->stream, h->u_one, h->u_guess, h->B, h->m, h->N - 1, layout);
    HIPCHK(hipGetLastError());
    h->u_pending = true;
    h->u_zero = false;
  }
  return MI_ILQR_OK;
}

int mi_ilqr_host_alloc(size_t bytes, void** out) {
  if (!out || bytes == 0) return MI_ILQR_E_BAD_ARG;
  *out = nullptr;
  HIPCHK(hipHostMalloc(out, bytes, hipHostMallocMapped));            // page-locked AND device-visible (result sinks)
  return MI_ILQR_OK;
}

int mi_ilqr_set_result_sink(mi_ilqr_t* h, double* x_bar_host, double* u_bar_host, double* cost_host) {
  if (!h) return MI_ILQR_E_BAD_ARG;
  if (h->large || h->batch_minor) return MI_ILQR_E_UNSUPPORTED;     // (their kernel layouts are not the boundary's)
  HIPCHK(hipSetDevice(h->d.device_id));
  HIPCHK(hipStreamSynchronize(h->stream));
  h->sink_x = h->sink_u = h->sink_cost = nullptr;
  if (!x_bar_host && !u_bar_host && !cost_host) return MI_ILQR_OK;
  if (!x_bar_host || !u_bar_host || !cost_host) return MI_ILQR_E_BAD_ARG;
  void *dx = nullptr, *du = nullptr, *dc = nullptr;
  if (hipHostGetDevicePointer(&dx, x_bar_host, 0) != hipSuccess || hipHostGetDevicePointer(&du, u_bar_host, 0) != hipSuccess ||
      hipHostGetDevicePointer(&dc, cost_host, 0) != hipSuccess) {
    (void)hipGetLastError();
    return MI_ILQR_E_BAD_ARG;                                        // not page-locked, device-visible host memory
  }
  h->sink_x = static_cast<double*>(dx); h->sink_u = static_cast<double*>(du); h->sink_cost = static_cast<double*>(dc);
  return MI_ILQR_OK;
}

int mi_ilqr_solve_into(mi_ilqr_t* h, double* x_bar, double* u_bar, double* cost, int32_t page_locked, int32_t n_extra,
                       const int32_t* which, void* const* dst, const size_t* bytes, mi_ilqr_stats* stats, int32_t* sink_used) {
  if (!h || !x_bar || !u_bar || !cost || n_extra < 0 || (n_extra > 0 && (!which || !dst || !bytes))) return MI_ILQR_E_BAD_ARG;
  if (sink_used) *sink_used = 0;
  int rc;
  bool sink = false;
  if (page_locked && !h->large && !h->batch_minor) {
    rc = mi_ilqr_set_result_sink(h, x_bar, u_bar, cost);
    if (rc == MI_ILQR_OK) sink = true;
    else if (rc != MI_ILQR_E_BAD_ARG) return rc;                      // (BAD_ARG: not page-locked after all - copy out instead)
  }
  auto clear = [&]() { h->sink_x = h->sink_u = h->sink_cost = nullptr; };
  if ((rc = mi_ilqr_solve_async(h)) != MI_ILQR_OK) { clear(); return rc; }
  if (!sink) {
    const int f3[3] = {MI_F_X_BAR, MI_F_U_BAR, MI_F_COST};
    void* const d3[3] = {x_bar, u_bar, cost};
    for (int i = 0; i < 3; ++i)
      if ((rc = mi_ilqr_get_async(h, f3[i], d3[i], field_of(h, f3[i]).bytes)) != MI_ILQR_OK) return rc;
  }
  // records the kernel writes straight into host memory (tiny batches, host.hpp: host_records): a memcpy after the synchronization
  auto host_side = [&](int w) -> const char* {
    if (!h->host_records || !(w == MI_F_HIST || w == MI_F_ITER_CYCLES || w == MI_I64_STAGE_CYCLES || w == MI_I_ITERS || w == MI_I_STATUS)) return nullptr;
    return h->host_records + (static_cast<const char*>(field_of(h, w).ptr) - h->host_records_dev);
  };
  for (int i = 0; i < n_extra; ++i) {
    if (host_side(which[i])) {
      const size_t fb = field_of(h, which[i]).bytes;
      if (bytes[i] != fb && !prefix_ok(which[i], bytes[i], fb)) { clear(); return MI_ILQR_E_BAD_SHAPE; }
      continue;
    }
    if ((rc = mi_ilqr_get_async(h, which[i], dst[i], bytes[i])) != MI_ILQR_OK) { clear(); return rc; }
  }
  rc = stats ? mi_ilqr_collect_stats_n(h, 1, stats) : mi_ilqr_synchronize(h);
  clear();                                                             // (the stream is idle: no later kernel of the handle writes the caller's arrays)
  if (rc == MI_ILQR_OK)
    for (int i = 0; i < n_extra; ++i)
      if (const char* src = host_side(which[i])) std::memcpy(dst[i], src, bytes[i]);
  if (sink_used) *sink_used = sink ? 1 : 0;
  return rc;
}

int mi_ilqr_host_free(void* p) {
  if (p) HIPCHK(hipHostFree(p));
  return MI_ILQR_OK;
}

int mi_ilqr_reset(mi_ilqr_t* h) {
  if (!h) return MI_ILQR_E_BAD_ARG;
  h->cold = true;          // zeros are materialized lazily (the kernels skip the HBM read)
  h->u_zero = true;        // a fresh reference object has u_bar = 0 until SetInitialGuess (ilqr.py:71,148-156)
  h->u_pending = false;
  return MI_ILQR_OK;
}

int mi_ilqr_rearm_initial_guess(mi_ilqr_t* h) {
  if (!h) return MI_ILQR_E_BAD_ARG;
  h->u_pending = true;     // next kernel takes u_bar from the resident u_guess again
  h->u_zero = false;
  return MI_ILQR_OK;
}

int mi_ilqr_synchronize(mi_ilqr_t* h) {
  if (!h) return MI_ILQR_E_BAD_ARG;
  HIPCHK(hipSetDevice(h->d.device_id));
  HIPCHK(hipStreamSynchronize(h->stream));
  return MI_ILQR_OK;
}

int mi_ilqr_solve_async(mi_ilqr_t* h) {
  if (!h) return MI_ILQR_E_BAD_ARG;
  const int slot = (int)(h->seq++ % mi_ilqr::kStatsRing);
  select_stats_slot(h, slot);
  // pipelined solves: the events ride on one launch in `time_every` (every other entry point times its launches)
  h->timed_launch = h->time_every > 0 && (h->seq_timed++ % h->time_every) == 0;
  h->in_async_solve = true;
  int rc = launch(h, MODE_SOLVE);
  h->in_async_solve = false;
  h->timed_launch = true;
  if (rc != MI_ILQR_OK) return rc;
  // (the batch statistics of this solve: by the kernel itself, or by reduce_pending_stats when somebody collects)
  if (stats_in_kernel(h) && h->stats_done == h->seq - 1) h->stats_done = h->seq;
  h->cold = false;
  h->u_pending = false;
  return MI_ILQR_OK;
}

static void fill_stats(mi_ilqr* h, int slot, mi_ilqr_stats* st) {
  std::memset(st, 0, sizeof(*st));
  const DevStats ds = h->h_ring[slot];   // written by stats_kernel, visible after a stream sync
  st->total_iters = ds.total_iters; st->total_ls_trials = ds.total_ls;
  st->n_converged = ds.n_conv; st->n_max_iters = ds.n_max; st->n_ls_failed = ds.n_fail;
  st->max_iters_seen = ds.max_iters_seen; st->best_cost = ds.best_cost; st->best_index = ds.best_index;
  st->n_internal = ds.n_internal; st->n_not_pd = ds.n_not_pd;
  float ms = 0.f;
  if (h->ring_timed[slot] && hipEventElapsedTime(&ms, h->ring_ev0[slot], h->ring_ev1[slot]) == hipSuccess) st->kernel_ms = ms;
  // bytes_iter is affine in ls: sum over iterations = ls_total*roll + iters*(deriv+back)
  const double per_ls = bytes_per_iteration(h->n, h->m, h->N, 1) - bytes_per_iteration(h->n, h->m, h->N, 0);
  const double fixed = bytes_per_iteration(h->n, h->m, h->N, 0);
  st->algorithmic_bytes = per_ls * (double)st->total_ls_trials + fixed * (double)st->total_iters;
}

int mi_ilqr_collect_stats(mi_ilqr_t* h, mi_ilqr_stats* st) {
  if (!h || !st) return MI_ILQR_E_BAD_ARG;
  HIPCHK(hipSetDevice(h->d.device_id));
  { const int rc = reduce_pending_stats(h); if (rc != MI_ILQR_OK) return rc; }
  HIPCHK(hipStreamSynchronize(h->stream));
  fill_stats(h, (int)(h->h_stats - h->h_ring), st);
  return MI_ILQR_OK;
}

int mi_ilqr_collect_stats_n(mi_ilqr_t* h, int32_t count, mi_ilqr_stats* st) {
  if (!h || !st || count < 1 || count > mi_ilqr::kStatsRing || count > h->seq) return MI_ILQR_E_BAD_ARG;
  HIPCHK(hipSetDevice(h->d.device_id));
  { const int rc = reduce_pending_stats(h); if (rc != MI_ILQR_OK) return rc; }
  HIPCHK(hipStreamSynchronize(h->stream));
  for (int i = 0; i < count; ++i) fill_stats(h, (int)((h->seq - count + i) % mi_ilqr::kStatsRing), st + i);
  return MI_ILQR_OK;
}

int mi_ilqr_solve(mi_ilqr_t* h, mi_ilqr_stats* stats) {
  int rc = mi_ilqr_solve_async(h);
  if (rc != MI_ILQR_OK) return rc;
  if (stats) return mi_ilqr_collect_stats(h, stats);
  return mi_ilqr_synchronize(h);
}

static int upload_stage_in(mi_ilqr* h, const double* v) {
  if (!v) return MI_ILQR_E_BAD_ARG;
  HIPCHK(hipSetDevice(h->d.device_id));
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipMemcpy(h->stage_in, v, (size_t)h->B * 8, hipMemcpyHostToDevice));
  return MI_ILQR_OK;
}

int mi_ilqr_rollout(mi_ilqr_t* h, const double* eps) {
  if (!h) return MI_ILQR_E_BAD_ARG;
  int rc = upload_stage_in(h, eps);
  if (rc != MI_ILQR_OK) return rc;
  rc = launch(h, MODE_ROLLOUT);          // reads only; cold/u_pending stay as they are
  if (rc != MI_ILQR_OK) return rc;
  return mi_ilqr_synchronize(h);
}

int mi_ilqr_forward(mi_ilqr_t* h, const double* L_last) {
  if (!h) return MI_ILQR_E_BAD_ARG;
  int rc = upload_stage_in(h, L_last);
  if (rc != MI_ILQR_OK) return rc;
  if ((rc = materialize_zero_state(h)) != MI_ILQR_OK) return rc;   // forward leaves K/kappa/dV untouched
  rc = launch(h, MODE_FORWARD);
  if (rc != MI_ILQR_OK) return rc;
  h->u_pending = false;
  return mi_ilqr_synchronize(h);
}

int mi_ilqr_linearize(mi_ilqr_t* h) {
  if (!h) return MI_ILQR_E_BAD_ARG;
  int rc;
  if ((rc = materialize_zero_state(h)) != MI_ILQR_OK) return rc;
  if ((rc = materialize_u(h)) != MI_ILQR_OK) return rc;
  rc = launch(h, MODE_LINEARIZE);
  if (rc != MI_ILQR_OK) return rc;
  return mi_ilqr_synchronize(h);
}

int mi_ilqr_backward(mi_ilqr_t* h) {
  if (!h) return MI_ILQR_E_BAD_ARG;
  int rc;
  if ((rc = materialize_zero_state(h)) != MI_ILQR_OK) return rc;
  if ((rc = materialize_u(h)) != MI_ILQR_OK) return rc;
  rc = launch(h, MODE_BACKWARD);
  if (rc != MI_ILQR_OK) return rc;
  return mi_ilqr_synchronize(h);
}

int mi_ilqr_mpc_shift(mi_ilqr_t* h, int32_t replan_steps) {
  if (!h) return MI_ILQR_E_BAD_ARG;
  if (replan_steps < 1 || replan_steps >= h->N - 1) return MI_ILQR_E_BAD_ARG;
  HIPCHK(hipSetDevice(h->d.device_id));
  int rc;
  if ((rc = materialize_zero_state(h)) != MI_ILQR_OK) return rc;
  if ((rc = materialize_u(h)) != MI_ILQR_OK) return rc;
  if (h->batch_minor)
    hipLaunchKernelGGL(mpc_shift_kernel_bm, dim3((h->B + 255) / 256), dim3(256), 0, h->stream, h->x_bar, h->u_bar, h->x0,
                       h->u_guess, h->B, h->n, h->m, h->N, (int)replan_steps);
  else if (h->large)
    hipLaunchKernelGGL(mpc_shift_kernel_tm, dim3(h->B), dim3(64), 0, h->stream, h->x_bar, h->u_bar, h->x0, h->u_guess,
                       h->B, h->n, h->m, h->N, (int)replan_steps);
  else
    hipLaunchKernelGGL(mpc_shift_kernel, dim3(h->B), dim3(64), 0, h->stream, h->x_bar, h->u_bar, h->x0, h->u_guess,
                       h->B, h->n, h->m, h->N, (int)replan_steps);
  HIPCHK(hipGetLastError());
  h->u_pending = true;
  return MI_ILQR_OK;
}

// host-loop form of the receding-horizon loop: the record the single-launch kernels write themselves (x0 | cost | iterations).
// Same policy as theirs: a problem whose re-solve fails (line search, internal, NOT_PD) gets that re-solve's row and none after
// it (`dead`: one flag per problem behind the log; the rows stay the zeros mi_ilqr_mpc_run fills the log with).
__global__ void __launch_bounds__(256) mpc_log_fill_kernel(const double* __restrict__ x0, const double* __restrict__ cost,
                                                           const int32_t* __restrict__ iters, const int32_t* __restrict__ status,
                                                           double* __restrict__ log, double* __restrict__ dead, int B, int n, int resolves, int r) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B || dead[b] != 0.0) return;
  double* lg = log + ((size_t)b * resolves + r) * (n + 2);
  for (int i = 0; i < n; ++i) lg[i] = x0[(size_t)b * n + i];
  lg[n] = cost[b];
  lg[n + 1] = (double)iters[b];
  if ((status[b] & ~MI_STATUS_FLAG_INDEFINITE) >= MI_STATUS_LINESEARCH_FAILED) dead[b] = 1.0;
}

int mi_ilqr_mpc_run(mi_ilqr_t* h, int32_t num_resolves, int32_t replan_steps, const double* target_step, mi_ilqr_stats* stats) {
  if (!h) return MI_ILQR_E_BAD_ARG;
  if (num_resolves < 1 || replan_steps < 1 || replan_steps >= h->N - 1) return MI_ILQR_E_BAD_ARG;
  HIPCHK(hipSetDevice(h->d.device_id));
  int rc;
  if (h->mpc_log_resolves < num_resolves) {
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->mpc_log) HIPCHK(hipFree(h->mpc_log));
    h->mpc_log = nullptr;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->mpc_log), ((size_t)h->B * num_resolves * (h->n + 2) + h->B) * 8));
    h->mpc_log_resolves = num_resolves;
  }
  // rows a problem never reaches (it failed in an earlier re-solve: both forms of the loop stop logging it there) read as zeros
  double* const mpc_dead = h->mpc_log + (size_t)h->B * h->mpc_log_resolves * (h->n + 2);
  HIPCHK(hipMemsetAsync(h->mpc_log, 0, ((size_t)h->B * h->mpc_log_resolves * (h->n + 2) + h->B) * 8, h->stream));
  h->mpc_resolves = num_resolves; h->mpc_replan = replan_steps;     // (up front: a failing re-solve must not leave the log's shape stale)
  const bool large_on_device = h->large && (size_t)h->m * (h->N - 1) <= 8 * (size_t)kLargeThreads;
  if ((h->large && !large_on_device) || h->batch_minor || (!h->large && (h->N > 512 || h->n > 8))) {
    // lane-per-problem path (and horizons the in-kernel shift does not cover): loop shift + solve on the host
    std::vector<double> xn(h->n);
    // x_nom from the host mirror: mi_ilqr_set_cost uploads asynchronously on the handle's stream, so a blocking
    // null-stream read of the device copy could still see the previous target
    if (target_step) std::memcpy(xn.data(), h->h_costmat.data() + 2 * (size_t)h->n * h->n + (size_t)h->m * h->m, h->n * 8);
    mi_ilqr_stats acc; std::memset(&acc, 0, sizeof(acc)); acc.best_cost = INFINITY; acc.best_index = -1;
    for (int r = 0; r < num_resolves; ++r) {
      if ((rc = mi_ilqr_mpc_shift(h, replan_steps)) != MI_ILQR_OK) return rc;
      if (target_step) {
        for (int i = 0; i < h->n; ++i) xn[i] += target_step[i];
        if ((rc = mi_ilqr_set_cost(h, nullptr, nullptr, nullptr, xn.data())) != MI_ILQR_OK) return rc;
      }
      mi_ilqr_stats st;
      if ((rc = mi_ilqr_solve(h, &st)) != MI_ILQR_OK) return rc;
      hipLaunchKernelGGL(mpc_log_fill_kernel, dim3((h->B + 255) / 256), dim3(256), 0, h->stream, h->x0, h->cost, h->iters, h->status, h->mpc_log,
                         mpc_dead, h->B, h->n, num_resolves, r);
      HIPCHK(hipGetLastError());
      acc.total_iters += st.total_iters; acc.total_ls_trials += st.total_ls_trials; acc.kernel_ms += st.kernel_ms;
      acc.algorithmic_bytes += st.algorithmic_bytes;
      acc.n_converged = st.n_converged; acc.n_max_iters = st.n_max_iters; acc.n_ls_failed = st.n_ls_failed; acc.n_internal = st.n_internal; acc.n_not_pd = st.n_not_pd;
      if (st.max_iters_seen > acc.max_iters_seen) acc.max_iters_seen = st.max_iters_seen;
      acc.best_cost = st.best_cost; acc.best_index = st.best_index;
    }
    h->mpc_resolves = num_resolves; h->mpc_replan = replan_steps;
    if (stats) *stats = acc;
    return MI_ILQR_OK;
  }
  if ((rc = materialize_zero_state(h)) != MI_ILQR_OK) return rc;
  if ((rc = materialize_u(h)) != MI_ILQR_OK) return rc;
  h->mpc_resolves = num_resolves; h->mpc_replan = replan_steps;
  for (int i = 0; i < kMaxStateDim; ++i) h->mpc_target_step[i] = (target_step && i < h->n) ? target_step[i] : 0.0;
  rc = launch(h, MODE_MPC);
  if (rc != MI_ILQR_OK) return rc;
  if (!stats_in_kernel(h)) {
    hipLaunchKernelGGL(stats_kernel, dim3(1), dim3(256), 0, h->stream, h->iters_ring, h->status_ring, h->ls_ring, h->cost_ring,
                       h->B, h->d_ring, h->cur_slot, (int)mi_ilqr::kStatsRing);
    HIPCHK(hipGetLastError());
  }
  if (target_step) {          // keep the handle's x_nom in step with what the kernel accumulated
    std::vector<double> xn(h->n);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(xn.data(), h->costmat + 2 * (size_t)h->n * h->n + (size_t)h->m * h->m, h->n * 8, hipMemcpyDeviceToHost));
    for (int i = 0; i < h->n; ++i) xn[i] += num_resolves * target_step[i];
    HIPCHK(hipMemcpy(h->costmat + 2 * (size_t)h->n * h->n + (size_t)h->m * h->m, xn.data(), h->n * 8, hipMemcpyHostToDevice));
    std::memcpy(h->h_costmat.data() + 2 * (size_t)h->n * h->n + (size_t)h->m * h->m, xn.data(), h->n * 8);   // (the host mirror too)
  }
  if (stats) return mi_ilqr_collect_stats(h, stats);
  return mi_ilqr_synchronize(h);
}

int mi_ilqr_get_mpc_log(mi_ilqr_t* h, double* dst, size_t bytes) {
  if (!h || !dst || !h->mpc_log) return MI_ILQR_E_BAD_ARG;
  if (bytes != (size_t)h->B * h->mpc_resolves * (h->n + 2) * 8) return MI_ILQR_E_BAD_SHAPE;
  HIPCHK(hipSetDevice(h->d.device_id));
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipMemcpy(dst, h->mpc_log, bytes, hipMemcpyDeviceToHost));
  return MI_ILQR_OK;
}

int mi_ilqr_get(mi_ilqr_t* h, int which, double* dst, size_t bytes) {
  if (!h || !dst) return MI_ILQR_E_BAD_ARG;
  Field f = field_of(h, which);
  if (!f.ptr || f.is_int) return MI_ILQR_E_BAD_ARG;
  if (bytes != f.bytes && !prefix_ok(which, bytes, f.bytes)) return MI_ILQR_E_BAD_SHAPE;
  HIPCHK(hipSetDevice(h->d.device_id));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (h->cold && is_state_field(which)) { std::memset(dst, 0, bytes); return MI_ILQR_OK; }
  if (which == MI_F_U_BAR && h->u_zero && !h->u_pending) { std::memset(dst, 0, bytes); return MI_ILQR_OK; }
  const double* src = (h->u_pending && which == MI_F_U_BAR) ? h->u_guess : static_cast<const double*>(f.ptr);
  int len = 0;
  const int rows = (h->large || h->batch_minor) ? traj_rows(h, which, &len) : 0;
  if (rows > 1 || (h->batch_minor && rows == 1)) {
    int rc = ensure_scratch(h, bytes);
    if (rc != MI_ILQR_OK) return rc;
    if ((rc = relayout(h, src, h->scratch, rows, len, false)) != MI_ILQR_OK) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(dst, h->scratch, bytes, hipMemcpyDeviceToHost));
    return MI_ILQR_OK;
  }
  HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyDefault));            // (tiny batches keep some fields in mapped host memory: host.hpp)
  return MI_ILQR_OK;
}

int mi_ilqr_get_async(mi_ilqr_t* h, int which, void* dst, size_t bytes) {
  if (!h || !dst) return MI_ILQR_E_BAD_ARG;
  Field f = field_of(h, which);
  if (!f.ptr) return MI_ILQR_E_BAD_ARG;
  if (bytes != f.bytes && !prefix_ok(which, bytes, f.bytes)) return MI_ILQR_E_BAD_SHAPE;
  HIPCHK(hipSetDevice(h->d.device_id));
  if (!f.is_int) {
    int len = 0;
    const int rows = (h->large || h->batch_minor) ? traj_rows(h, which, &len) : 0;
    // fields that need a layout conversion (or are known-zero) take the blocking path
    if (rows > 1 || (h->batch_minor && rows == 1) || (h->cold && is_state_field(which)) ||
        (which == MI_F_U_BAR && h->u_zero && !h->u_pending))
      return mi_ilqr_get(h, which, static_cast<double*>(dst), bytes);
  }
  const void* src = (!f.is_int && h->u_pending && which == MI_F_U_BAR) ? h->u_guess : f.ptr;
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, h->stream));
  return MI_ILQR_OK;
}

int mi_ilqr_get_int(mi_ilqr_t* h, int which, int32_t* dst, size_t bytes) {
  if (!h || !dst) return MI_ILQR_E_BAD_ARG;
  if (which == MI_I64_CLUSTER_WORDS && (!h->cluster_sync || !h->last_clustered)) {
    // the header's promise: zeros when the last solve / MPC launch was not clustered (the device words would be an older launch's),
    // and for handles of the other kernel families, which have no such words
    if (bytes != (size_t)h->B * MI_ILQR_CLUSTER_WORDS * 8) return MI_ILQR_E_BAD_SHAPE;
    HIPCHK(hipSetDevice(h->d.device_id));
    HIPCHK(hipStreamSynchronize(h->stream));
    std::memset(dst, 0, bytes);
    return MI_ILQR_OK;
  }
  Field f = field_of(h, which);
  if (!f.ptr || !f.is_int) return MI_ILQR_E_BAD_ARG;
  if (bytes != f.bytes) return MI_ILQR_E_BAD_SHAPE;
  HIPCHK(hipSetDevice(h->d.device_id));
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipMemcpy(dst, f.ptr, bytes, hipMemcpyDefault));
  return MI_ILQR_OK;
}

int mi_ilqr_set(mi_ilqr_t* h, int which, const double* src, size_t bytes) {
  if (!h || !src) return MI_ILQR_E_BAD_ARG;
  Field f = field_of(h, which);
  if (!f.ptr || f.is_int) return MI_ILQR_E_BAD_ARG;
  if (bytes != f.bytes) return MI_ILQR_E_BAD_SHAPE;
  HIPCHK(hipSetDevice(h->d.device_id));
  if (is_state_field(which)) { int rc = materialize_zero_state(h); if (rc != MI_ILQR_OK) return rc; }
  HIPCHK(hipStreamSynchronize(h->stream));
  int len = 0;
  const int rows = (h->large || h->batch_minor) ? traj_rows(h, which, &len) : 0;
  if (rows > 1 || (h->batch_minor && rows == 1)) {
    int rc = ensure_scratch(h, bytes);
    if (rc != MI_ILQR_OK) return rc;
    HIPCHK(hipMemcpy(h->scratch, src, bytes, hipMemcpyHostToDevice));
    if ((rc = relayout(h, h->scratch, static_cast<double*>(f.ptr), rows, len, true)) != MI_ILQR_OK) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
  } else {
    HIPCHK(hipMemcpy(f.ptr, src, bytes, hipMemcpyDefault));
  }
  if (which == MI_F_U_BAR) { h->u_pending = false; h->u_zero = false; }
  return MI_ILQR_OK;
}

int mi_ilqr_device_ptr(mi_ilqr_t* h, int which, void** ptr, size_t* bytes) {
  if (!h || !ptr) return MI_ILQR_E_BAD_ARG;
  Field f = field_of(h, which);
  if (!f.ptr) return MI_ILQR_E_BAD_ARG;
  *ptr = f.ptr;
  if (bytes) *bytes = f.bytes;
  return MI_ILQR_OK;
}

int mi_ilqr_set_timing(mi_ilqr_t* h, int32_t every) {
  if (!h || every < 0) return MI_ILQR_E_BAD_ARG;
  h->time_every = every;
  h->seq_timed = 0;
  return MI_ILQR_OK;
}

int mi_ilqr_last_kernel_ms(mi_ilqr_t* h, float* ms) {
  if (!h || !ms) return MI_ILQR_E_BAD_ARG;
  if (!h->last_timed) { *ms = 0.0f; return MI_ILQR_OK; }
  HIPCHK(hipSetDevice(h->d.device_id));
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipEventElapsedTime(ms, h->ev0, h->ev1));
  return MI_ILQR_OK;
}

int mi_ilqr_get_cycles(mi_ilqr_t* h, int64_t* dst, size_t bytes) {
  if (!h || !dst) return MI_ILQR_E_BAD_ARG;
  if (bytes != (size_t)h->B * 4 * 8) return MI_ILQR_E_BAD_SHAPE;
  HIPCHK(hipSetDevice(h->d.device_id));
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipMemcpy(dst, h->prof, bytes, hipMemcpyDefault));
  return MI_ILQR_OK;
}

int mi_ilqr_get_stream(mi_ilqr_t* h, void** hip_stream) {
  if (!h || !hip_stream) return MI_ILQR_E_BAD_ARG;
  *hip_stream = reinterpret_cast<void*>(h->stream);
  return MI_ILQR_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------
// The path's one collective: all-reduce(min) of the best costs over the ranks, RCCL over xGMI.
// librccl is bound at run time (dlopen), so libmi_ilqr.so has no link-time dependency on it; a
// process that already holds an RCCL (e.g. PyTorch's) shares that instance.
// ---------------------------------------------------------------------------------------------------
namespace {
struct RcclApi {
  // the few entry points used, with the types of rccl.h (ncclResult_t = int, ncclComm_t = opaque pointer,
  // ncclUniqueId = 128 bytes by value, ncclDataType_t / ncclRedOp_t = int enums whose values are taken from <rccl/rccl.h>)
  struct UniqueId { char internal[MI_ILQR_COMM_ID_BYTES]; };
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*CommCount)(void*, int*) = nullptr;
  int (*CommUserRank)(void*, int*) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};
// the data type and reduction codes come from the header the library is compiled against - never from memory (round 2
// shipped ncclAvg = 4 under the name "min")
constexpr int kNcclFloat64 = (int)ncclFloat64, kNcclMin = (int)ncclMin;
static_assert(sizeof(ncclUniqueId) == MI_ILQR_COMM_ID_BYTES, "the communicator id crosses the C ABI as MI_ILQR_COMM_ID_BYTES bytes");
static_assert((int)ncclSuccess == 0, "RCCLCHK treats 0 as success");

const RcclApi& rccl() {
  static const RcclApi api = [] {
    RcclApi a;
    void* lib = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so"})            // an instance already in the process first
      if (!lib) lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
      if (!lib) lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (!lib) return a;
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(lib, "ncclAllReduce"));
    a.CommCount = reinterpret_cast<decltype(a.CommCount)>(dlsym(lib, "ncclCommCount"));
    a.CommUserRank = reinterpret_cast<decltype(a.CommUserRank)>(dlsym(lib, "ncclCommUserRank"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
    a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce;
    return a;
  }();
  return api;
}
#define RCCLCHK(expr)                                                                               \
  do {                                                                                              \
    const int r_ = (expr);                                                                          \
    if (r_ != 0) {                                                                                  \
      std::fprintf(stderr, "mi_ilqr: %s failed: %s (%s:%d)\n", #expr,                               \
                   rccl().GetErrorString ? rccl().GetErrorString(r_) : "?", __FILE__, __LINE__);    \
      return MI_ILQR_E_RCCL;                                                                        \
    }                                                                                               \
  } while (0)
}  // namespace

struct mi_ilqr_comm {
  void* comm = nullptr;
  int rank = 0, world = 1, device = 0;
  hipStream_t stream = nullptr;     // the collective's own stream: it overlaps the solves of the handles
  hipEvent_t done = nullptr;
  double* d_buf = nullptr;          // MI_ILQR_COMM_MAX_COUNT doubles on the device
  double* h_buf = nullptr;          // pinned staging
  int in_flight = 0;                // count of the reduction started and not yet waited for
};

extern "C" {

int mi_ilqr_comm_unique_id(void* id_bytes) {
  if (!id_bytes) return MI_ILQR_E_BAD_ARG;
  if (!rccl().ok) return MI_ILQR_E_RCCL;
  RcclApi::UniqueId id;
  RCCLCHK(rccl().GetUniqueId(&id));
  std::memcpy(id_bytes, id.internal, MI_ILQR_COMM_ID_BYTES);
  return MI_ILQR_OK;
}

int mi_ilqr_comm_create(const void* id_bytes, int32_t rank, int32_t world, int32_t device_id, mi_ilqr_comm_t** out) {
  if (!id_bytes || !out || world < 1 || rank < 0 || rank >= world) return MI_ILQR_E_BAD_ARG;
  *out = nullptr;
  if (!rccl().ok) return MI_ILQR_E_RCCL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) return MI_ILQR_E_NO_DEVICE;
  HIPCHK(hipSetDevice(device_id));
  mi_ilqr_comm* c = new (std::nothrow) mi_ilqr_comm();
  if (!c) return MI_ILQR_E_BAD_ARG;
  c->rank = rank; c->world = world; c->device = device_id;
  RcclApi::UniqueId id;
  std::memcpy(id.internal, id_bytes, MI_ILQR_COMM_ID_BYTES);
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&c->done) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&c->d_buf), MI_ILQR_COMM_MAX_COUNT * 8) != hipSuccess ||
      hipHostMalloc(reinterpret_cast<void**>(&c->h_buf), MI_ILQR_COMM_MAX_COUNT * 8, hipHostMallocDefault) != hipSuccess) {
    mi_ilqr_comm_destroy(c);
    return MI_ILQR_E_HIP;
  }
  const int r = rccl().CommInitRank(&c->comm, world, id, rank);
  if (r != 0) {
    std::fprintf(stderr, "mi_ilqr: ncclCommInitRank failed: %s\n", rccl().GetErrorString ? rccl().GetErrorString(r) : "?");
    c->comm = nullptr;
    mi_ilqr_comm_destroy(c);
    return MI_ILQR_E_RCCL;
  }
  *out = c;
  return MI_ILQR_OK;
}

void mi_ilqr_comm_destroy(mi_ilqr_comm_t* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm && rccl().ok) (void)rccl().CommDestroy(c->comm);
  if (c->d_buf) (void)hipFree(c->d_buf);
  if (c->h_buf) (void)hipHostFree(c->h_buf);
  if (c->done) (void)hipEventDestroy(c->done);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int mi_ilqr_comm_count(mi_ilqr_comm_t* c, int32_t* ranks, int32_t* rank) {
  // what the COMMUNICATOR says (ncclCommCount / ncclCommUserRank), not what the caller passed to mi_ilqr_comm_create
  if (!c || !c->comm || !ranks) return MI_ILQR_E_BAD_ARG;
  if (!rccl().CommCount) return MI_ILQR_E_RCCL;
  int n = 0, r = -1;
  RCCLCHK(rccl().CommCount(c->comm, &n));
  if (rank && rccl().CommUserRank) RCCLCHK(rccl().CommUserRank(c->comm, &r));
  *ranks = n;
  if (rank) *rank = r;
  return MI_ILQR_OK;
}

int mi_ilqr_allreduce_min_start(mi_ilqr_comm_t* c, const double* values, int32_t count) {
  if (!c || !values || count < 1 || count > MI_ILQR_COMM_MAX_COUNT || c->in_flight) return MI_ILQR_E_BAD_ARG;
  HIPCHK(hipSetDevice(c->device));
  std::memcpy(c->h_buf, values, (size_t)count * 8);
  HIPCHK(hipMemcpyAsync(c->d_buf, c->h_buf, (size_t)count * 8, hipMemcpyHostToDevice, c->stream));
  RCCLCHK(rccl().AllReduce(c->d_buf, c->d_buf, (size_t)count, kNcclFloat64, kNcclMin, c->comm, c->stream));
  HIPCHK(hipMemcpyAsync(c->h_buf, c->d_buf, (size_t)count * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipEventRecord(c->done, c->stream));
  c->in_flight = count;
  return MI_ILQR_OK;
}

int mi_ilqr_allreduce_min_wait(mi_ilqr_comm_t* c, double* values, int32_t count) {
  if (!c || !values || count != c->in_flight || count < 1) return MI_ILQR_E_BAD_ARG;
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipEventSynchronize(c->done));
  std::memcpy(values, c->h_buf, (size_t)count * 8);
  c->in_flight = 0;
  return MI_ILQR_OK;
}

int mi_ilqr_allreduce_min(mi_ilqr_comm_t* c, double* values, int32_t count) {
  const int rc = mi_ilqr_allreduce_min_start(c, values, count);
  if (rc != MI_ILQR_OK) return rc;
  return mi_ilqr_allreduce_min_wait(c, values, count);
}

}  // extern "C"
