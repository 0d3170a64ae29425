/*
 * mi_ilqr.h — C ABI of libmi_ilqr.so: the MI355X-native batched iLQR/DDP hot path.
 *
 * This is the drop-in boundary for the path BASELINE.json:north_star names: the
 * body of IterativeLinearQuadraticRegulator.Solve() in the reference
 * (/root/reference/ilqr.py:669-710) and the stages it calls, for a BATCH of B
 * independent problems that share model, horizon and cost matrices.  Each entry
 * point cites the reference interface it replaces.  Plain pointers and sizes
 * only; all `double*` arguments are HOST memory unless the name says `device`;
 * the library copies in/out and never keeps or frees caller memory.
 *
 * Array layout is the reference's, per problem (time on the LAST
 * axis, C order), with a leading batch axis:
 *     x_bar (B,n,N)   u_bar (B,m,N-1)   K (B,m,n,N-1)   kappa (B,m,N-1)
 *     fx (B,n,n,N-1)  fu (B,n,m,N-1)    dV_coeff (B,N-1)
 *
 * Threading: a handle is not thread-safe (the reference class is not either:
 * it mutates self.context, ilqr.py:223-229); distinct handles are independent.
 * One handle = one GPU = one HIP stream.  Multi-GPU = one process per GPU, each
 * with its own handle over its shard of the batch (no data-path collective).
 *
 * There is NO CPU fallback: every compute entry returns MI_ILQR_E_NO_DEVICE when
 * no gfx950 device is usable.
 */
#ifndef MI_ILQR_H
#define MI_ILQR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_ILQR_ABI_VERSION 9   /* 9: mi_ilqr_comm_count; partial reads of MI_F_HIST / MI_F_ITER_CYCLES; MI_I64_CLUSTER_WORDS reads as zeros where no cluster ran;
                                   7: mi_ilqr_desc.on_indefinite, mi_ilqr_model_plugin.m_user, 256 plugin slots;
                                   8: MI_STATUS_FLAG_INDEFINITE, on_indefinite = 1 inverts with partial pivoting, asymmetric costs for n <= 32,
                                      the diagnostic field MI_I64_CLUSTER_WORDS */
#define MI_ILQR_MAX_PARAMS 16
#define MI_ILQR_CLUSTER_WORDS 40 /* 64-bit words per problem of the diagnostic field MI_I64_CLUSTER_WORDS */

/* Error codes (0 = OK).  The Python wrapper maps them onto the exception types
 * the reference raises (SURVEY.md §8b "Error convention"). */
enum {
  MI_ILQR_OK = 0,
  MI_ILQR_E_BAD_SHAPE = -1,      /* reference: assert on shapes, ilqr.py:130-131,145,155 */
  MI_ILQR_E_BAD_METHOD = -2,     /* reference: Exception('unknown interpolation method'), ilqr.py:404 */
  MI_ILQR_E_LINESEARCH = -3,     /* reference: RuntimeError("linesearch failed ..."), ilqr.py:337 */
  MI_ILQR_E_HIP = -4,
  MI_ILQR_E_NO_DEVICE = -5,
  MI_ILQR_E_BAD_ARG = -6,
  MI_ILQR_E_UNSUPPORTED = -7,    /* model / size / cost-matrix combination no kernel covers */
  MI_ILQR_E_RCCL = -8            /* librccl missing, or a collective / communicator call failed */
};

/* Device dynamics models (the `system` argument of ilqr.py:21 becomes a model
 * descriptor; Drake systems cannot run on the GPU — SURVEY.md §8b).  Parameter
 * vectors are documented in drake_ddp_amd/csrc/models.hpp. */
enum {
  MI_MODEL_PENDULUM = 0,       /* n=2  m=1  */
  MI_MODEL_ACROBOT = 1,        /* n=4  m=1  */
  MI_MODEL_CARTPOLE = 2,       /* n=4  m=1  */
  MI_MODEL_CARTPOLE_WALL = 3,  /* n=4  m=1  */
  MI_MODEL_SYNTH36 = 4,        /* n=36 m=12 */
  MI_MODEL_PLANAR_QUAD = 5,    /* n=36 m=12: planar floating-base quadruped (articulated-body algorithm, ground
                                  contact); can declare a step INFEASIBLE - such a line-search trial costs +inf, as
                                  when Drake's update throws (ilqr.py:315-323) */
  MI_MODEL_QUAD3D = 6,         /* n=37 m=12: 3-D floating-base quadruped with mini_cheetah.py:41-52's state layout
                                  (unit quaternion | position | 12 joints | 18 velocities), feet contact; can declare a
                                  step infeasible like the planar one */
  MI_MODEL_ARM27 = 7,          /* n=27 m=7: 7-joint arm pushing a free ball - the state kinova_gen3.py:52-70 / panda_fr3.py
                                  stack (7 joint angles | the ball's unit quaternion, position | 13 velocities); served by
                                  the mid-size workgroup-per-problem kernels */
  MI_MODEL_ARM27C = 8          /* n=27 m=7: the same arm, ball and contacts with COUPLED rigid-body joint dynamics - the joint-space
                                  mass matrix of three point masses + rotor inertias, centripetal / Coriolis and gravity terms,
                                  M(q) qdd = tau - ... solved per step (L D L^T); 16 parameters (Arm27's + m_wrist) */
};

/* utils_derivs_interpolation.derivs_interpolation.keypoint_method
 * (/root/reference/utils_derivs_interpolation.py:4-9; strings at ilqr.py:396-400). */
enum { MI_KP_SET_INTERVAL = 0, MI_KP_ADAPTIVE_JERK = 1, MI_KP_ITERATIVE_ERROR = 2 };

/* How fx/fu are obtained (replaces _calc_dynamics_partials, ilqr.py:233-272). */
enum { MI_JAC_FD_CENTRAL = 0, MI_JAC_AUTODIFF = 1 };

/* Which kernel family serves a small-state model (mi_ilqr_desc.kernel_mode):
 *   LATENCY    wave-per-problem, state in LDS: minimal time-to-solution, up to ~2k problems/GPU in flight;
 *   THROUGHPUT lane-per-problem, batch-minor state streamed through HBM: for tens of thousands of
 *              problems (every key-point method - a key-point list per lane; stage-level entries are not
 *              available; built-in models and family-0 plugins with n <= 6);
 *   AUTO       THROUGHPUT when B >= 8192 and the configuration allows it - except n = 2 models with
 *              N <= 257, whose LATENCY kernel (rollout and Riccati sweep parallel in time) is the faster one
 *              at every batch size - else LATENCY. */
enum { MI_KERNEL_AUTO = 0, MI_KERNEL_LATENCY = 1, MI_KERNEL_THROUGHPUT = 2 };

/* Per-problem status written by solve/forward. */
enum { MI_STATUS_CONVERGED = 0, MI_STATUS_MAX_ITERS = 1, MI_STATUS_LINESEARCH_FAILED = 2,
       MI_STATUS_INTERNAL = 3, /* a helper workgroup of the problem's cluster stopped answering (never seen) */
       /* 4: unused */
       MI_STATUS_NOT_PD = 5,   /* workgroup-per-problem kernels (n >= 5 models with m > 2), mi_ilqr_desc.on_indefinite = 0: a backward
                                  pass met a Quu that is not positive definite (a pivot of its unpivoted elimination <= 0 or not
                                  finite) - an indefinite cost expansion or one ruined by round-off - and the problem STOPPED there:
                                  its gains are NOT to be used.  The reference inverts such a Quu all the same (np.linalg.inv,
                                  ilqr.py:655) and carries on: that is on_indefinite = 1 */
       MI_STATUS_FLAG_INDEFINITE = 16 /* OR-ed onto the outcome above (ABI 8), on_indefinite = 1: a backward pass of this solve (of this
                                  mpc_run) met such a Quu, inverted it with partial pivoting like the reference and carried on;
                                  `status & ~MI_STATUS_FLAG_INDEFINITE` is the solve's outcome.  Counted in stats.n_not_pd too */ };

/* Selector for mi_ilqr_get / mi_ilqr_set / mi_ilqr_device_ptr. */
enum {
  MI_F_X_BAR = 0, MI_F_U_BAR = 1, MI_F_K = 2, MI_F_KAPPA = 3, MI_F_DV = 4, MI_F_FX = 5, MI_F_FU = 6,
  MI_F_COST = 7,        /* (B,)   total cost L of x_bar/u_bar                          */
  MI_F_X0 = 8,          /* (B,n)                                                        */
  MI_F_HIST = 9,        /* (B,hist_cap,4) rows (L, eps, ls_trials, percentage_derivs)   */
  MI_F_X_TRIAL = 10,    /* (B,n,N)   last mi_ilqr_rollout trajectory                    */
  MI_F_U_TRIAL = 11,    /* (B,m,N-1)                                                    */
  MI_F_TRIAL_COST = 12, /* (B,2) (L, expected_improvement) of the last rollout          */
  MI_F_ITER_CYCLES = 13,/* (B,hist_cap,4) per-iteration stopwatches of the last solve, shader-clock cycles: line search,
                           linearization (0 when the rollout linearized on its way), backward pass, whole iteration - the
                           reference's time_fp / time_getDerivs / time_backwardsPass / iter time (ilqr.py:364-372,696-702);
                           wave- and workgroup-per-problem kernels */
  /* int32 fields (mi_ilqr_get_int) */
  MI_I_ITERS = 100,     /* (B,) iterations of the last solve                            */
  MI_I_STATUS = 101,    /* (B,)                                                         */
  MI_I_LS_TRIALS = 102, /* (B,) reference-equivalent line-search trials, summed         */
  MI_I_KP_COUNT = 103,  /* (B,) key-points used by the last linearization               */
  MI_I_KP_LIST = 104,   /* (B,N-1) the key-point indices, first KP_COUNT valid          */
  MI_I64_STAGE_CYCLES = 200, /* (B,4) int64: what mi_ilqr_get_cycles returns - here so that mi_ilqr_get_async can queue it behind a solve */
  MI_I64_CLUSTER_WORDS = 201 /* (B,MI_ILQR_CLUSTER_WORDS) uint64 (mi_ilqr_get_int; diagnostic): the handshake words of the last solve / MPC launch
                                when it shared its linearizations among clusters of workgroups (workgroup-per-problem kernels) - zeros when it
                                did not, and for handles of the other kernel families - [1] helper shares
                                finished, [2] & 0xffff helpers that took part, [2] >> (16 + 6 x) & 63 how many of them ran on XCD x,
                                [3] >> 32 rounds (regular + early), [3] >> 8 & 0xffffff rounds that found every helper on the leader's
                                own XCD (one L2: no cache-wide invalidate), [4] progress word of the last early round, [5] >> 32 early
                                rounds opened (the helpers linearize the line search's first trial while it is being rolled out),
                                [5] & 0xffffffff early rounds whose trial was accepted, [7] candidate-group rounds (mid-size kernels: the
                                helpers roll out the line-search candidates 4 .. beside the leader's four), [8 ..] the costs of those */
};

typedef struct mi_ilqr mi_ilqr_t;

/* Everything IterativeLinearQuadraticRegulator.__init__ takes (ilqr.py:21-22,
 * 51-58, 97-100), plus the batch size and the model descriptor. */
typedef struct {
  int32_t n, m;            /* must equal the model's dimensions (ilqr.py:57-58) */
  int32_t N;               /* num_timesteps (ilqr.py:51) */
  int32_t B;               /* problems in this handle's shard */
  int32_t model_id;
  int32_t n_params;
  double model_params[MI_ILQR_MAX_PARAMS];
  double dt;
  double delta, beta, gamma;                 /* ilqr.py:52-54 */
  int32_t keypoint_method, minN, maxN;       /* derivs_interpolation fields */
  double jerk_threshold, iterative_error_threshold;
  int32_t jacobian_mode;
  double fd_step;                            /* absolute central-difference step */
  int32_t max_iters;                         /* safety cap (reference has none); <=0 -> 1000 */
  int32_t hist_cap;                          /* per-problem iteration rows kept; <=0 -> 64 */
  int32_t device_id;                         /* HIP device ordinal */
  int32_t kernel_mode;                       /* MI_KERNEL_AUTO / _LATENCY / _THROUGHPUT */
  int32_t on_indefinite;                     /* workgroup-per-problem kernels, a Quu that is not positive definite in a backward pass:
                                                0 = stop that problem with MI_STATUS_NOT_PD (default of this struct); 1 = what the reference
                                                does (np.linalg.inv, ilqr.py:655): invert it all the same - with partial pivoting, a cold
                                                path no positive definite Quu ever enters - carry on, and flag the problem's status with
                                                MI_STATUS_FLAG_INDEFINITE.  (The wave- and lane-per-problem kernels always behave like 1,
                                                without the flag: their m <= 2 inverses are closed forms.) */
} mi_ilqr_desc;

typedef struct {
  int64_t total_iters;        /* sum_b iterations */
  int64_t total_ls_trials;    /* sum_b sum_i ls_{b,i} (reference-equivalent trials) */
  int32_t n_converged, n_max_iters, n_ls_failed;
  int32_t max_iters_seen;
  double best_cost;           /* min_b L_b over converged problems */
  int32_t best_index;
  float kernel_ms;            /* HIP-event time of the solve kernel(s) on the handle's stream; 0 for a launch
                                 * that mi_ilqr_set_timing left without events */
  double algorithmic_bytes;   /* sum_b sum_i bytes_iter(ls_{b,i}) — SURVEY.md §8d formula */
  int32_t n_internal;         /* problems aborted with MI_STATUS_INTERNAL (a lost cluster helper; counted apart from
                               * n_ls_failed since ABI 5: their x_bar / u_bar are NOT a solution) */
  int32_t n_not_pd;           /* problems stopped with MI_STATUS_NOT_PD (ABI 6; the field was reserved before) */
} mi_ilqr_stats;

int mi_ilqr_abi_version(void);
/* sizeof(mi_ilqr_desc), sizeof(mi_ilqr_stats), sizeof(mi_ilqr_model_plugin) as the LIBRARY was compiled: a binding that restates the
 * structs in its own language (ctypes, cgo, JNI) compares them with its own at load time - a field added on one side only
 * would otherwise shift every later field silently.  Any pointer may be NULL. */
void mi_ilqr_struct_sizes(int32_t* desc_bytes, int32_t* stats_bytes, int32_t* plugin_bytes);
const char* mi_ilqr_strerror(int code);

/* Model registry: dimensions and default parameters of a model id. */
int mi_ilqr_model_info(int model_id, int32_t* n, int32_t* m, int32_t* n_params, double* default_params);

/* OPEN MODEL INTERFACE.  The reference takes any discrete System (ilqr.py:21,37-58); here a model is a C++ struct with
 * `template <class T> static void step(const T* x, const T* u, T* xn, const double* params, double dt)` (T = double, or
 * the forward-mode dual types of csrc/dual.hpp) compiled against the kernel headers of drake_ddp_amd/csrc into a PLUGIN
 * shared library - one translation unit, ~25 s of hipcc, nothing of libmi_ilqr.so is rebuilt
 * (drake_ddp_amd/plugin.py writes and builds that unit; INTEGRATION.md section 5 shows it by hand).  The plugin hands the
 * library this record; the returned id (>= MI_MODEL_PLUGIN_BASE) is used as mi_ilqr_desc.model_id like a built-in one.
 *   family 0: wave-per-problem kernels (state in LDS; any n, m <= 2 - n = 2 takes the time-parallel passes, n = 3..4 the
 *             matrix-core backward step, other n the scalar recursion);
 *   family 1: workgroup-per-problem kernels - n <= 32 with ANY m <= 16 (one or two 16-row tiles: the shapes of a quadrotor
 *             (12, 4), a 7-joint arm (14, 7), kinova_gen3.py's arm + free body (27, 7)), or 32 < n <= 40 with m <= 16,
 *             m % 4 == 0, 2 m <= n; dynamics as `step` per Jacobian column and a one-lane step in the rollout unless the
 *             model provides the cooperative hooks of csrc/models.hpp.  A model with 32 < n <= 40 and another number of
 *             controls declares m as the next multiple of 4 and ignores the extra ones; with any positive cost on them
 *             (R block-diagonal) and a zero initial guess their gains, feed-forward terms and values stay EXACT zeros
 *             (zero columns of fu, zero rows of Qux) and every other result is what the unpadded problem gives.  The
 *             Python mirror does this by itself (drake_ddp_amd/plugin.py pads, drake_ddp_amd/ilqr.py hides it).
 * Family-0 plugins with n <= 6 also carry the lane-per-problem THROUGHPUT kernels (kernel_mode, batches >= 8192 and
 * horizons beyond LDS under AUTO, like the built-in small models); other plugins are served by their family only. */
enum { MI_MODEL_PLUGIN_BASE = 100, MI_ILQR_MAX_PLUGINS = 256 };
typedef struct {
  int32_t abi_version;              /* MI_ILQR_ABI_VERSION of the headers the plugin was compiled against ... */
  int32_t kernel_args_bytes;        /* ... their sizeof(mi::KArgs) ... */
  int32_t handle_bytes;             /* ... and their sizeof(struct mi_ilqr) (the plugin's launch code reads the handle's stream,
                                       events and LDS size): a plugin built from other headers is refused (ABI 6) */
  int32_t m_user;                   /* 0, or the number of controls the model's step READS when that is fewer than m: controls
                                       m_user .. m-1 are padding (see family 1 above); informational - hosts size their arrays with m */
  int32_t n, m, n_params, family;
  double default_params[MI_ILQR_MAX_PARAMS];
  int (*launch)(mi_ilqr_t* h, int mode, const void* kernel_args);   /* instantiates and launches the model's kernels */
  size_t (*lds_bytes)(int32_t N, int32_t n_store);                  /* dynamic LDS of one problem (family 1: n_store < 0 asks for the
                                                                       size with the horizon's cost gradients kept in HBM) */
} mi_ilqr_model_plugin;
int mi_ilqr_register_model(const mi_ilqr_model_plugin* plugin, int32_t* model_id_out);

/* ilqr.py:21-100 — allocate the solver state on the device, zeroed (ilqr.py:70-83). */
int mi_ilqr_create(const mi_ilqr_desc* desc, mi_ilqr_t** out);
void mi_ilqr_destroy(mi_ilqr_t* h);

/* SetRunningCost / SetTerminalCost / SetTargetState (ilqr.py:111-146): Q (n,n), R (m,m),
 * Qf (n,n), x_nom (n), shared by the batch.  Any pointer may be NULL = keep.  ANY finite matrices are accepted, like the
 * reference (lxx = 2Q, luu = 2R, lx = 2Qx - 2 x_nom^T Q, never symmetrized - ilqr.py:180-184), by every kernel family:
 * wave- and lane-per-problem kernels - symmetric positive semi-definite Q, Qf and positive definite R take the
 * time-parallel / matrix-core backward passes, anything else the reference's recursion verbatim;  mid-size
 * workgroup-per-problem kernels (m > 2 or n > 6, n <= 32: Arm27, plugin family 1) - the matrix-core pass itself uses no
 * symmetry when a matrix is not symmetric (ABI 8);  the n = 33..40 kernels (n = 36 / 37 models, plugin family 1 above 32
 * states) - their matrix-core chain mirrors tiles of the symmetric products, so matrices that are not symmetric take a
 * plain-arithmetic form of the pass (ABI 9: large_backward_asym, about four times the cycles per step; MI_ILQR_E_UNSUPPORTED
 * before), and asymmetries of up to 8 ulp of the largest entry - round-off of an A^T A - are averaged away so that such
 * matrices keep the fast pass.  Definiteness is required of none - it is
 * checked where it matters: mi_ilqr_desc.on_indefinite says what a backward pass does with a Quu that is not positive
 * definite. */
int mi_ilqr_set_cost(mi_ilqr_t* h, const double* Q, const double* R, const double* Qf, const double* x_nom);

/* SetInitialState / SetInitialGuess (ilqr.py:102-109,148-156): x0 (B,n), u_guess (B,m,N-1).
 * u_guess becomes u_bar (the reference aliases it, ilqr.py:156).  NULL = keep.
 * Ordering against mi_ilqr_solve_async: the new inputs are those of the NEXT solve.  Handles of more than four problems copy them
 * on the handle's stream (the call returns at once, behind a solve that is still running); wave-per-problem handles of up to four
 * problems keep x0 / u_guess in mapped host memory that the kernels read directly (no copy engine in front of a single-problem
 * solve), so there the call WAITS for a solve still in flight before it overwrites them - a caller that pipelines
 * solve_async / set_initial / collect on such a handle serializes at set_initial. */
int mi_ilqr_set_initial(mi_ilqr_t* h, const double* x0, const double* u_guess);

/* The same with ONE control sequence u_guess_one (m,N-1) for every problem of the batch - the argument the
 * reference's SetInitialGuess takes (ilqr.py:148-156): m(N-1) doubles cross the bus instead of B*m*(N-1), the
 * device writes the batch's copies. */
int mi_ilqr_set_initial_shared(mi_ilqr_t* h, const double* x0, const double* u_guess_one);

/* Page-locked host memory for the buffers a caller hands to mi_ilqr_set / _get / _set_initial: the copies then
 * run as direct DMA (~55 GB/s over PCIe 5 x16) instead of through the runtime's staging of pageable memory.
 * Needs a usable device; independent of any handle. */
int mi_ilqr_host_alloc(size_t bytes, void** out);
int mi_ilqr_host_free(void* p);

/* Zero the persistent solver state (x_bar,u_bar,K,kappa,dV,fx,fu) = a freshly constructed
 * reference object (ilqr.py:70-83): a solve after it without mi_ilqr_set_initial(u_guess) /
 * mi_ilqr_rearm_initial_guess starts from u_bar = 0.  Without it the state persists across solves (F10). */
int mi_ilqr_reset(mi_ilqr_t* h);

/* Benchmark/MPC helper: make the resident u_guess (last mi_ilqr_set_initial or
 * mi_ilqr_mpc_shift) the initial guess of the next solve again, without host traffic. */
int mi_ilqr_rearm_initial_guess(mi_ilqr_t* h);

/* Solve (ilqr.py:669-710) for every problem of the batch, on the device, to convergence.
 * Blocking, like the reference call; `stats` may be NULL.  _solve_async only enqueues the
 * kernel on the handle's stream - ONE dispatch: every solve leaves its per-problem cost / iterations / status /
 * line-search trials in its own slot of a 32-deep ring, and _collect_stats(_n) reduces all slots still owed
 * their batch statistics in one launch, synchronizes and returns them (more than 32 uncollected solves:
 * the oldest are overwritten). */
int mi_ilqr_solve(mi_ilqr_t* h, mi_ilqr_stats* stats);
int mi_ilqr_solve_async(mi_ilqr_t* h);
int mi_ilqr_collect_stats(mi_ilqr_t* h, mi_ilqr_stats* stats);
/* Up to 32 solves may be enqueued with _solve_async before collecting (each keeps its own kernel
 * events and statistics record): the statistics of the last `count` of them, oldest first. */
int mi_ilqr_collect_stats_n(mi_ilqr_t* h, int32_t count, mi_ilqr_stats* stats);

/* Stage-level entries (parity tests).
 * rollout : one line-search trial per problem with the given eps (ilqr.py:306-327)
 *           -> MI_F_X_TRIAL / MI_F_U_TRIAL / MI_F_TRIAL_COST.
 * forward : _forward_pass (ilqr.py:339-378): line search against L_last (B,; +inf allowed),
 *           linearization at the accepted trajectory, commit to x_bar/u_bar -> MI_F_COST,
 *           HIST row 0 holds (L, eps, ls, pct).
 * linearize: _get_derivatives (ilqr.py:380-415) at the current x_bar/u_bar.
 * backward: _backward_pass (ilqr.py:623-667). */
int mi_ilqr_rollout(mi_ilqr_t* h, const double* eps);
int mi_ilqr_forward(mi_ilqr_t* h, const double* L_last);
int mi_ilqr_linearize(mi_ilqr_t* h);
int mi_ilqr_backward(mi_ilqr_t* h);

/* MPC warm start on the device (acrobot.py:147-152, mini_cheetah.py:193-198):
 * x0 <- x_bar[:, replan_steps]; u_bar <- [u_bar[:, replan_steps:], repeat(u_bar[:, -1])]. */
int mi_ilqr_mpc_shift(mi_ilqr_t* h, int32_t replan_steps);

/* The whole receding-horizon loop of acrobot.py:145-155 / mini_cheetah.py:190-201 on the device:
 * `num_resolves` times { mpc_shift(replan_steps); x_nom += target_step (may be NULL); Solve }.
 * For the wave-per-problem kernels this is ONE launch and the solver state stays in LDS between
 * re-solves; the workgroup-per-problem kernel (n = 36) runs the loop in one launch as well.  Per
 * re-solve the log keeps (x0 (n), cost, iterations) for every problem:
 * mi_ilqr_get_mpc_log -> (B, num_resolves, n+2).  stats aggregate the whole loop; the per-problem status
 * is that of the LAST re-solve (a loop that hits a line-search failure stops there).
 * Limits of the single-launch form: wave-per-problem kernels N <= 512 (the in-kernel shift holds eight
 * controls per lane), workgroup-per-problem kernel m*(N-1) <= 2048.  Beyond them, and for the
 * lane-per-problem "throughput" kernels, the same loop runs as shift + solve launches from the host:
 * same results, same log (filled after each re-solve).
 * A problem whose re-solve r FAILS (line search, MI_STATUS_NOT_PD, internal) gets row r of the log and none after it:
 * the log is zero-filled at every call and both forms stop logging the problem there (rows r+1.. read as zeros).  The
 * single-launch forms also stop RE-SOLVING it (its status is that of re-solve r); the host-loop form's batched launches
 * cannot leave a problem out - it is shifted and solved on, and its final status is the last re-solve's. */
int mi_ilqr_mpc_run(mi_ilqr_t* h, int32_t num_resolves, int32_t replan_steps, const double* target_step, mi_ilqr_stats* stats);
int mi_ilqr_get_mpc_log(mi_ilqr_t* h, double* dst, size_t bytes);

/* Copy a field out / in (host memory; `bytes` must equal the field size - except that the per-iteration records MI_F_HIST and
 * MI_F_ITER_CYCLES may be read in part: `bytes` = any whole number of 32-byte rows < the field size returns the LEADING rows of
 * the first problem, also through mi_ilqr_get_async and mi_ilqr_solve_into; the drop-in class keeps a 4096-row log and copies 64
 * rows with the solve, the rest only after a solve that took more iterations - the reference's table has every row, ilqr.py:704). */
int mi_ilqr_get(mi_ilqr_t* h, int which, double* dst, size_t bytes);
int mi_ilqr_get_int(mi_ilqr_t* h, int which, int32_t* dst, size_t bytes);
int mi_ilqr_set(mi_ilqr_t* h, int which, const double* src, size_t bytes);

/* Enqueue the copy-out of a field (double or int32) on the handle's stream and return: `dst` is defined after
 * the next mi_ilqr_synchronize / _collect_stats.  Meant for page-locked destinations (mi_ilqr_host_alloc): several
 * results of a solve then cost one synchronization instead of one each.  Fields that need a layout conversion
 * (the n = 36 and lane-per-problem kernels' trajectory arrays) are copied before the call returns, like mi_ilqr_get. */
int mi_ilqr_get_async(mi_ilqr_t* h, int which, void* dst, size_t bytes);

/* Result sink (wave-per-problem kernels; MI_ILQR_E_UNSUPPORTED for the n = 36/37 and lane-per-problem layouts): three
 * host arrays from mi_ilqr_host_alloc - x_bar (B,n,N), u_bar (B,m,N-1), cost (B,), the boundary's layouts - that every
 * later solve / mpc_run kernel ALSO writes its results into, problem by problem as each finishes, over the host link:
 * after mi_ilqr_collect_stats / mi_ilqr_synchronize they hold what mi_ilqr_get would return, and the copy-out of the
 * batch has overlapped the launch's slowest problems instead of following it (Solve() through the class surface:
 * 0.32 -> 0.2x ms at B = 1024).  The arrays must stay allocated while the sink is set; three NULLs clear it.
 * The persistent state in HBM is written as always (warm starts, mi_ilqr_get). */
int mi_ilqr_set_result_sink(mi_ilqr_t* h, double* x_bar_host, double* u_bar_host, double* cost_host);

/* One blocking solve whose results land in caller memory, ONE host synchronization, one call across the boundary (the
 * single-problem Solve() of the drop-in class: C1's latency is host overhead as much as kernel time): x_bar (B,n,N), u_bar
 * (B,m,N-1), cost (B,) - written by the kernel itself as each problem finishes when `page_locked` is non-zero, the three come
 * from mi_ilqr_host_alloc and the kernel family has a result sink (set for THIS solve only), copied out behind the solve
 * otherwise - plus `n_extra` further fields (double or int selectors: which[i] -> dst[i], bytes[i] as for mi_ilqr_get_async)
 * queued behind it.  stats and sink_used (1: the kernel wrote the three arrays itself) may be NULL. */
int mi_ilqr_solve_into(mi_ilqr_t* h, double* x_bar, double* u_bar, double* cost, int32_t page_locked, int32_t n_extra,
                       const int32_t* which, void* const* dst, const size_t* bytes, mi_ilqr_stats* stats, int32_t* sink_used);

/* Raw device pointer of a double field (for zero-copy consumers, e.g. a torch tensor
 * view feeding the RCCL best-cost reduction), and the handle's stream.  The per-problem result scalars
 * (MI_F_COST, MI_I_ITERS, MI_I_STATUS, MI_I_LS_TRIALS) rotate through a ring, one slot per
 * mi_ilqr_solve / mi_ilqr_solve_async: ask for their pointer again after each solve. */
int mi_ilqr_device_ptr(mi_ilqr_t* h, int which, void** ptr, size_t* bytes);
int mi_ilqr_get_stream(mi_ilqr_t* h, void** hip_stream);
int mi_ilqr_synchronize(mi_ilqr_t* h);

/* ---- multi-GPU: the path's one collective -------------------------------------------
 * Problems are independent: every rank (one process per GPU) owns a handle over its contiguous shard of
 * the batch and no kernel ever exchanges data.  The only cross-rank step is the reduction of the best
 * total cost at the end of a batched solve: ONE RCCL all-reduce(min) of a few doubles over xGMI, here so
 * that a C caller needs nothing but this header.  librccl is loaded on first use (dlopen); without it
 * these entries return MI_ILQR_E_RCCL and everything else keeps working.
 *   rank 0:  mi_ilqr_comm_unique_id(id)  ->  ship the MI_ILQR_COMM_ID_BYTES bytes to every rank over the
 *            launcher's own channel (MPI_Bcast, a file, a TCP store)
 *   all:     mi_ilqr_comm_create(id, rank, world, device_id, &comm)          (collective call)
 *   all:     mi_ilqr_allreduce_min(comm, values, count)                      (blocking; in place)
 *        or  mi_ilqr_allreduce_min_start / _wait: the reduction runs on the communicator's own stream and
 *            overlaps the next solve; at most one in flight per communicator; count <= 64. */
#define MI_ILQR_COMM_ID_BYTES 128
#define MI_ILQR_COMM_MAX_COUNT 64
typedef struct mi_ilqr_comm mi_ilqr_comm_t;
int mi_ilqr_comm_unique_id(void* id_bytes);
int mi_ilqr_comm_create(const void* id_bytes, int32_t rank, int32_t world, int32_t device_id, mi_ilqr_comm_t** out);
void mi_ilqr_comm_destroy(mi_ilqr_comm_t* c);
/* The communicator's own answer (ncclCommCount, ncclCommUserRank): how many ranks it spans and which one this is - what a
 * multi-GPU bench line reports so that a scaling record can be checked from the line alone (ABI 9).  `rank` may be NULL. */
int mi_ilqr_comm_count(mi_ilqr_comm_t* c, int32_t* ranks, int32_t* rank);
int mi_ilqr_allreduce_min(mi_ilqr_comm_t* c, double* values, int32_t count);
int mi_ilqr_allreduce_min_start(mi_ilqr_comm_t* c, const double* values, int32_t count);
int mi_ilqr_allreduce_min_wait(mi_ilqr_comm_t* c, double* values, int32_t count);

/* Per-problem in-kernel stopwatches of the last solve, shader-clock cycles, (B,4):
 * line search, linearization, backward pass, whole Solve loop — the device counterpart
 * of the reference's time_fp / time_getDerivs / time_backwardsPass (ilqr.py:364-372,696-699). */
int mi_ilqr_get_cycles(mi_ilqr_t* h, int64_t* dst, size_t bytes);

/* HIP-event duration of the most recent kernel launch on the handle's stream (0 if that launch was not timed). */
int mi_ilqr_last_kernel_ms(mi_ilqr_t* h, float* ms);

/* Which launches carry their start/stop events: every one (every = 1, the default), the first of every `every`
 * solves, or none (every = 0).  The events ride on the solve kernel's own dispatch packet, but a profiled dispatch
 * still serializes a pipelined stream by ~5 us; a caller that enqueues solves back to back (mi_ilqr_solve_async)
 * and wants kernel_ms only as a sample asks for one in k.  Untimed launches report kernel_ms = 0. */
int mi_ilqr_set_timing(mi_ilqr_t* h, int32_t every);

/* Algorithmic bytes of one iteration of one problem with `ls` line-search trials. */
double mi_ilqr_bytes_per_iteration(int32_t n, int32_t m, int32_t N, int32_t ls);

/* LDS bytes one problem occupies in the wave-per-problem kernels (0 if unsupported). */
size_t mi_ilqr_lds_bytes(const mi_ilqr_desc* desc);

#ifdef __cplusplus
}
#endif
#endif /* MI_ILQR_H */
