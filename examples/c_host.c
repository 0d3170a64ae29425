/* A C host of libmi_ilqr.so: nothing but include/mi_ilqr.h, the HIP-free C ABI.
 *
 * The pendulum swing-up of /root/reference/pendulum.py (:18-34,85-94: T = 2 s, dt = 1e-2, Q = 0.01 diag(0,1),
 * R = 0.01, Qf = 100 I, delta = 1e-2, beta = 0.95) for a small batch of initial states: create, set the cost and the
 * initial conditions, solve, read the results back into page-locked buffers with ONE synchronization.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/c_host.c -Ldrake_ddp_amd/lib -lmi_ilqr -Wl,-rpath,$PWD/drake_ddp_amd/lib -lm -o c_host
 *   ./c_host            prints one line per problem: iterations, line-search trials, status, cost, final angle
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mi_ilqr.h"

#define CHECK(call)                                                                            \
  do {                                                                                         \
    int rc_ = (call);                                                                          \
    if (rc_ != MI_ILQR_OK) { fprintf(stderr, "%s: %s\n", #call, mi_ilqr_strerror(rc_)); return 1; } \
  } while (0)

int main(void) {
  enum { B = 4, N = 200, n = 2, m = 1 };
  const double dt = 1e-2, pi = 3.14159265358979323846;
  if (mi_ilqr_abi_version() != MI_ILQR_ABI_VERSION) { fprintf(stderr, "header / library ABI mismatch\n"); return 1; }

  mi_ilqr_desc d;
  memset(&d, 0, sizeof d);
  d.n = n; d.m = m; d.N = N; d.B = B;
  d.model_id = MI_MODEL_PENDULUM;
  CHECK(mi_ilqr_model_info(d.model_id, NULL, NULL, &d.n_params, d.model_params));   /* the registry's default parameters */
  d.dt = dt; d.delta = 1e-2; d.beta = 0.95; d.gamma = 0.0;
  d.keypoint_method = MI_KP_SET_INTERVAL; d.minN = 1;                                /* derivatives at every step (ilqr.py:97-100) */
  d.jacobian_mode = MI_JAC_FD_CENTRAL; d.fd_step = 1e-5;
  d.max_iters = 1000; d.hist_cap = 16; d.device_id = 0; d.kernel_mode = MI_KERNEL_AUTO;

  mi_ilqr_t* h = NULL;
  CHECK(mi_ilqr_create(&d, &h));

  const double Q[4] = {0.0, 0.0, 0.0, 0.01 * dt}, R[1] = {0.01 * dt}, Qf[4] = {100.0, 0.0, 0.0, 100.0}, x_nom[2] = {pi, 0.0};
  CHECK(mi_ilqr_set_cost(h, Q, R, Qf, x_nom));

  double x0[B * n];
  for (int b = 0; b < B; ++b) { x0[2 * b] = 0.3 * b - 0.4; x0[2 * b + 1] = 0.1 * b; }
  double u_guess[m * (N - 1)];
  memset(u_guess, 0, sizeof u_guess);
  CHECK(mi_ilqr_set_initial_shared(h, x0, u_guess));      /* one guess for the whole batch, like SetInitialGuess */

  /* page-locked result buffers.  x_bar, u_bar and the costs are the handle's RESULT SINK: the solve kernel writes them
   * there itself, problem by problem as each finishes; the integer results are copy-outs enqueued behind the solve.
   * One synchronization (collect_stats) covers it all. */
  void *xb = NULL, *ub = NULL, *cost = NULL, *iters = NULL, *status = NULL, *trials = NULL;
  CHECK(mi_ilqr_host_alloc(sizeof(double) * B * n * N, &xb));
  CHECK(mi_ilqr_host_alloc(sizeof(double) * B * m * (N - 1), &ub));
  CHECK(mi_ilqr_host_alloc(sizeof(double) * B, &cost));
  CHECK(mi_ilqr_set_result_sink(h, (double*)xb, (double*)ub, (double*)cost));
  CHECK(mi_ilqr_host_alloc(sizeof(int32_t) * B, &iters));
  CHECK(mi_ilqr_host_alloc(sizeof(int32_t) * B, &status));
  CHECK(mi_ilqr_host_alloc(sizeof(int32_t) * B, &trials));
  CHECK(mi_ilqr_solve_async(h));
  CHECK(mi_ilqr_get_async(h, MI_I_ITERS, iters, sizeof(int32_t) * B));
  CHECK(mi_ilqr_get_async(h, MI_I_STATUS, status, sizeof(int32_t) * B));
  CHECK(mi_ilqr_get_async(h, MI_I_LS_TRIALS, trials, sizeof(int32_t) * B));
  mi_ilqr_stats st;
  CHECK(mi_ilqr_collect_stats(h, &st));

  for (int b = 0; b < B; ++b) {
    const double* x = (const double*)xb + (size_t)b * n * N;          /* x_bar[b] is (n, N), time last */
    printf("problem %d: iterations %d trials %d status %d cost %.12g theta_N %.9f\n", b, ((int32_t*)iters)[b],
           ((int32_t*)trials)[b], ((int32_t*)status)[b], ((double*)cost)[b], x[0 * N + (N - 1)]);
  }
  printf("batch: %lld iterations, %d converged, best cost %.12g (problem %d), kernel %.3f ms\n", (long long)st.total_iters,
         st.n_converged, st.best_cost, st.best_index, st.kernel_ms);

  CHECK(mi_ilqr_set_result_sink(h, NULL, NULL, NULL));
  mi_ilqr_host_free(xb); mi_ilqr_host_free(ub); mi_ilqr_host_free(cost); mi_ilqr_host_free(iters); mi_ilqr_host_free(status); mi_ilqr_host_free(trials);
  mi_ilqr_destroy(h);
  return st.n_converged == B ? 0 : 2;
}
