/*
 * ORACLE (test infrastructure / reported CPU baseline) — plain-C restatement of the
 * iLQR hot path of vincekurtz/drake_ddp for the build-owned models.
 *
 * Follows /root/reference/ilqr.py function by function (lines cited below); pinned
 * against the NumPy oracle (oracle/ilqr_np.py, itself pinned against the unmodified
 * reference through tests/golden/) by tests/test_c_oracle.py.  Jacobians use the
 * same central finite differences as the HIP path.  Covered: Solve() from a cold start
 * or from persistent solver state (the receding-horizon loops of acrobot.py / mini_cheetah.py,
 * oracle_mpc_batch), key-point method 'setInterval' (any minN, with interpolation).  Problems of
 * a batch are independent and run on OpenMP threads.
 *
 * NOT part of the product: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXN 36
#define MAXM 12

typedef struct {
  int n, m, N, model_id;
  double params[16];
  double dt, delta, beta, gamma;
  int minN;          /* setInterval spacing (ilqr.py:417-432) */
  double fd_h;
  int max_iters;
} oracle_cfg;

/* ---- models: same formulas / operation order as oracle/models_np.py ---- */
static double softplus(double z) { return z > 0.0 ? z + log1p(exp(-z)) : log1p(exp(z)); }

static void step(const oracle_cfg* c, const double* x, const double* u, double* xn) {
  const double* p = c->params;
  const double dt = c->dt;
  switch (c->model_id) {
    case 0: { /* pendulum */
      const double acc = (u[0] - p[1] * x[1] - p[2] * sin(x[0])) / p[0];
      const double wn = x[1] + dt * acc;
      xn[0] = x[0] + dt * wn; xn[1] = wn;
      break;
    }
    case 1: { /* acrobot */
      const double m1 = p[0], m2 = p[1], l1 = p[2], lc1 = p[3], lc2 = p[4], Ic1 = p[5], Ic2 = p[6], b1 = p[7], b2 = p[8], g = p[9];
      const double q1 = x[0], q2 = x[1], v1 = x[2], v2 = x[3];
      const double I1 = Ic1 + m1 * lc1 * lc1, I2 = Ic2 + m2 * lc2 * lc2;
      const double s1 = sin(q1), s2 = sin(q2), c2 = cos(q2), s12 = sin(q1 + q2), h = m2 * l1 * lc2;
      const double M11 = I1 + I2 + m2 * l1 * l1 + 2.0 * h * c2, M12 = I2 + h * c2, M22 = I2;
      const double cb1 = -2.0 * h * s2 * v2 * v1 - h * s2 * v2 * v2, cb2 = h * s2 * v1 * v1;
      const double g1 = g * (m1 * lc1 + m2 * l1) * s1 + g * m2 * lc2 * s12, g2 = g * m2 * lc2 * s12;
      const double r1 = -cb1 - g1 - b1 * v1, r2 = u[0] - cb2 - g2 - b2 * v2;
      const double det = M11 * M22 - M12 * M12;
      const double a1 = (M22 * r1 - M12 * r2) / det, a2 = (M11 * r2 - M12 * r1) / det;
      const double v1n = v1 + dt * a1, v2n = v2 + dt * a2;
      xn[0] = q1 + dt * v1n; xn[1] = q2 + dt * v2n; xn[2] = v1n; xn[3] = v2n;
      break;
    }
    case 2: case 3: { /* cart-pole (+ wall) */
      const double mc = p[0], mp = p[1], l = p[2], g = p[3];
      const double px = x[0], th = x[1], vx = x[2], w = x[3];
      const double s = sin(th), cth = cos(th);
      const double M11 = mc + mp, M12 = mp * l * cth, M22 = mp * l * l;
      double r1 = u[0] + mp * l * w * w * s, r2 = -mp * g * l * s;
      if (c->model_id == 3) {
        const double tip = px + l * s, phi = tip - p[5] - p[4];
        const double F = p[6] * p[7] * softplus(-phi / p[7]);
        r1 = r1 + F; r2 = r2 + F * l * cth;
      }
      const double det = M11 * M22 - M12 * M12;
      const double a1 = (M22 * r1 - M12 * r2) / det, a2 = (M11 * r2 - M12 * r1) / det;
      const double vxn = vx + dt * a1, wn = w + dt * a2;
      xn[0] = px + dt * vxn; xn[1] = th + dt * wn; xn[2] = vxn; xn[3] = wn;
      break;
    }
    default: { /* synth36 */
      const double ks = p[0], cd = p[1], kc = p[2], bu = p[3];
      const int nq = 18;
      for (int i = 0; i < nq; ++i) {
        double a = -ks * sin(x[i]) - cd * x[nq + i];
        if (i < nq - 1) a = a + kc * sin(x[i + 1] - x[i]);
        if (i > 0) a = a - kc * sin(x[i] - x[i - 1]);
        if (i >= 6) a = a + u[i - 6]; else a = a + bu * (u[2 * i] - u[2 * i + 1]);
        const double vn = x[nq + i] + dt * a;
        xn[nq + i] = vn; xn[i] = x[i] + dt * vn;
      }
    }
  }
}

/* time-last accessors, exactly the reference's array layout */
#define X(a, i, t) (a)[(i) * N + (t)]
#define U(a, k, t) (a)[(k) * (N - 1) + (t)]
#define KK(a, k, j, t) (a)[((k) * n + (j)) * (N - 1) + (t)]
#define FX(a, i, j, t) (a)[((i) * n + (j)) * (N - 1) + (t)]
#define FU(a, i, k, t) (a)[((i) * m + (k)) * (N - 1) + (t)]

typedef struct {
  double *x_bar, *u_bar, *K, *kappa, *dV, *fx, *fu, *x, *u;
} work;

/* one line-search trial: ilqr.py:306-327 */
static double rollout(const oracle_cfg* c, const double* Q, const double* R, const double* Qf, const double* xnom,
                      const double* x0, const work* w, double eps, double* expected) {
  const int n = c->n, m = c->m, N = c->N;
  double L = 0.0, ex = 0.0, xt[MAXN], ut[MAXM], xn[MAXN];
  for (int i = 0; i < n; ++i) { xt[i] = x0[i]; X(w->x, i, 0) = x0[i]; }
  for (int t = 0; t < N - 1; ++t) {
    for (int k = 0; k < m; ++k) {
      double acc = 0.0;
      for (int j = 0; j < n; ++j) acc += KK(w->K, k, j, t) * (xt[j] - X(w->x_bar, j, t));
      ut[k] = U(w->u_bar, k, t) - eps * U(w->kappa, k, t) - acc;     /* :313 */
      U(w->u, k, t) = ut[k];
    }
    step(c, xt, ut, xn);                                              /* :316 */
    double q = 0.0, r = 0.0;
    for (int i = 0; i < n; ++i) { double s = 0.0; for (int j = 0; j < n; ++j) s += Q[i * n + j] * (xt[j] - xnom[j]); q += (xt[i] - xnom[i]) * s; }
    for (int i = 0; i < m; ++i) { double s = 0.0; for (int j = 0; j < m; ++j) s += R[i * m + j] * ut[j]; r += ut[i] * s; }
    L += q + r;                                                       /* :325 */
    ex += -eps * (1 - eps / 2) * w->dV[t];                            /* :326 */
    for (int i = 0; i < n; ++i) { xt[i] = xn[i]; X(w->x, i, t + 1) = xn[i]; }
  }
  double q = 0.0;
  for (int i = 0; i < n; ++i) { double s = 0.0; for (int j = 0; j < n; ++j) s += Qf[i * n + j] * (xt[j] - xnom[j]); q += (xt[i] - xnom[i]) * s; }
  *expected = ex;
  return L + q;                                                       /* :327 */
}

/* explicit inverse, Gauss-Jordan with partial pivoting (the reference calls np.linalg.inv, :655) */
static void invert(int m, const double* A, double* Ai) {
  double a[MAXM][2 * MAXM];
  for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) { a[i][j] = A[i * m + j]; a[i][m + j] = (i == j); }
  for (int k = 0; k < m; ++k) {
    int piv = k;
    for (int i = k + 1; i < m; ++i) if (fabs(a[i][k]) > fabs(a[piv][k])) piv = i;
    if (piv != k) for (int j = 0; j < 2 * m; ++j) { double t = a[k][j]; a[k][j] = a[piv][j]; a[piv][j] = t; }
    const double d = a[k][k];
    for (int j = 0; j < 2 * m; ++j) a[k][j] /= d;
    for (int i = 0; i < m; ++i) if (i != k) { const double f = a[i][k]; if (f != 0.0) for (int j = 0; j < 2 * m; ++j) a[i][j] -= f * a[k][j]; }
  }
  for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) Ai[i * m + j] = a[i][m + j];
}

/* _get_derivatives with setInterval key-points + interpolation: ilqr.py:380-432, 596-621 */
static void linearize(const oracle_cfg* c, const work* w) {
  const int n = c->n, m = c->m, N = c->N, minN = c->minN;
  const double h = c->fd_h, inv2h = 1.0 / (2.0 * h);
  int nk = (N - 2) / minN + 1;
  int* kp = (int*)malloc(sizeof(int) * nk);
  for (int i = 0; i < nk; ++i) kp[i] = i * minN;
  if (kp[nk - 1] != N - 2) kp[nk - 1] = N - 2;                        /* overwrite, not append (:428-430) */
  double xt[MAXN], ut[MAXM], xp[MAXN], up[MAXM], fp[MAXN], fm[MAXN];
  for (int q = 0; q < nk; ++q) {
    const int t = kp[q];
    for (int i = 0; i < n; ++i) xt[i] = X(w->x, i, t);
    for (int k = 0; k < m; ++k) ut[k] = U(w->u, k, t);
    for (int col = 0; col < n + m; ++col) {
      memcpy(xp, xt, sizeof(double) * n); memcpy(up, ut, sizeof(double) * m);
      if (col < n) xp[col] = xt[col] + h; else up[col - n] = ut[col - n] + h;
      step(c, xp, up, fp);
      if (col < n) xp[col] = xt[col] - h; else up[col - n] = ut[col - n] - h;
      step(c, xp, up, fm);
      for (int i = 0; i < n; ++i) {
        const double d = (fp[i] - fm[i]) * inv2h;
        if (col < n) FX(w->fx, i, col, t) = d; else FU(w->fu, i, col - n, t) = d;
      }
    }
  }
  if (minN != 1) {                                                    /* :414 */
    for (int q = 0; q + 1 < nk; ++q) {
      const int s = kp[q], e = kp[q + 1];
      for (int j = s + 1; j < e; ++j) {
        for (int r = 0; r < n * n; ++r) { const double fs = w->fx[r * (N - 1) + s], fe = w->fx[r * (N - 1) + e]; w->fx[r * (N - 1) + j] = fs + (fe - fs) * (j - s) / (e - s); }
        for (int r = 0; r < n * m; ++r) { const double fs = w->fu[r * (N - 1) + s], fe = w->fu[r * (N - 1) + e]; w->fu[r * (N - 1) + j] = fs + (fe - fs) * (j - s) / (e - s); }
      }
    }
  }
  free(kp);
}

/* _backward_pass with the cost partials folded in: ilqr.py:161-206, 623-667 */
static void backward(const oracle_cfg* c, const double* Q, const double* R, const double* Qf, const double* xnom, const work* w) {
  const int n = c->n, m = c->m, N = c->N;
  double Vx[MAXN], Vxx[MAXN * MAXN], fx[MAXN * MAXN], fu[MAXN * MAXM], A[MAXN * MAXN], Bm[MAXM * MAXN];
  double Qx[MAXN], Qu[MAXM], Qxx[MAXN * MAXN], Quu[MAXM * MAXM], Qux[MAXM * MAXN], Qi[MAXM * MAXM], QuQi[MAXM], T[MAXN * MAXM];
  for (int i = 0; i < n; ++i) {
    double s = 0.0, g = 0.0;
    for (int j = 0; j < n; ++j) { s += 2 * Qf[i * n + j] * X(w->x_bar, j, N - 1); g += 2 * xnom[j] * Qf[j * n + i]; Vxx[i * n + j] = 2 * Qf[i * n + j]; }
    Vx[i] = s - g;                                                    /* :203-204 */
  }
  for (int t = N - 2; t >= 0; --t) {
    for (int i = 0; i < n; ++i) { for (int j = 0; j < n; ++j) fx[i * n + j] = FX(w->fx, i, j, t); for (int k = 0; k < m; ++k) fu[i * m + k] = FU(w->fu, i, k, t); }
    for (int i = 0; i < n; ++i) {
      double s = 0.0, g = 0.0, f = 0.0;
      for (int j = 0; j < n; ++j) { s += 2 * Q[i * n + j] * X(w->x_bar, j, t); g += 2 * xnom[j] * Q[j * n + i]; f += fx[j * n + i] * Vx[j]; }
      Qx[i] = (s - g) + f;                                            /* :651 */
    }
    for (int a = 0; a < m; ++a) {
      double s = 0.0, f = 0.0;
      for (int j = 0; j < m; ++j) s += 2 * R[a * m + j] * U(w->u_bar, j, t);
      for (int j = 0; j < n; ++j) f += fu[j * m + a] * Vx[j];
      Qu[a] = s + f;                                                  /* :652 */
    }
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = 0.0; for (int k = 0; k < n; ++k) s += fx[k * n + i] * Vxx[k * n + j]; A[i * n + j] = s; }
    for (int a = 0; a < m; ++a) for (int j = 0; j < n; ++j) { double s = 0.0; for (int k = 0; k < n; ++k) s += fu[k * m + a] * Vxx[k * n + j]; Bm[a * n + j] = s; }
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = 0.0; for (int k = 0; k < n; ++k) s += A[i * n + k] * fx[k * n + j]; Qxx[i * n + j] = 2 * Q[i * n + j] + s; }   /* :653 */
    for (int a = 0; a < m; ++a) {
      for (int b = 0; b < m; ++b) { double s = 0.0; for (int k = 0; k < n; ++k) s += Bm[a * n + k] * fu[k * m + b]; Quu[a * m + b] = 2 * R[a * m + b] + s; }  /* :654 */
      for (int j = 0; j < n; ++j) { double s = 0.0; for (int k = 0; k < n; ++k) s += Bm[a * n + k] * fx[k * n + j]; Qux[a * n + j] = s; }                      /* :656 */
    }
    invert(m, Quu, Qi);                                               /* :655 */
    double dv = 0.0;
    for (int a = 0; a < m; ++a) {
      double s = 0.0, q = 0.0;
      for (int b = 0; b < m; ++b) { s += Qi[a * m + b] * Qu[b]; q += Qu[b] * Qi[b * m + a]; }
      U(w->kappa, a, t) = s;                                          /* :659 */
      QuQi[a] = q;
      for (int j = 0; j < n; ++j) { double g = 0.0; for (int b = 0; b < m; ++b) g += Qi[a * m + b] * Qux[b * n + j]; KK(w->K, a, j, t) = g; }   /* :660 */
    }
    for (int a = 0; a < m; ++a) dv += QuQi[a] * Qu[a];
    w->dV[t] = dv;                                                    /* :663 */
    for (int j = 0; j < n; ++j) { double s = 0.0; for (int a = 0; a < m; ++a) s += QuQi[a] * Qux[a * n + j]; Vx[j] = Qx[j] - s; }          /* :666 */
    for (int i = 0; i < n; ++i) for (int b = 0; b < m; ++b) { double s = 0.0; for (int a = 0; a < m; ++a) s += Qux[a * n + i] * Qi[a * m + b]; T[i * m + b] = s; }
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = 0.0; for (int b = 0; b < m; ++b) s += T[i * m + b] * Qux[b * n + j]; Vxx[i * n + j] = Qxx[i * n + j] - s; }  /* :667 */
  }
}

/* The persistent solver state of a freshly constructed reference object: all zero (ilqr.py:70-83). */
static void fresh_state(const oracle_cfg* c, work* w) {
  const int n = c->n, m = c->m, N = c->N;
  memset(w->x_bar, 0, sizeof(double) * n * N);
  memset(w->u_bar, 0, sizeof(double) * m * (N - 1));
  memset(w->K, 0, sizeof(double) * m * n * (N - 1));
  memset(w->kappa, 0, sizeof(double) * m * (N - 1));
  memset(w->dV, 0, sizeof(double) * (N - 1));
  memset(w->fx, 0, sizeof(double) * n * n * (N - 1));
  memset(w->fu, 0, sizeof(double) * n * m * (N - 1));
}

/* Solve (ilqr.py:669-710) for one problem FROM THE STATE IN `w` (x_bar, u_bar, K, kappa, dV persist across
 * calls exactly as the attributes of the reference object do - SURVEY F10: the first rollout of a re-solve
 * applies the previous solve's gains about the previous x_bar and is accepted unconditionally, L_last = inf).
 * status: 0 ok, 1 max_iters, 2 linesearch failed */
static int solve_one(const oracle_cfg* c, const double* Q, const double* R, const double* Qf, const double* xnom,
                     const double* x0, work* w, double* cost, int* iters, int* ls_trials) {
  const int n = c->n, m = c->m, N = c->N;
  double L = INFINITY, improvement = INFINITY;
  int it = 0, ls = 0, status = 0;
  while (improvement > c->delta) {
    if (it >= c->max_iters) { status = 1; break; }
    double eps = 1.0, Lnew = 0.0, ex;
    int accepted = 0;
    while (eps >= 1e-8) {                                             /* :302 */
      ls++;
      Lnew = rollout(c, Q, R, Qf, xnom, x0, w, eps, &ex);
      if (L - Lnew > c->gamma * ex) { accepted = 1; break; }          /* :330-331 */
      eps *= c->beta;                                                 /* :335 */
    }
    if (!accepted) { status = 2; break; }
    linearize(c, w);                                                  /* :370 */
    memcpy(w->x_bar, w->x, sizeof(double) * n * N);                   /* :375-376 */
    memcpy(w->u_bar, w->u, sizeof(double) * m * (N - 1));
    backward(c, Q, R, Qf, xnom, w);                                   /* :697 */
    improvement = L - Lnew;                                           /* :706 */
    L = Lnew;
    it++;
  }
  *cost = L; *iters = it; *ls_trials = ls;
  return status;
}

static void work_alloc(work* w, int n, int m, int N) {
  w->x_bar = (double*)malloc(sizeof(double) * n * N); w->x = (double*)malloc(sizeof(double) * n * N);
  w->u_bar = (double*)malloc(sizeof(double) * m * (N - 1)); w->u = (double*)malloc(sizeof(double) * m * (N - 1));
  w->K = (double*)malloc(sizeof(double) * m * n * (N - 1)); w->kappa = (double*)malloc(sizeof(double) * m * (N - 1));
  w->dV = (double*)malloc(sizeof(double) * (N - 1));
  w->fx = (double*)malloc(sizeof(double) * n * n * (N - 1)); w->fu = (double*)malloc(sizeof(double) * n * m * (N - 1));
}
static void work_free(work* w) { free(w->x_bar); free(w->x); free(w->u_bar); free(w->u); free(w->K); free(w->kappa); free(w->dV); free(w->fx); free(w->fu); }

/* Batched cold-start solve; outputs may be NULL.  Returns the number of threads used. */
int oracle_solve_batch(const oracle_cfg* c, int B, const double* Q, const double* R, const double* Qf, const double* xnom,
                       const double* x0, const double* u_guess, double* x_bar, double* u_bar, double* K, double* kappa,
                       double* cost, int* iters, int* ls_trials, int* status, int nthreads) {
  const int n = c->n, m = c->m, N = c->N;
  if (n > MAXN || m > MAXM) return -1;
  int used = 1;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel
#endif
  {
    work w;
    work_alloc(&w, n, m, N);
#ifdef _OPENMP
#pragma omp single
    used = omp_get_num_threads();
#pragma omp for schedule(dynamic, 1)
#endif
    for (int b = 0; b < B; ++b) {
      double L; int it, ls;
      fresh_state(c, &w);
      if (u_guess) memcpy(w.u_bar, u_guess + (size_t)b * m * (N - 1), sizeof(double) * m * (N - 1));   /* SetInitialGuess (:148-156) */
      const int st = solve_one(c, Q, R, Qf, xnom, x0 + (size_t)b * n, &w, &L, &it, &ls);
      if (cost) cost[b] = L;
      if (iters) iters[b] = it;
      if (ls_trials) ls_trials[b] = ls;
      if (status) status[b] = st;
      if (x_bar) memcpy(x_bar + (size_t)b * n * N, w.x_bar, sizeof(double) * n * N);
      if (u_bar) memcpy(u_bar + (size_t)b * m * (N - 1), w.u_bar, sizeof(double) * m * (N - 1));
      if (K) memcpy(K + (size_t)b * m * n * (N - 1), w.K, sizeof(double) * m * n * (N - 1));
      if (kappa) memcpy(kappa + (size_t)b * m * (N - 1), w.kappa, sizeof(double) * m * (N - 1));
    }
    work_free(&w);
  }
  return used;
}

/* The receding-horizon loop of the reference's callers (acrobot.py:131-162, mini_cheetah.py:186-213) for every
 * problem of a batch: a cold Solve() from u_guess, then `resolves` times
 *     u_guess <- [u[:, replan:], repeat(u[:, -1], replan)];  x0 <- x[:, replan]     (acrobot.py:147-152)
 *     x_nom  += target_step (NULL = fixed target; mini_cheetah.py:151-156)
 *     Solve() on the SAME solver object: gains, x_bar and dV persist.
 * log: (B, resolves, n+2) = x0 of the re-solve | cost | iterations, like mi_ilqr_get_mpc_log; first: (B,2) cost and
 * iterations of the cold solve; the final x_bar/u_bar/K/kappa are optional.  Returns the threads used. */
int oracle_mpc_batch(const oracle_cfg* c, int B, const double* Q, const double* R, const double* Qf, const double* xnom,
                     const double* x0, const double* u_guess, int resolves, int replan, const double* target_step,
                     double* log, double* first, double* x_bar, double* u_bar, double* K, double* kappa,
                     int* ls_trials, int* status, int nthreads) {
  const int n = c->n, m = c->m, N = c->N;
  if (n > MAXN || m > MAXM || replan < 1 || replan >= N - 1) return -1;
  int used = 1;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel
#endif
  {
    work w;
    work_alloc(&w, n, m, N);
#ifdef _OPENMP
#pragma omp single
    used = omp_get_num_threads();
#pragma omp for schedule(dynamic, 1)
#endif
    for (int b = 0; b < B; ++b) {
      double L, x0b[MAXN], xn[MAXN], ush[MAXM];
      int it, ls, ls_sum = 0, st;
      memcpy(x0b, x0 + (size_t)b * n, sizeof(double) * n);
      memcpy(xn, xnom, sizeof(double) * n);
      fresh_state(c, &w);
      if (u_guess) memcpy(w.u_bar, u_guess + (size_t)b * m * (N - 1), sizeof(double) * m * (N - 1));
      st = solve_one(c, Q, R, Qf, xn, x0b, &w, &L, &it, &ls);
      ls_sum += ls;
      if (first) { first[2 * b] = L; first[2 * b + 1] = (double)it; }
      for (int r = 0; r < resolves && st != 2; ++r) {
        for (int i = 0; i < n; ++i) x0b[i] = X(w.x_bar, i, replan);
        for (int k = 0; k < m; ++k) {
          ush[k] = U(w.u_bar, k, N - 2);
          for (int t = 0; t < N - 1; ++t) U(w.u_bar, k, t) = (t + replan < N - 1) ? U(w.u_bar, k, t + replan) : ush[k];
        }
        if (target_step) for (int i = 0; i < n; ++i) xn[i] += target_step[i];
        st = solve_one(c, Q, R, Qf, xn, x0b, &w, &L, &it, &ls);
        ls_sum += ls;
        if (log) {
          double* lg = log + ((size_t)b * resolves + r) * (n + 2);
          for (int i = 0; i < n; ++i) lg[i] = x0b[i];
          lg[n] = L; lg[n + 1] = (double)it;
        }
      }
      if (ls_trials) ls_trials[b] = ls_sum;
      if (status) status[b] = st;
      if (x_bar) memcpy(x_bar + (size_t)b * n * N, w.x_bar, sizeof(double) * n * N);
      if (u_bar) memcpy(u_bar + (size_t)b * m * (N - 1), w.u_bar, sizeof(double) * m * (N - 1));
      if (K) memcpy(K + (size_t)b * m * n * (N - 1), w.K, sizeof(double) * m * n * (N - 1));
      if (kappa) memcpy(kappa + (size_t)b * m * (N - 1), w.kappa, sizeof(double) * m * (N - 1));
    }
    work_free(&w);
  }
  return used;
}
