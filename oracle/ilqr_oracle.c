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

#define MAXN 40
#define MAXM 12

typedef struct {
  int n, m, N, model_id;
  double params[16];
  double dt, delta, beta, gamma;
  int minN;          /* setInterval spacing (ilqr.py:417-432) */
  double fd_h;
  int max_iters;
  /* key-point method (ilqr.py:393-404): 0 setInterval, 1 adaptiveJerk, 2 iterativeError; zero-initialized = setInterval */
  int kp_method, maxN;
  double jerk_thr, err_thr;
} oracle_cfg;

/* ---- models: same formulas / operation order as oracle/models_np.py ---- */
static double softplus(double z) { return z > 0.0 ? z + log1p(exp(-z)) : log1p(exp(z)); }

/* ---- PLANAR_QUAD (model 5): planar floating-base quadruped, articulated-body algorithm in world-aligned planar
 * coordinates - the same formulas, body table and operation order as oracle/models_np.py:quad_accel ---- */
#define QNB 16
static const int q_parent[QNB] = {-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14};
static const double q_len[QNB] = {0.0, 0.20, 0.18, 0.14, 0.20, 0.18, 0.14, 0.20, 0.18, 0.14, 0.20, 0.18, 0.14, 0.12, 0.12, 0.12};
static const double q_mass[QNB] = {4.0, 0.60, 0.40, 0.30, 0.60, 0.40, 0.30, 0.60, 0.40, 0.30, 0.60, 0.40, 0.30, 0.10, 0.10, 0.10};
static const double q_atx[QNB] = {0, 0.19, 0, 0, 0.19, 0, 0, -0.19, 0, 0, -0.19, 0, 0, -0.25, 0, 0};
static const double q_atz[QNB] = {0, 0.0, 0, 0, 0.0, 0, 0, 0.0, 0, 0, 0.0, 0, 0, 0.02, 0, 0};
static const double q_tail_rest[3] = {-1.2, -0.2, -0.2};

static void quad_accel(const double* x, const double* u, const double* p, double* qdd) {
  const double g = p[0], kc = p[1], sig = p[2], dn = p[3], mu = p[4], b_leg = p[5], b_tail = p[6], k_tail = p[7];
  const int nq = 18;
  const double *q = x, *v = x + nq;
  double th[QNB], om[QNB], px[QNB], pz[QNB], vx[QNB], vz[QNB], sn[QNB], cs[QNB], rx[QNB], rz[QNB];
  double dx[QNB], dz[QNB], cbx[QNB], cbz[QNB], tau[QNB], inert[QNB];
  double J[QNB], hx[QNB], hz[QNB], mxx[QNB], mxz[QNB], mzz[QNB], bn[QNB], bx[QNB], bz[QNB];
  double Dj[QNB], Ux[QNB], Uz[QNB], uu[QNB], al[QNB], acx[QNB], acz[QNB];
  inert[0] = 0.06;
  for (int i = 1; i < QNB; ++i) inert[i] = q_mass[i] * q_len[i] * q_len[i] / 12;
  th[0] = q[2]; om[0] = v[2]; px[0] = q[0]; pz[0] = q[1]; vx[0] = v[0]; vz[0] = v[1];
  sn[0] = sin(th[0]); cs[0] = cos(th[0]); rx[0] = 0.0; rz[0] = 0.0;
  dx[0] = dz[0] = cbx[0] = cbz[0] = tau[0] = 0.0;
  for (int i = 1; i < QNB; ++i) {
    const int par = q_parent[i];
    if (par != 0) { const double lp = q_len[par]; dx[i] = lp * sn[par]; dz[i] = -lp * cs[par]; }
    else { dx[i] = cs[0] * q_atx[i] - sn[0] * q_atz[i]; dz[i] = sn[0] * q_atx[i] + cs[0] * q_atz[i]; }
    th[i] = th[par] + q[2 + i];
    om[i] = om[par] + v[2 + i];
    sn[i] = sin(th[i]); cs[i] = cos(th[i]);
    px[i] = px[par] + dx[i]; pz[i] = pz[par] + dz[i];
    vx[i] = vx[par] - om[par] * dz[i]; vz[i] = vz[par] + om[par] * dx[i];
    const double hl = 0.5 * q_len[i];
    rx[i] = hl * sn[i]; rz[i] = -hl * cs[i];
    const double w2 = om[par] * om[par];
    cbx[i] = -w2 * dx[i]; cbz[i] = -w2 * dz[i];
  }
  for (int i = 1; i < 13; ++i) tau[i] = u[i - 1] - b_leg * v[2 + i];
  for (int k = 0; k < 3; ++k) { const int i = 13 + k; tau[i] = -b_tail * v[2 + i] - k_tail * (q[2 + i] - q_tail_rest[k]); }
  for (int i = 0; i < QNB; ++i) {
    const double m_ = q_mass[i], w2 = om[i] * om[i];
    J[i] = inert[i] + m_ * (rx[i] * rx[i] + rz[i] * rz[i]);
    hx[i] = -m_ * rz[i]; hz[i] = m_ * rx[i];
    mxx[i] = m_; mxz[i] = 0.0; mzz[i] = m_;
    bn[i] = m_ * g * rx[i];
    bx[i] = -m_ * w2 * rx[i];
    bz[i] = -m_ * w2 * rz[i] + m_ * g;
  }
  static const int tips[5] = {3, 6, 9, 12, 15};
  for (int c = 0; c < 5; ++c) {
    const int i = tips[c];
    const double ex = 2.0 * rx[i], ez = 2.0 * rz[i];
    const double tz = pz[i] + ez;
    const double tvx = vx[i] - om[i] * ez, tvz = vz[i] + om[i] * ex;
    const double fn0 = kc * sig * softplus(-tz / sig);
    const double fn = fn0 * (1.0 - dn * tvz);
    const double ft = -mu * fn0 * tvx;
    bn[i] = bn[i] - (ex * fn - ez * ft);
    bx[i] = bx[i] - ft;
    bz[i] = bz[i] - fn;
  }
  for (int i = QNB - 1; i >= 1; --i) {
    const int par = q_parent[i];
    Dj[i] = J[i]; Ux[i] = hx[i]; Uz[i] = hz[i];
    uu[i] = tau[i] - bn[i];
    const double invD = 1.0 / Dj[i];
    const double exx = mxx[i] - Ux[i] * Ux[i] * invD;
    const double exz = mxz[i] - Ux[i] * Uz[i] * invD;
    const double ezz = mzz[i] - Uz[i] * Uz[i] * invD;
    const double s_ = uu[i] * invD;
    const double fx = bx[i] + exx * cbx[i] + exz * cbz[i] + Ux[i] * s_;
    const double fz = bz[i] + exz * cbx[i] + ezz * cbz[i] + Uz[i] * s_;
    const double gx = -exx * dz[i] + exz * dx[i];
    const double gz = -exz * dz[i] + ezz * dx[i];
    J[par] = J[par] + (-dz[i] * gx + dx[i] * gz);
    hx[par] = hx[par] + gx; hz[par] = hz[par] + gz;
    mxx[par] = mxx[par] + exx; mxz[par] = mxz[par] + exz; mzz[par] = mzz[par] + ezz;
    bn[par] = bn[par] + tau[i] - dz[i] * fx + dx[i] * fz;
    bx[par] = bx[par] + fx; bz[par] = bz[par] + fz;
  }
  {
    const double a11 = mxx[0], a12 = mxz[0], a13 = hx[0], a22 = mzz[0], a23 = hz[0], a33 = J[0];
    const double r1 = -bx[0], r2 = -bz[0], r3 = -bn[0];
    const double l21 = a12 / a11, l31 = a13 / a11;
    const double d2 = a22 - l21 * a12, e23 = a23 - l21 * a13;
    const double l32 = e23 / d2;
    const double d3 = a33 - l31 * a13 - l32 * e23;
    const double y2 = r2 - l21 * r1;
    const double y3 = r3 - l31 * r1 - l32 * y2;
    const double alpha = y3 / d3;
    const double az = (y2 - e23 * alpha) / d2;
    const double ax = (r1 - a12 * az - a13 * alpha) / a11;
    al[0] = alpha; acx[0] = ax; acz[0] = az;
    qdd[0] = ax; qdd[1] = az; qdd[2] = alpha;
  }
  for (int i = 1; i < QNB; ++i) {
    const int par = q_parent[i];
    const double apx = acx[par] - al[par] * dz[i] + cbx[i];
    const double apz = acz[par] + al[par] * dx[i] + cbz[i];
    const double qi = (uu[i] - (Dj[i] * al[par] + Ux[i] * apx + Uz[i] * apz)) / Dj[i];
    qdd[2 + i] = qi;
    al[i] = al[par] + qi; acx[i] = apx; acz[i] = apz;
  }
}

/* ---- 3-D quadruped (model 6): the formulas and operation order of oracle/models_np.py:quad3d_leg / quad3d_step ---- */
#define Q3_L0 0.062
#define Q3_L1 0.209
#define Q3_L2 0.195
#define Q3_HIPX 0.19
#define Q3_HIPY 0.049
/* leg k: contact force in the world frame fw[3], its moment about the trunk's origin in the body frame tq[3], joint accelerations */
static void quad3d_leg(int k, const double R[3][3], const double* om, const double* vlin, double pz, const double* q, const double* jd,
                       const double* u, const double* p, double* fw, double* tq, double* jacc) {
  const double kc = p[1], sig = p[2], dn = p[3], mu = p[4], b_j = p[5];
  const double sx = k < 2 ? 1.0 : -1.0, sy = (k % 2) == 0 ? -1.0 : 1.0;
  const double a = q[0], b = q[1], c = q[2];
  const double sa = sin(a), ca = cos(a), sb = sin(b), cb = cos(b), sbc = sin(b + c), cbc = cos(b + c);
  const double X = -(Q3_L1 * sb + Q3_L2 * sbc), Z = -(Q3_L1 * cb + Q3_L2 * cbc), Y = sy * Q3_L0;
  const double fb[3] = {sx * Q3_HIPX + X, sy * Q3_HIPY + (Y * ca - Z * sa), Y * sa + Z * ca};
  const double Ja[3] = {0.0, -(Y * sa) - Z * ca, Y * ca - Z * sa};
  const double Jb[3] = {Z, X * sa, -(X * ca)};
  const double dXc = -(Q3_L2 * cbc), dZc = Q3_L2 * sbc;
  const double Jc[3] = {dXc, -(dZc * sa), dZc * ca};
  const double zf = pz + (R[2][0] * fb[0] + R[2][1] * fb[1] + R[2][2] * fb[2]);
  const double vb[3] = {om[1] * fb[2] - om[2] * fb[1] + (Ja[0] * jd[0] + Jb[0] * jd[1] + Jc[0] * jd[2]),
                        om[2] * fb[0] - om[0] * fb[2] + (Ja[1] * jd[0] + Jb[1] * jd[1] + Jc[1] * jd[2]),
                        om[0] * fb[1] - om[1] * fb[0] + (Ja[2] * jd[0] + Jb[2] * jd[1] + Jc[2] * jd[2])};
  double vf[3], fbd[3];
  for (int i = 0; i < 3; ++i) vf[i] = vlin[i] + (R[i][0] * vb[0] + R[i][1] * vb[1] + R[i][2] * vb[2]);
  const double fn0 = kc * sig * softplus(-zf / sig);
  fw[0] = -(mu * fn0) * vf[0]; fw[1] = -(mu * fn0) * vf[1]; fw[2] = fn0 * (1.0 - dn * vf[2]);
  for (int i = 0; i < 3; ++i) fbd[i] = R[0][i] * fw[0] + R[1][i] * fw[1] + R[2][i] * fw[2];
  tq[0] = fb[1] * fbd[2] - fb[2] * fbd[1]; tq[1] = fb[2] * fbd[0] - fb[0] * fbd[2]; tq[2] = fb[0] * fbd[1] - fb[1] * fbd[0];
  jacc[0] = (u[0] - b_j * jd[0] + (Ja[0] * fbd[0] + Ja[1] * fbd[1] + Ja[2] * fbd[2])) / p[11];
  jacc[1] = (u[1] - b_j * jd[1] + (Jb[0] * fbd[0] + Jb[1] * fbd[1] + Jb[2] * fbd[2])) / p[12];
  jacc[2] = (u[2] - b_j * jd[2] + (Jc[0] * fbd[0] + Jc[1] * fbd[1] + Jc[2] * fbd[2])) / p[13];
}
static void quad3d_step(const double* x, const double* u, const double* p, double dt, double* xn) {
  const double g = p[0], mt = p[7], Ix = p[8], Iy = p[9], Iz = p[10];
  const double qw = x[0], qx = x[1], qy = x[2], qz = x[3];
  const double *pos = x + 4, *jq = x + 7, *om = x + 19, *vl = x + 22, *jd = x + 25;
  const double s2 = 2.0 / (qw * qw + qx * qx + qy * qy + qz * qz);
  const double R[3][3] = {{1.0 - s2 * (qy * qy + qz * qz), s2 * (qx * qy - qw * qz), s2 * (qx * qz + qw * qy)},
                          {s2 * (qx * qy + qw * qz), 1.0 - s2 * (qx * qx + qz * qz), s2 * (qy * qz - qw * qx)},
                          {s2 * (qx * qz - qw * qy), s2 * (qy * qz + qw * qx), 1.0 - s2 * (qx * qx + qy * qy)}};
  double fw[4][3], tq[4][3], ja[4][3], F[3], T[3];
  for (int k = 0; k < 4; ++k) quad3d_leg(k, R, om, vl, pos[2], jq + 3 * k, jd + 3 * k, u + 3 * k, p, fw[k], tq[k], ja[k]);
  for (int i = 0; i < 3; ++i) { F[i] = (fw[0][i] + fw[2][i]) + (fw[1][i] + fw[3][i]); T[i] = (tq[0][i] + tq[2][i]) + (tq[1][i] + tq[3][i]); }
  const double al[3] = {F[0] / mt, F[1] / mt, F[2] / mt - g};
  const double aw[3] = {(T[0] - (Iz - Iy) * om[1] * om[2]) / Ix, (T[1] - (Ix - Iz) * om[2] * om[0]) / Iy, (T[2] - (Iy - Ix) * om[0] * om[1]) / Iz};
  double omn[3], vln[3];
  for (int i = 0; i < 3; ++i) { omn[i] = om[i] + dt * aw[i]; vln[i] = vl[i] + dt * al[i]; }
  const double hd = 0.5 * dt;
  xn[0] = qw + hd * (-(qx * omn[0]) - qy * omn[1] - qz * omn[2]);
  xn[1] = qx + hd * (qw * omn[0] + qy * omn[2] - qz * omn[1]);
  xn[2] = qy + hd * (qw * omn[1] + qz * omn[0] - qx * omn[2]);
  xn[3] = qz + hd * (qw * omn[2] + qx * omn[1] - qy * omn[0]);
  for (int i = 0; i < 3; ++i) { xn[4 + i] = pos[i] + dt * vln[i]; xn[19 + i] = omn[i]; xn[22 + i] = vln[i]; }
  for (int k = 0; k < 4; ++k) for (int i = 0; i < 3; ++i) {
    const double jdn = jd[3 * k + i] + dt * ja[k][i];
    xn[25 + 3 * k + i] = jdn; xn[7 + 3 * k + i] = jq[3 * k + i] + dt * jdn;
  }
}

/* ---- arm + ball (model 7): the formulas and operation order of oracle/models_np.py:arm27_kinematics / arm27_step ---- */
#define A27_H0 0.28
#define A27_L1 0.42
#define A27_L2 0.31
#define A27_L3 0.27
static void cross3(const double* a, const double* b, double* o) {
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
static void arm27_kinematics(const double* q, const double* p, double* hand, double* elbow, double axes[7][3], double orgs[7][3]) {
  double ex[3] = {1.0, 0.0, 0.0}, ey[3] = {0.0, 1.0, 0.0}, ez[3] = {0.0, 0.0, 1.0}, pos[3] = {0.0, 0.0, A27_H0};
  for (int i = 0; i < 7; ++i) {
    const double s = sin(q[i]), c = cos(q[i]);
    if (i % 2 == 0) {
      for (int k = 0; k < 3; ++k) { axes[i][k] = ez[k]; orgs[i][k] = pos[k]; }
      for (int k = 0; k < 3; ++k) { const double a = c * ex[k] + s * ey[k], b = c * ey[k] - s * ex[k]; ex[k] = a; ey[k] = b; }
    } else {
      for (int k = 0; k < 3; ++k) { axes[i][k] = ey[k]; orgs[i][k] = pos[k]; }
      for (int k = 0; k < 3; ++k) { const double a = c * ex[k] - s * ez[k], b = c * ez[k] + s * ex[k]; ex[k] = a; ez[k] = b; }
    }
    if (i == 2) { for (int k = 0; k < 3; ++k) { pos[k] = pos[k] + A27_L1 * ez[k]; elbow[k] = pos[k]; } }
    else if (i == 4) { for (int k = 0; k < 3; ++k) pos[k] = pos[k] + A27_L2 * ez[k]; }
  }
  for (int k = 0; k < 3; ++k) hand[k] = pos[k] + (p[14] * ex[k] + A27_L3 * ez[k]);
}
static void arm27_step(const double* x, const double* u, const double* p, double dt, double* xn) {
  const double g = p[0], kc = p[1], sig = p[2], dn = p[3], mu = p[4], bj = p[5];
  const double mb = p[6], rb = p[7], re = p[8], m_el = p[9], m_hd = p[10];
  const double Ij[7] = {p[11], p[11], p[12], p[12], p[13], p[13], p[13]};
  const double *q = x, *qd = x + 14, *pb = x + 11, *om = x + 21, *vb = x + 24;
  const double qw = x[7], qx = x[8], qy = x[9], qz = x[10];
  double hand[3], elbow[3], axes[7][3], orgs[7][3], J[7][3], JEz[3], r[3], t3[3];
  arm27_kinematics(q, p, hand, elbow, axes, orgs);
  for (int i = 0; i < 7; ++i) { for (int k = 0; k < 3; ++k) r[k] = hand[k] - orgs[i][k]; cross3(axes[i], r, J[i]); }
  for (int i = 0; i < 3; ++i) { for (int k = 0; k < 3; ++k) r[k] = elbow[k] - orgs[i][k]; cross3(axes[i], r, t3); JEz[i] = t3[2]; }
  double vh[3], d[3], nr[3], wxn[3], rel[3], vt[3], Fc[3], nxv[3], tc[3];
  for (int k = 0; k < 3; ++k)
    vh[k] = ((J[0][k] * qd[0] + J[1][k] * qd[1]) + (J[2][k] * qd[2] + J[3][k] * qd[3])) + ((J[4][k] * qd[4] + J[5][k] * qd[5]) + J[6][k] * qd[6]);
  for (int k = 0; k < 3; ++k) d[k] = pb[k] - hand[k];
  const double dist = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  const double idist = 1.0 / dist;
  for (int k = 0; k < 3; ++k) nr[k] = d[k] * idist;
  const double phi = dist - (rb + re);
  const double fn0 = (kc * sig) * softplus(-phi / sig);
  cross3(om, nr, wxn);
  for (int k = 0; k < 3; ++k) rel[k] = vb[k] - rb * wxn[k] - vh[k];
  const double vn = rel[0] * nr[0] + rel[1] * nr[1] + rel[2] * nr[2];
  for (int k = 0; k < 3; ++k) vt[k] = rel[k] - vn * nr[k];
  const double fnn = fn0 * (1.0 - dn * vn);
  for (int k = 0; k < 3; ++k) Fc[k] = fnn * nr[k] - (mu * fn0) * vt[k];
  cross3(nr, vt, nxv);
  for (int k = 0; k < 3; ++k) tc[k] = (rb * mu) * fn0 * nxv[k];
  const double fg0 = (kc * sig) * softplus(-(pb[2] - rb) / sig);
  const double vcx = vb[0] - rb * om[1], vcy = vb[1] + rb * om[0];
  const double Fg[3] = {-(mu * fg0) * vcx, -(mu * fg0) * vcy, fg0 * (1.0 - dn * vb[2])};
  const double tg[3] = {rb * Fg[1], -(rb * Fg[0]), 0.0};
  for (int i = 0; i < 7; ++i) {
    const double grav = g * (m_hd * J[i][2] + (i < 3 ? m_el * JEz[i] : 0.0));
    const double jf = J[i][0] * Fc[0] + J[i][1] * Fc[1] + J[i][2] * Fc[2];
    const double acc = (u[i] - bj * qd[i] - grav - jf) / Ij[i];
    const double qdn = qd[i] + dt * acc;
    xn[14 + i] = qdn; xn[i] = q[i] + dt * qdn;
  }
  const double ib = 1.0 / (0.4 * mb * rb * rb);
  double omn[3], vbn[3];
  for (int k = 0; k < 3; ++k) omn[k] = om[k] + dt * ((tc[k] + tg[k]) * ib);
  const double al[3] = {(Fc[0] + Fg[0]) / mb, (Fc[1] + Fg[1]) / mb, (Fc[2] + Fg[2]) / mb - g};
  for (int k = 0; k < 3; ++k) { vbn[k] = vb[k] + dt * al[k]; xn[11 + k] = pb[k] + dt * vbn[k]; xn[21 + k] = omn[k]; xn[24 + k] = vbn[k]; }
  const double hd = 0.5 * dt;
  xn[7] = qw + hd * (-(omn[0] * qx) - omn[1] * qy - omn[2] * qz);
  xn[8] = qx + hd * (qw * omn[0] + (omn[1] * qz - omn[2] * qy));
  xn[9] = qy + hd * (qw * omn[1] + (omn[2] * qx - omn[0] * qz));
  xn[10] = qz + hd * (qw * omn[2] + (omn[0] * qy - omn[1] * qx));
}

/* ---- arm + ball with coupled rigid-body joint dynamics (model 8): oracle/models_np.py:arm27c_step, same formulas, same order ---- */
static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void rot_acc(const double* al, const double* w, const double* r, double* out) {
  double a1[3], t[3], a2[3];
  cross3(al, r, a1); cross3(w, r, t); cross3(w, t, a2);
  for (int k = 0; k < 3; ++k) out[k] = a1[k] + a2[k];
}
static void arm27c_step(const double* x, const double* u, const double* p, double dt, double* xn) {
  const double g = p[0], kc = p[1], sig = p[2], dn = p[3], mu = p[4], bj = p[5];
  const double mb = p[6], rb = p[7], re = p[8], m_el = p[9], m_hd = p[10], m_wr = p[15];
  const double Ia[7] = {p[11], p[11], p[12], p[12], p[13], p[13], p[13]};
  const double *q = x, *qd = x + 14, *pb = x + 11, *om = x + 21, *vb = x + 24;
  const double qw = x[7], qx = x[8], qy = x[9], qz = x[10];
  double hand[3], wrist[3], elbow[3], axes[7][3], orgs[7][3], J[7][3], JW[5][3], JE[3][3], r[3];
  {                                                                   /* arm27c_kinematics */
    double ex[3] = {1.0, 0.0, 0.0}, ey[3] = {0.0, 1.0, 0.0}, ez[3] = {0.0, 0.0, 1.0}, pos[3] = {0.0, 0.0, A27_H0};
    for (int i = 0; i < 7; ++i) {
      const double s = sin(q[i]), c = cos(q[i]);
      if (i % 2 == 0) {
        for (int k = 0; k < 3; ++k) { axes[i][k] = ez[k]; orgs[i][k] = pos[k]; }
        for (int k = 0; k < 3; ++k) { const double a = c * ex[k] + s * ey[k], b = c * ey[k] - s * ex[k]; ex[k] = a; ey[k] = b; }
      } else {
        for (int k = 0; k < 3; ++k) { axes[i][k] = ey[k]; orgs[i][k] = pos[k]; }
        for (int k = 0; k < 3; ++k) { const double a = c * ex[k] - s * ez[k], b = c * ez[k] + s * ex[k]; ex[k] = a; ez[k] = b; }
      }
      if (i == 2) { for (int k = 0; k < 3; ++k) { pos[k] = pos[k] + A27_L1 * ez[k]; elbow[k] = pos[k]; } }
      else if (i == 4) { for (int k = 0; k < 3; ++k) { pos[k] = pos[k] + A27_L2 * ez[k]; wrist[k] = pos[k]; } }
    }
    for (int k = 0; k < 3; ++k) hand[k] = pos[k] + (p[14] * ex[k] + A27_L3 * ez[k]);
  }
  for (int i = 0; i < 7; ++i) { for (int k = 0; k < 3; ++k) r[k] = hand[k] - orgs[i][k]; cross3(axes[i], r, J[i]); }
  for (int i = 0; i < 5; ++i) { for (int k = 0; k < 3; ++k) r[k] = wrist[k] - orgs[i][k]; cross3(axes[i], r, JW[i]); }
  for (int i = 0; i < 3; ++i) { for (int k = 0; k < 3; ++k) r[k] = elbow[k] - orgs[i][k]; cross3(axes[i], r, JE[i]); }
  double vh[3], d[3], nr[3], wxn[3], rel[3], vt[3], Fc[3], nxv[3], tc[3];
  for (int k = 0; k < 3; ++k)
    vh[k] = ((J[0][k] * qd[0] + J[1][k] * qd[1]) + (J[2][k] * qd[2] + J[3][k] * qd[3])) + ((J[4][k] * qd[4] + J[5][k] * qd[5]) + J[6][k] * qd[6]);
  for (int k = 0; k < 3; ++k) d[k] = pb[k] - hand[k];
  const double dist = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  const double idist = 1.0 / dist;
  for (int k = 0; k < 3; ++k) nr[k] = d[k] * idist;
  const double phi = dist - (rb + re);
  const double fn0 = (kc * sig) * softplus(-phi / sig);
  cross3(om, nr, wxn);
  for (int k = 0; k < 3; ++k) rel[k] = vb[k] - rb * wxn[k] - vh[k];
  const double vn = rel[0] * nr[0] + rel[1] * nr[1] + rel[2] * nr[2];
  for (int k = 0; k < 3; ++k) vt[k] = rel[k] - vn * nr[k];
  const double fnn = fn0 * (1.0 - dn * vn);
  for (int k = 0; k < 3; ++k) Fc[k] = fnn * nr[k] - (mu * fn0) * vt[k];
  cross3(nr, vt, nxv);
  for (int k = 0; k < 3; ++k) tc[k] = (rb * mu) * fn0 * nxv[k];
  const double fg0 = (kc * sig) * softplus(-(pb[2] - rb) / sig);
  const double vcx = vb[0] - rb * om[1], vcy = vb[1] + rb * om[0];
  const double Fg[3] = {-(mu * fg0) * vcx, -(mu * fg0) * vcy, fg0 * (1.0 - dn * vb[2])};
  const double tg[3] = {rb * Fg[1], -(rb * Fg[0]), 0.0};
  /* velocity-product accelerations of the point masses */
  double w[3] = {0.0, 0.0, 0.0}, al[3] = {0.0, 0.0, 0.0}, wl[7][3], all_[7][3], wxa[3];
  for (int i = 0; i < 7; ++i) {
    cross3(w, axes[i], wxa);
    for (int k = 0; k < 3; ++k) al[k] = al[k] + wxa[k] * qd[i];
    for (int k = 0; k < 3; ++k) w[k] = w[k] + axes[i][k] * qd[i];
    for (int k = 0; k < 3; ++k) { wl[i][k] = w[k]; all_[i][k] = al[k]; }
  }
  double r2[3], r4[3], r6[3], aE[3], aW[3], aH[3], t4[3], t6[3];
  for (int k = 0; k < 3; ++k) { r2[k] = elbow[k] - orgs[2][k]; r4[k] = wrist[k] - elbow[k]; r6[k] = hand[k] - wrist[k]; }
  rot_acc(all_[2], wl[2], r2, aE);
  rot_acc(all_[4], wl[4], r4, t4);
  for (int k = 0; k < 3; ++k) aW[k] = aE[k] + t4[k];
  rot_acc(all_[6], wl[6], r6, t6);
  for (int k = 0; k < 3; ++k) aH[k] = aW[k] + t6[k];
  const double gE[3] = {aE[0], aE[1], aE[2] + g}, gW[3] = {aW[0], aW[1], aW[2] + g}, gH[3] = {aH[0], aH[1], aH[2] + g};
  double rhs[7], M[7][7], L[7][7], dd[7], idd[7], y[7], acc[7];
  for (int i = 0; i < 7; ++i) {
    double h = m_hd * dot3(J[i], gH);
    if (i < 5) h = h + m_wr * dot3(JW[i], gW);
    if (i < 3) h = h + m_el * dot3(JE[i], gE);
    rhs[i] = u[i] - bj * qd[i] - h - dot3(J[i], Fc);
  }
  for (int i = 0; i < 7; ++i)
    for (int j = 0; j <= i; ++j) {
      double v = m_hd * dot3(J[i], J[j]);
      if (i < 5) v = v + m_wr * dot3(JW[i], JW[j]);
      if (i < 3) v = v + m_el * dot3(JE[i], JE[j]);
      if (i == j) v = v + Ia[i];
      M[i][j] = v;
    }
  for (int j = 0; j < 7; ++j) {
    double v = M[j][j];
    for (int k = 0; k < j; ++k) v = v - (L[j][k] * L[j][k]) * dd[k];
    dd[j] = v; idd[j] = 1.0 / v;
    for (int i = j + 1; i < 7; ++i) {
      v = M[i][j];
      for (int k = 0; k < j; ++k) v = v - (L[i][k] * L[j][k]) * dd[k];
      L[i][j] = v * idd[j];
    }
  }
  for (int i = 0; i < 7; ++i) { double v = rhs[i]; for (int k = 0; k < i; ++k) v = v - L[i][k] * y[k]; y[i] = v; }
  for (int i = 6; i >= 0; --i) { double v = y[i] * idd[i]; for (int k = i + 1; k < 7; ++k) v = v - L[k][i] * acc[k]; acc[i] = v; }
  for (int i = 0; i < 7; ++i) { const double qdn = qd[i] + dt * acc[i]; xn[14 + i] = qdn; xn[i] = q[i] + dt * qdn; }
  const double ib = 1.0 / (0.4 * mb * rb * rb);
  double omn[3], vbn[3];
  for (int k = 0; k < 3; ++k) omn[k] = om[k] + dt * ((tc[k] + tg[k]) * ib);
  const double alb[3] = {(Fc[0] + Fg[0]) / mb, (Fc[1] + Fg[1]) / mb, (Fc[2] + Fg[2]) / mb - g};
  for (int k = 0; k < 3; ++k) { vbn[k] = vb[k] + dt * alb[k]; xn[11 + k] = pb[k] + dt * vbn[k]; xn[21 + k] = omn[k]; xn[24 + k] = vbn[k]; }
  const double hd = 0.5 * dt;
  xn[7] = qw + hd * (-(omn[0] * qx) - omn[1] * qy - omn[2] * qz);
  xn[8] = qx + hd * (qw * omn[0] + (omn[1] * qz - omn[2] * qy));
  xn[9] = qy + hd * (qw * omn[1] + (omn[2] * qx - omn[0] * qz));
  xn[10] = qz + hd * (qw * omn[2] + (omn[0] * qy - omn[1] * qx));
}

/* A model may declare a step infeasible (Drake's discrete update throwing, caught at ilqr.py:315-323). */
static int step_infeasible(const oracle_cfg* c, const double* xn) {
  if (c->model_id == 5) { for (int i = 18; i < 36; ++i) if (!(fabs(xn[i]) <= c->params[8])) return 1; }
  if (c->model_id == 6) { for (int i = 19; i < 37; ++i) if (!(fabs(xn[i]) <= c->params[6])) return 1; }
  return 0;
}

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
    case 5: { /* planar quadruped */
      double qdd[18];
      quad_accel(x, u, p, qdd);
      for (int i = 0; i < 18; ++i) { const double vn = x[18 + i] + dt * qdd[i]; xn[18 + i] = vn; xn[i] = x[i] + dt * vn; }
      break;
    }
    case 6: quad3d_step(x, u, p, dt, xn); break;    /* 3-D quadruped */
    case 7: arm27_step(x, u, p, dt, xn); break;     /* 7-joint arm + free ball */
    case 8: arm27c_step(x, u, p, dt, xn); break;    /* the same with coupled rigid-body joint dynamics */
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
  int* kp;          /* key-points of the last linearization (N entries) */
  int nk;
  unsigned char* done;   /* iterativeError: derivative evaluated at this index */
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
    if (step_infeasible(c, xn)) {                                     /* :317-323: L = inf, stop simulating ... */
      double qT = 0.0;                                                /* ... the terminal term is still added, at x[:,-1] = 0 (:327) */
      for (int i = 0; i < n; ++i) { double s = 0.0; for (int j = 0; j < n; ++j) s += Qf[i * n + j] * (0.0 - xnom[j]); qT += (0.0 - xnom[i]) * s; }
      *expected = ex;
      return INFINITY + qT;
    }
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

/* _calc_dynamics_partials at step t of the trial (central differences stand in for AutoDiff): ilqr.py:233-272 */
static void partials_at(const oracle_cfg* c, const work* w, int t) {
  const int n = c->n, m = c->m, N = c->N;
  const double h = c->fd_h, inv2h = 1.0 / (2.0 * h);
  double xt[MAXN], ut[MAXM], xp[MAXN], up[MAXM], fp[MAXN], fm[MAXN];
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

/* get_keypoints_set_interval: ilqr.py:417-432 (the LAST entry is overwritten, not appended) */
static int keypoints_set_interval(const oracle_cfg* c, int* kp) {
  const int N = c->N, minN = c->minN;
  const int nk = (N - 2) / minN + 1;
  for (int i = 0; i < nk; ++i) kp[i] = i * minN;
  if (kp[nk - 1] != N - 2) kp[nk - 1] = N - 2;
  return nk;
}

/* get_keypoints_adaptive_jerk + calc_jerk_profile: ilqr.py:434-486 (signed second difference of the velocity rows,
 * dof = int(n/2); the last entry is overwritten with N-2) */
static int keypoints_adaptive_jerk(const oracle_cfg* c, const work* w, int* kp) {
  const int n = c->n, N = c->N, dof = n / 2;
  int nk = 0, counter = 0;
  kp[nk++] = 0;
  for (int t = 0; t < N - 3; ++t) {
    counter += 1;
    if (counter >= c->minN) {
      for (int i = 0; i < dof; ++i) {
        const double a1 = X(w->x, i + dof, t + 2) - X(w->x, i + dof, t + 1);
        const double a2 = X(w->x, i + dof, t + 1) - X(w->x, i + dof, t);
        if (a1 - a2 > c->jerk_thr) { kp[nk++] = t; counter = 0; break; }
      }
    }
    if (counter >= c->maxN) { kp[nk++] = t; counter = 0; }
  }
  if (kp[nk - 1] != N - 2) kp[nk - 1] = N - 2;
  return nk;
}

/* get_keypoints_iterative_error + check_one_matrix_error: ilqr.py:488-593.  Level-synchronous bisection; a bin is
 * split when sum_ij ((fx[s]+fx[e])/2 - fx[mid])^2 / (2n) > threshold; bins with e - s <= minN pass untested.  The
 * Jacobians it evaluates are written into fx/fu as it goes, like the reference. */
static int keypoints_iterative_error(const oracle_cfg* c, const work* w, int* kp) {
  const int n = c->n, N = c->N;
  memset(w->done, 0, (size_t)N);
  int* bins = (int*)malloc(sizeof(int) * 4 * (size_t)N);
  int* next = (int*)malloc(sizeof(int) * 4 * (size_t)N);
  int nb = 1;
  bins[0] = 0; bins[1] = N - 2;
  while (nb > 0) {
    int nn = 0;
    for (int q = 0; q < nb; ++q) {
      const int s = bins[2 * q], e = bins[2 * q + 1];
      if (e - s <= c->minN) continue;
      const int mid = (s + e) / 2;
      if (!w->done[s]) { partials_at(c, w, s); w->done[s] = 1; }
      if (!w->done[mid]) { partials_at(c, w, mid); w->done[mid] = 1; }
      if (!w->done[e]) { partials_at(c, w, e); w->done[e] = 1; }
      double sum = 0.0;
      for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
        const double lin = (FX(w->fx, i, j, e) + FX(w->fx, i, j, s)) / 2;
        const double d = lin - FX(w->fx, i, j, mid);
        sum += d * d;
      }
      if (sum / (2 * n) > c->err_thr) { next[2 * nn] = s; next[2 * nn + 1] = mid; nn++; next[2 * nn] = mid; next[2 * nn + 1] = e; nn++; }
    }
    int* tmp = bins; bins = next; next = tmp;
    nb = nn;
  }
  free(bins); free(next);
  int nk = 0;
  for (int t = 0; t < N - 1; ++t) if (w->done[t]) kp[nk++] = t;
  return nk;
}

/* _get_derivatives: key-points, partials at them, interpolation in between - ilqr.py:380-415, 596-621 */
static void linearize(const oracle_cfg* c, work* w) {
  const int n = c->n, m = c->m, N = c->N;
  int* kp = w->kp;
  int nk;
  if (c->kp_method == 1) nk = keypoints_adaptive_jerk(c, w, kp);
  else if (c->kp_method == 2) nk = keypoints_iterative_error(c, w, kp);
  else nk = keypoints_set_interval(c, kp);
  w->nk = nk;
  if (c->kp_method != 2) for (int q = 0; q < nk; ++q) partials_at(c, w, kp[q]);      /* :409-411 */
  if (!(c->kp_method == 0 && c->minN == 1)) {                        /* :414 */
    for (int q = 0; q + 1 < nk; ++q) {
      const int s = kp[q], e = kp[q + 1];
      for (int j = s + 1; j < e; ++j) {
        for (int r = 0; r < n * n; ++r) { const double fs = w->fx[r * (N - 1) + s], fe = w->fx[r * (N - 1) + e]; w->fx[r * (N - 1) + j] = fs + (fe - fs) * (j - s) / (e - s); }
        for (int r = 0; r < n * m; ++r) { const double fs = w->fu[r * (N - 1) + s], fe = w->fu[r * (N - 1) + e]; w->fu[r * (N - 1) + j] = fs + (fe - fs) * (j - s) / (e - s); }
      }
    }
  }
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
/* hist (optional): hist_cap rows of (cost, eps, line-search trials, key-point count) per iteration, like the rows of
 * the reference's console table (ilqr.py:704) */
static int solve_one_h(const oracle_cfg* c, const double* Q, const double* R, const double* Qf, const double* xnom,
                       const double* x0, work* w, double* cost, int* iters, int* ls_trials, double* hist, int hist_cap) {
  const int n = c->n, m = c->m, N = c->N;
  double L = INFINITY, improvement = INFINITY;
  int it = 0, ls = 0, status = 0;
  while (improvement > c->delta) {
    if (it >= c->max_iters) { status = 1; break; }
    double eps = 1.0, Lnew = 0.0, ex;
    int accepted = 0, ls_it = 0;
    while (eps >= 1e-8) {                                             /* :302 */
      ls++; ls_it++;
      Lnew = rollout(c, Q, R, Qf, xnom, x0, w, eps, &ex);
      if (L - Lnew > c->gamma * ex) { accepted = 1; break; }          /* :330-331 */
      eps *= c->beta;                                                 /* :335 */
    }
    if (!accepted) { status = 2; break; }
    linearize(c, w);                                                  /* :370 */
    if (hist && it < hist_cap) { hist[4 * it] = Lnew; hist[4 * it + 1] = eps; hist[4 * it + 2] = (double)ls_it; hist[4 * it + 3] = (double)w->nk; }
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
static int solve_one(const oracle_cfg* c, const double* Q, const double* R, const double* Qf, const double* xnom,
                     const double* x0, work* w, double* cost, int* iters, int* ls_trials) {
  return solve_one_h(c, Q, R, Qf, xnom, x0, w, cost, iters, ls_trials, NULL, 0);
}

static void work_alloc(work* w, int n, int m, int N) {
  w->x_bar = (double*)malloc(sizeof(double) * n * N); w->x = (double*)malloc(sizeof(double) * n * N);
  w->u_bar = (double*)malloc(sizeof(double) * m * (N - 1)); w->u = (double*)malloc(sizeof(double) * m * (N - 1));
  w->K = (double*)malloc(sizeof(double) * m * n * (N - 1)); w->kappa = (double*)malloc(sizeof(double) * m * (N - 1));
  w->dV = (double*)malloc(sizeof(double) * (N - 1));
  w->fx = (double*)malloc(sizeof(double) * n * n * (N - 1)); w->fu = (double*)malloc(sizeof(double) * n * m * (N - 1));
  w->kp = (int*)calloc((size_t)N + 1, sizeof(int)); w->nk = 0;
  w->done = (unsigned char*)calloc((size_t)N + 1, 1);
}
static void work_free(work* w) { free(w->x_bar); free(w->x); free(w->u_bar); free(w->u); free(w->K); free(w->kappa); free(w->dV); free(w->fx); free(w->fu); free(w->kp); free(w->done); }

/* Batched cold-start solve; outputs may be NULL.  Returns the number of threads used. */
int oracle_solve_batch_ex(const oracle_cfg* c, int B, const double* Q, const double* R, const double* Qf, const double* xnom,
                          const double* x0, const double* u_guess, double* x_bar, double* u_bar, double* K, double* kappa,
                          double* cost, int* iters, int* ls_trials, int* status, int nthreads,
                          double* hist, int hist_cap, int* kp_count, int* kp_list);
int oracle_solve_batch(const oracle_cfg* c, int B, const double* Q, const double* R, const double* Qf, const double* xnom,
                       const double* x0, const double* u_guess, double* x_bar, double* u_bar, double* K, double* kappa,
                       double* cost, int* iters, int* ls_trials, int* status, int nthreads) {
  return oracle_solve_batch_ex(c, B, Q, R, Qf, xnom, x0, u_guess, x_bar, u_bar, K, kappa, cost, iters, ls_trials, status, nthreads,
                               NULL, 0, NULL, NULL);
}

/* ... with the per-iteration history (B, hist_cap, 4) and the key-points of the LAST linearization (counts (B,), lists
 * (B, N-1)); any of them may be NULL. */
int oracle_solve_batch_ex(const oracle_cfg* c, int B, const double* Q, const double* R, const double* Qf, const double* xnom,
                          const double* x0, const double* u_guess, double* x_bar, double* u_bar, double* K, double* kappa,
                          double* cost, int* iters, int* ls_trials, int* status, int nthreads,
                          double* hist, int hist_cap, int* kp_count, int* kp_list) {
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
      const int st = solve_one_h(c, Q, R, Qf, xnom, x0 + (size_t)b * n, &w, &L, &it, &ls,
                                 hist ? hist + (size_t)b * hist_cap * 4 : NULL, hist_cap);
      if (kp_count) kp_count[b] = w.nk;
      if (kp_list) memcpy(kp_list + (size_t)b * (N - 1), w.kp, sizeof(int) * (size_t)(w.nk < N - 1 ? w.nk : N - 1));
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
