// Build-owned discrete-time dynamics x+ = f(x,u), device side.
//
// The reference evaluates f through Drake (CalcForcedDiscreteVariableUpdate,
// /root/reference/ilqr.py:223-229) — external, absent here.  These
// closed-form models (semi-implicit Euler) are the build's own definitions; the
// identical formulas, in the same operation order, live in oracle/models_np.py
// (NumPy) and oracle/ilqr_oracle.c (C).  Ids/parameter layout: include/mi_ilqr.h.
#pragma once
#include "dual.hpp"

namespace mi {

struct Pendulum {            // params [ml2, b, mgl]
  static constexpr int n = 2, m = 1, n_params = 3;
  static constexpr bool kNewtonRollout = true;   // step<Dual2> is available: time-parallel Newton rollout (ilqr_small.hpp)
  template <class T>
  __device__ static inline void step(const T* x, const T* u, T* xn, const double* p, double dt) {
    const double ml2 = p[0], b = p[1], mgl = p[2];
    const T th = x[0], w = x[1];
    const double inv_ml2 = 1.0 / ml2;      // loop-invariant: hoisted out of the time loops
    const T acc = (u[0] - b * w - mgl * mi_sin(th)) * inv_ml2;
    const T wn = w + dt * acc;
    xn[0] = th + dt * wn;
    xn[1] = wn;
  }
};

struct Acrobot {             // params [m1,m2,l1,lc1,lc2,Ic1,Ic2,b1,b2,g]
  static constexpr int n = 4, m = 1, n_params = 10;
  template <class T>
  __device__ static inline void step(const T* x, const T* u, T* xn, const double* p, double dt) {
    const double m1 = p[0], m2 = p[1], l1 = p[2], lc1 = p[3], lc2 = p[4];
    const double Ic1 = p[5], Ic2 = p[6], b1 = p[7], b2 = p[8], g = p[9];
    const T q1 = x[0], q2 = x[1], v1 = x[2], v2 = x[3];
    const double I1 = Ic1 + m1 * lc1 * lc1;
    const double I2 = Ic2 + m2 * lc2 * lc2;
    const T s1 = mi_sin(q1), s2 = mi_sin(q2), c2 = mi_cos(q2), s12 = mi_sin(q1 + q2);
    const double h = m2 * l1 * lc2;
    const T M11 = I1 + I2 + m2 * l1 * l1 + 2.0 * h * c2;
    const T M12 = I2 + h * c2;
    const double M22 = I2;
    const T cb1 = -2.0 * h * s2 * v2 * v1 - h * s2 * v2 * v2;
    const T cb2 = h * s2 * v1 * v1;
    const T g1 = g * (m1 * lc1 + m2 * l1) * s1 + g * m2 * lc2 * s12;
    const T g2 = g * m2 * lc2 * s12;
    const T r1 = -cb1 - g1 - b1 * v1;
    const T r2 = u[0] - cb2 - g2 - b2 * v2;
    const T idet = mi_rcp(M11 * M22 - M12 * M12);
    const T a1 = (M22 * r1 - M12 * r2) * idet;
    const T a2 = (M11 * r2 - M12 * r1) * idet;
    const T v1n = v1 + dt * a1, v2n = v2 + dt * a2;
    xn[0] = q1 + dt * v1n;
    xn[1] = q2 + dt * v2n;
    xn[2] = v1n;
    xn[3] = v2n;
  }
};

template <bool WALL>
struct CartPoleT {           // params [mc, mp, l, g, wall_face_x, ball_radius, k, sigma]
  static constexpr int n = 4, m = 1, n_params = WALL ? 8 : 4;
  // The contact model's exp / log1p constants can be handed in by a caller that keeps them in vector registers
  // across a long loop (fastmath.hpp: SoftplusPool) - the plain-double rollout of the wave-per-problem kernels.
  static constexpr bool kHasStepPool = WALL;
  using StepPool = SoftplusPool;
  __device__ static inline void step_pooled(const double* x, const double* u, double* xn, const double* p, double dt, const SoftplusPool& pool) {
    step_impl<double>(x, u, xn, p, dt, [&](double z) __attribute__((always_inline)) { return mi_softplus(z, pool); });
  }
  template <class T>
  __device__ static inline void step(const T* x, const T* u, T* xn, const double* p, double dt) {
    step_impl<T>(x, u, xn, p, dt, [](T z) __attribute__((always_inline)) { return mi_softplus(z); });
  }
  template <class T, class SP>
  __device__ static inline void step_impl(const T* x, const T* u, T* xn, const double* p, double dt, SP softplus) {
    const double mc = p[0], mp = p[1], l = p[2], g = p[3];
    const T px = x[0], th = x[1], vx = x[2], w = x[3];
    const T s = mi_sin(th), c = mi_cos(th);
    const double M11 = mc + mp;
    const T M12 = mp * l * c;
    const double M22 = mp * l * l;
    T r1 = u[0] + mp * l * w * w * s;
    T r2 = -mp * g * l * s;
    if (WALL) {
      const double face = p[4], rad = p[5], k = p[6], sig = p[7];
      const double inv_sig = 1.0 / sig;      // loop-invariant: hoisted (a full fp64 division is ~12 instructions)
      const T tip = px + l * s;
      const T phi = tip - rad - face;
      const T F = k * sig * softplus(-phi * inv_sig);
      r1 = r1 + F;
      r2 = r2 + F * l * c;
    }
    const T idet = mi_rcp(M11 * M22 - M12 * M12);
    const T a1 = (M22 * r1 - M12 * r2) * idet;
    const T a2 = (M11 * r2 - M12 * r1) * idet;
    const T vxn = vx + dt * a1, wn = w + dt * a2;
    xn[0] = px + dt * vxn;
    xn[1] = th + dt * wn;
    xn[2] = vxn;
    xn[3] = wn;
  }
};
using CartPole = CartPoleT<false>;
using CartPoleWall = CartPoleT<true>;

struct Synth36 {             // params [ks, c, kc, bu]; 18 coupled pendula, dofs 6..17 actuated
  static constexpr bool kPivSplit = true;     // (launch_large.hpp: two forms of the kernels with a backward pass)
  static constexpr int n = 36, m = 12, n_params = 4, nq = 18;
  // One dof of the chain: usable dof-parallel (one lane per dof) by the large-n kernels.  x and u
  // are anything indexable (pointers, or accessors that perturb / seed one entry on the fly).
  template <class T, class XA, class UA>
  __device__ static inline void dof(int i, const XA& x, const UA& u, T& qn, T& vn, const double* p, double dt) {
    const double ks = p[0], c = p[1], kc = p[2], bu = p[3];
    const T qi = x[i], vi = x[nq + i];
    T a = -ks * mi_sin(qi) - c * vi;
    if (i < nq - 1) a = a + kc * mi_sin(x[i + 1] - qi);
    if (i > 0) a = a - kc * mi_sin(qi - x[i - 1]);
    if (i >= 6) a = a + u[i - 6];
    else a = a + bu * (u[2 * i] - u[2 * i + 1]);
    vn = vi + dt * a;
    qn = qi + dt * vn;
  }
  // Dynamics sparsity for the Jacobian code: the dofs whose update reads input column `col` of
  // [x | u] (q_j: j-1, j, j+1;  v_j: j;  u_k: 6+k and k/2).  Every other entry of that Jacobian
  // column is an exact zero (f(x+h e) and f(x-h e) coincide bitwise there).
  static constexpr int kMaxAffected = 3;
  __device__ static inline int affected(int col, int (&idx)[kMaxAffected]) {
    if (col < nq) {
      int c_ = 0;
      if (col > 0) idx[c_++] = col - 1;
      idx[c_++] = col;
      if (col < nq - 1) idx[c_++] = col + 1;
      return c_;
    }
    if (col < n) { idx[0] = col - nq; return 1; }
    const int k = col - n;
    idx[0] = k / 2; idx[1] = 6 + k;
    return 2;
  }
  template <class T>
  __device__ static inline void step(const T* x, const T* u, T* xn, const double* p, double dt) {
#pragma unroll
    for (int i = 0; i < nq; ++i) dof(i, x, u, xn[i], xn[nq + i], p, dt);
  }
};

// Planar floating-base articulated body of the mini-cheetah's SHAPE (mini_cheetah.py:41-52: 18 positions,
// 18 velocities, 12 actuators): trunk (x, z, pitch) + four 3-link legs (hip, knee, ankle: the actuated joints) +
// a passive 3-link tail; compliant ground contact at the four feet and the tail tip.  Accelerations by the
// articulated-body algorithm in world-aligned planar coordinates (a body's spatial quantities: moment about /
// rotation at its joint axis, x, z; transforms between bodies are pure translations) - the same formulas as
// oracle/models_np.py:quad_accel and oracle/ilqr_oracle.c.  params [g, k, sigma, dn, mu, b_leg, b_tail, k_tail, v_max].
//   q = [x, z, pitch | leg0 hip,knee,ankle | leg1 | leg2 | leg3 | tail 1,2,3],  u -> the 12 leg joints.
// The tree is five 3-body CHAINS hanging off the trunk, which is how the work is cut: a chain's leaves-to-root
// pass yields its articulated inertia and bias force at the trunk (chain_up), the trunk's 3x3 system gives the
// base acceleration (base_solve), a chain's root-to-leaves pass its joint accelerations (chain_down).  One
// lane per chain in the rollout (cooperative step of the workgroup-per-problem kernel), all five chains on
// one thread in the linearization's finite differences.
// The model can FAIL: a step that leaves |v| <= v_max is declared infeasible - the device analogue of Drake's
// discrete update throwing, which the reference's line search turns into L = inf (ilqr.py:315-323).
struct PlanarQuad {
  static constexpr int n = 36, m = 12, n_params = 9, nq = 18, kChains = 5;
  static constexpr bool kChainCooperative = true;
  static constexpr bool kPivSplit = true;     // (launch_large.hpp: two forms of the kernels with a backward pass; C5q + 2.5 %)
  static constexpr int kEarlyLeaderBlocks = 1;   // (ilqr_large.hpp: early linearization - the last block of the horizon is the leader's)
  static constexpr bool kCanFail = true;
  template <class T> struct Trunk { T sn, cs, om, pz, vx, vz; };
  template <class T> struct Agg { T J, hx, hz, mxx, mxz, mzz, bn, bx, bz; };
  template <class T> struct Saved { T D[3], Ux[3], Uz[3], uu[3], dx[3], dz[3], cbx[3], cbz[3]; };
  __device__ static constexpr double len(int c, int b) { return c < 4 ? (b == 0 ? 0.20 : (b == 1 ? 0.18 : 0.14)) : 0.12; }
  __device__ static constexpr double mass(int c, int b) { return c < 4 ? (b == 0 ? 0.60 : (b == 1 ? 0.40 : 0.30)) : 0.10; }
  __device__ static constexpr double atx(int c) { return c < 2 ? 0.19 : (c < 4 ? -0.19 : -0.25); }
  __device__ static constexpr double atz(int c) { return c < 4 ? 0.0 : 0.02; }
  __device__ static constexpr double tail_rest(int b) { return b == 0 ? -1.2 : -0.2; }
  static constexpr double kTrunkMass = 4.0, kTrunkInertia = 0.06;
  // which chain an input column of [x | u] belongs to (-1: the trunk's own coordinates - every chain reads them)
  __device__ static constexpr int chain_of_input(int col) {
    return col < 3 ? -1 : (col < nq ? (col - 3) / 3 : (col < nq + 3 ? -1 : (col < n ? (col - nq - 3) / 3 : (col - n) / 3)));
  }

  // the n + m input columns ordered by owner: legs 0..3 (3 angles, 3 velocities, 3 torques each), the tail
  // (3 angles, 3 velocities), the trunk (x, z, pitch and their velocities)
  __device__ static constexpr int input_by_owner(int rank) {
    if (rank < 36) { const int c = rank / 9, r = rank - 9 * c; return r < 3 ? 3 + 3 * c + r : (r < 6 ? nq + 3 + 3 * c + (r - 3) : n + 3 * c + (r - 6)); }
    if (rank < 42) { const int r = rank - 36; return r < 3 ? 15 + r : nq + 15 + (r - 3); }
    const int r = rank - 42;
    return r < 3 ? r : nq + (r - 3);
  }
  template <class T, class XA>
  __device__ static inline void trunk_state(const XA& x, Trunk<T>& tr) {
    const T th = x[2];
    tr.sn = mi_sin(th); tr.cs = mi_cos(th); tr.om = x[nq + 2]; tr.pz = x[1]; tr.vx = x[nq + 0]; tr.vz = x[nq + 1];
  }
  // trunk's own inertia about its origin (= its COM) and bias (gravity only: no centripetal term at the COM)
  template <class T>
  __device__ static inline void trunk_agg(const double* p, Agg<T>& a) {
    a.J = kTrunkInertia; a.hx = 0.0; a.hz = 0.0; a.mxx = kTrunkMass; a.mxz = 0.0; a.mzz = kTrunkMass;
    a.bn = 0.0; a.bx = 0.0; a.bz = kTrunkMass * p[0];
  }
  template <class T>
  __device__ static inline void agg_add(Agg<T>& a, const Agg<T>& b) {
    a.J = a.J + b.J; a.hx = a.hx + b.hx; a.hz = a.hz + b.hz; a.mxx = a.mxx + b.mxx; a.mxz = a.mxz + b.mxz; a.mzz = a.mzz + b.mzz;
    a.bn = a.bn + b.bn; a.bx = a.bx + b.bx; a.bz = a.bz + b.bz;
  }
  // Chain c (0..3 legs, 4 tail): kinematics, joint torques, contact at the tip, leaves-to-root pass.
  // `agg`: the chain's contribution to the trunk's articulated inertia / bias; `sv`: what chain_down needs.
  template <class T, class XA, class UA>
  __device__ static inline void chain_up(int c, const Trunk<T>& tr, const XA& x, const UA& u, const double* p, Agg<T>& agg, Saved<T>& sv) {
    const double g = p[0], kc = p[1], sig = p[2], dn = p[3], mu = p[4], b_leg = p[5], b_tail = p[6], k_tail = p[7];
    const double inv_sig = 1.0 / sig;
    const int j0 = 3 + 3 * c;                      // first joint dof of the chain
    T sn[3], cs[3], om[3], rx[3], rz[3], tau[3];
    T J[3], hx[3], hz[3], mxx[3], mxz[3], mzz[3], bn[3], bx[3], bz[3];
    // ---- pass 1 (root to leaves)
    T th_par = x[2], om_par = tr.om, sn_par = tr.sn, cs_par = tr.cs;
    T pz = tr.pz, vx = tr.vx, vz = tr.vz;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      if (b == 0) { sv.dx[0] = cs_par * atx(c) - sn_par * atz(c); sv.dz[0] = sn_par * atx(c) + cs_par * atz(c); }
      else { const double lp = len(c, b - 1); sv.dx[b] = lp * sn_par; sv.dz[b] = -lp * cs_par; }
      const T qj = x[j0 + b], vj = x[nq + j0 + b];
      const T th = th_par + qj;
      om[b] = om_par + vj;
      sn[b] = mi_sin(th); cs[b] = mi_cos(th);
      pz = pz + sv.dz[b];
      vx = vx - om_par * sv.dz[b]; vz = vz + om_par * sv.dx[b];
      const double hl = 0.5 * len(c, b);
      rx[b] = hl * sn[b]; rz[b] = -hl * cs[b];
      const T w2p = om_par * om_par;
      sv.cbx[b] = -w2p * sv.dx[b]; sv.cbz[b] = -w2p * sv.dz[b];
      if (c < 4) tau[b] = u[3 * c + b] - b_leg * vj;
      else tau[b] = -b_tail * vj - k_tail * (qj - tail_rest(b));
      const double m_ = mass(c, b), ic = m_ * len(c, b) * len(c, b) / 12;
      const T w2 = om[b] * om[b];
      J[b] = ic + m_ * (rx[b] * rx[b] + rz[b] * rz[b]);
      hx[b] = -m_ * rz[b]; hz[b] = m_ * rx[b];
      mxx[b] = m_; mxz[b] = 0.0; mzz[b] = m_;
      bn[b] = m_ * g * rx[b];
      bx[b] = -m_ * w2 * rx[b];
      bz[b] = -m_ * w2 * rz[b] + m_ * g;
      if (b == 2) {                                  // ground contact at the tip of the last link
        const T ex = 2.0 * rx[2], ez = 2.0 * rz[2];
        const T tz = pz + ez;
        const T tvx = vx - om[2] * ez, tvz = vz + om[2] * ex;
        const T fn0 = (kc * sig) * mi_softplus(-tz * inv_sig);
        const T fn = fn0 * (1.0 - dn * tvz);
        const T ft = -mu * fn0 * tvx;
        bn[2] = bn[2] - (ex * fn - ez * ft);
        bx[2] = bx[2] - ft;
        bz[2] = bz[2] - fn;
      }
      th_par = th; om_par = om[b]; sn_par = sn[b]; cs_par = cs[b];
    }
    // ---- pass 2 (leaves to root): eliminate joint b, shift what is left to the parent's axis
#pragma unroll
    for (int b = 2; b >= 0; --b) {
      sv.D[b] = J[b]; sv.Ux[b] = hx[b]; sv.Uz[b] = hz[b];
      sv.uu[b] = tau[b] - bn[b];
      const T invD = mi_rcp(sv.D[b]);
      const T exx = mxx[b] - sv.Ux[b] * sv.Ux[b] * invD;
      const T exz = mxz[b] - sv.Ux[b] * sv.Uz[b] * invD;
      const T ezz = mzz[b] - sv.Uz[b] * sv.Uz[b] * invD;
      const T s_ = sv.uu[b] * invD;
      const T fx = bx[b] + exx * sv.cbx[b] + exz * sv.cbz[b] + sv.Ux[b] * s_;
      const T fz = bz[b] + exz * sv.cbx[b] + ezz * sv.cbz[b] + sv.Uz[b] * s_;
      const T gx = -exx * sv.dz[b] + exz * sv.dx[b];
      const T gz = -exz * sv.dz[b] + ezz * sv.dx[b];
      const T dJ = -sv.dz[b] * gx + sv.dx[b] * gz;
      const T dn_ = tau[b] - sv.dz[b] * fx + sv.dx[b] * fz;
      if (b > 0) {
        J[b - 1] = J[b - 1] + dJ; hx[b - 1] = hx[b - 1] + gx; hz[b - 1] = hz[b - 1] + gz;
        mxx[b - 1] = mxx[b - 1] + exx; mxz[b - 1] = mxz[b - 1] + exz; mzz[b - 1] = mzz[b - 1] + ezz;
        bn[b - 1] = bn[b - 1] + dn_; bx[b - 1] = bx[b - 1] + fx; bz[b - 1] = bz[b - 1] + fz;
      } else {
        agg.J = dJ; agg.hx = gx; agg.hz = gz; agg.mxx = exx; agg.mxz = exz; agg.mzz = ezz;
        agg.bn = dn_; agg.bx = fx; agg.bz = fz;
      }
    }
  }
  // Floating base: I_A a = -p_A (3x3 symmetric, LDL^T in the order x, z, pitch).
  template <class T>
  __device__ static inline void base_solve(const Agg<T>& t, T& ax, T& az, T& alpha) {
    const T a11 = t.mxx, a12 = t.mxz, a13 = t.hx, a22 = t.mzz, a23 = t.hz, a33 = t.J;
    const T r1 = -t.bx, r2 = -t.bz, r3 = -t.bn;
    const T i11 = mi_rcp(a11);
    const T l21 = a12 * i11, l31 = a13 * i11;
    const T d2 = a22 - l21 * a12, e23 = a23 - l21 * a13;
    const T i22 = mi_rcp(d2);
    const T l32 = e23 * i22;
    const T d3 = a33 - l31 * a13 - l32 * e23;
    const T y2 = r2 - l21 * r1;
    const T y3 = r3 - l31 * r1 - l32 * y2;
    alpha = y3 * mi_rcp(d3);
    az = (y2 - e23 * alpha) * i22;
    ax = (r1 - a12 * az - a13 * alpha) * i11;
  }
  // Root-to-leaves pass of one chain: joint accelerations from the base acceleration.
  template <class T>
  __device__ static inline void chain_down(const Saved<T>& sv, T alpha, T ax, T az, T (&qdd)[3]) {
    T al = alpha, acx = ax, acz = az;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const T apx = acx - al * sv.dz[b] + sv.cbx[b];
      const T apz = acz + al * sv.dx[b] + sv.cbz[b];
      const T qi = (sv.uu[b] - (sv.D[b] * al + sv.Ux[b] * apx + sv.Uz[b] * apz)) * mi_rcp(sv.D[b]);
      qdd[b] = qi;
      al = al + qi; acx = apx; acz = apz;
    }
  }
  // All 18 generalized accelerations on one thread.  The chains' up-passes are run twice (once for the trunk's
  // aggregate, once more right before their down-pass) instead of keeping 5 x 24 saved values alive.
  template <class T, class XA, class UA>
  __device__ static inline void accel(const XA& x, const UA& u, const double* p, T (&qdd)[nq]) {
    Trunk<T> tr;
    trunk_state<T>(x, tr);
    Agg<T> tot;
    trunk_agg<T>(p, tot);
    for (int c = 0; c < kChains; ++c) {
      Agg<T> a; Saved<T> sv;
      chain_up<T>(c, tr, x, u, p, a, sv);
      agg_add(tot, a);
    }
    T ax, az, alpha;
    base_solve<T>(tot, ax, az, alpha);
    qdd[0] = ax; qdd[1] = az; qdd[2] = alpha;
    for (int c = 0; c < kChains; ++c) {
      Agg<T> a; Saved<T> sv;
      chain_up<T>(c, tr, x, u, p, a, sv);
      T q3[3];
      chain_down<T>(sv, alpha, ax, az, q3);
      qdd[3 + 3 * c] = q3[0]; qdd[4 + 3 * c] = q3[1]; qdd[5 + 3 * c] = q3[2];
    }
  }
  // semi-implicit Euler: v+ = v + dt a(q, v, u); q+ = q + dt v+
  template <class T, class XA, class UA>
  __device__ static inline void step_acc(const XA& x, const UA& u, T* xn, const double* p, double dt) {
    T qdd[nq];
    accel<T>(x, u, p, qdd);
    for (int i = 0; i < nq; ++i) {
      const T vn = x[nq + i] + dt * qdd[i];
      xn[nq + i] = vn;
      xn[i] = x[i] + dt * vn;
    }
  }
  template <class T>
  __device__ static inline void step(const T* x, const T* u, T* xn, const double* p, double dt) { step_acc<T>(x, u, xn, p, dt); }
  __device__ static inline bool infeasible_velocity(double vn, const double* p) { return !(fabs(vn) <= p[8]); }
};

// 3-D floating-base quadruped with the state layout of mini_cheetah.py:41-52: 19 positions (unit quaternion w,x,y,z |
// base position | 4 x (ab/ad, hip, knee)) + 18 velocities (body-frame angular | world-frame linear | joint rates),
// 12 joint torques: n = 37, m = 12.  The same formulas, in the same operation order, as oracle/models_np.py:quad3d_leg /
// quad3d_step and oracle/ilqr_oracle.c.  params [g, k, sigma, dn, mu, b_joint, v_max, m_trunk, Ixx, Iyy, Izz, I_abad,
// I_hip, I_knee].  The trunk is a 3-D rigid body (Euler's equations; the attitude quaternion is integrated as
// q+ = q + dt/2 q (x) (0, w+) and never renormalized inside a step, like the reference's plain-vector iLQR); legs have
// 3-D kinematics and massless links, every joint carries its actuator's reflected inertia; a foot's contact force
// (smooth penalty, normal damping, viscous friction) reaches the trunk as a wrench and the joints through J^T f.
// The work is cut per LEG: leg() yields the foot's force (world frame), its moment about the trunk origin (body frame) and
// the leg's three joint accelerations; the four legs are summed as (0 + 2) + (1 + 3) - the order a DPP row sum of one
// lane per leg produces in the rollout, and the order step() uses, so both give the same bits.
// The model can FAIL like PlanarQuad: a velocity outside [-v_max, v_max] makes the step infeasible (ilqr.py:315-323).
struct Quad3D {
  static constexpr bool kPivSplit = true;     // (launch_large.hpp)
  static constexpr int n = 37, m = 12, n_params = 14, nq = 19, nv = 18, kLegs = 4;
  static constexpr bool kLegCooperative = true;
  static constexpr bool kCanFail = true;
  static constexpr double kL0 = 0.062, kL1 = 0.209, kL2 = 0.195, kHipX = 0.19, kHipY = 0.049;
  template <class T> struct LegOut { T fw[3], tq[3], ja[3]; };
  // which leg an input column of [x | u] belongs to (-1: the trunk's own coordinates - every leg reads them)
  __device__ static constexpr int leg_of_input(int col) {
    return col < 7 ? -1 : (col < nq ? (col - 7) / 3 : (col < 25 ? -1 : (col < n ? (col - 25) / 3 : (col - n) / 3)));
  }
  // the n + m input columns ordered by owner: legs 0..3 (3 angles, 3 rates, 3 torques each), then the trunk's 13
  __device__ static constexpr int input_by_owner(int rank) {
    if (rank < 36) { const int k = rank / 9, r = rank - 9 * k; return r < 3 ? 7 + 3 * k + r : (r < 6 ? 25 + 3 * k + (r - 3) : n + 3 * k + (r - 6)); }
    const int r = rank - 36;
    return r < 7 ? r : 19 + (r - 7);
  }

  template <class T, class XA>
  __device__ static inline void rotation(const XA& x, T (&R)[3][3]) {
    const T qw = x[0], qx = x[1], qy = x[2], qz = x[3];
    const T s2 = 2.0 * mi_rcp(qw * qw + qx * qx + qy * qy + qz * qz);
    R[0][0] = 1.0 - s2 * (qy * qy + qz * qz); R[0][1] = s2 * (qx * qy - qw * qz); R[0][2] = s2 * (qx * qz + qw * qy);
    R[1][0] = s2 * (qx * qy + qw * qz); R[1][1] = 1.0 - s2 * (qx * qx + qz * qz); R[1][2] = s2 * (qy * qz - qw * qx);
    R[2][0] = s2 * (qx * qz - qw * qy); R[2][1] = s2 * (qy * qz + qw * qx); R[2][2] = 1.0 - s2 * (qx * qx + qy * qy);
  }
  template <class T, class XA, class UA>
  __device__ static inline void leg(int k, const T (&R)[3][3], const XA& x, const UA& u, const double* p, LegOut<T>& o) {
    const double kc = p[1], sig = p[2], dn = p[3], mu = p[4], b_j = p[5];
    const double sx = k < 2 ? 1.0 : -1.0, sy = (k & 1) == 0 ? -1.0 : 1.0;
    const T a = x[7 + 3 * k], b = x[8 + 3 * k], c = x[9 + 3 * k];
    const T jd0 = x[25 + 3 * k], jd1 = x[26 + 3 * k], jd2 = x[27 + 3 * k];
    const T om0 = x[19], om1 = x[20], om2 = x[21];
    const T sa = mi_sin(a), ca = mi_cos(a), sb = mi_sin(b), cb = mi_cos(b), sbc = mi_sin(b + c), cbc = mi_cos(b + c);
    const T X = -(kL1 * sb + kL2 * sbc), Z = -(kL1 * cb + kL2 * cbc);
    const double Y = sy * kL0;
    const T fb0 = sx * kHipX + X, fb1 = sy * kHipY + (Y * ca - Z * sa), fb2 = Y * sa + Z * ca;
    const T Ja1 = -(Y * sa) - Z * ca, Ja2 = Y * ca - Z * sa;                     // Ja0 = 0
    const T Jb0 = Z, Jb1 = X * sa, Jb2 = -(X * ca);
    const T dXc = -(kL2 * cbc), dZc = kL2 * sbc;
    const T Jc0 = dXc, Jc1 = -(dZc * sa), Jc2 = dZc * ca;
    const T zf = x[6] + (R[2][0] * fb0 + R[2][1] * fb1 + R[2][2] * fb2);
    const T vb0 = om1 * fb2 - om2 * fb1 + (0.0 * jd0 + Jb0 * jd1 + Jc0 * jd2);
    const T vb1 = om2 * fb0 - om0 * fb2 + (Ja1 * jd0 + Jb1 * jd1 + Jc1 * jd2);
    const T vb2 = om0 * fb1 - om1 * fb0 + (Ja2 * jd0 + Jb2 * jd1 + Jc2 * jd2);
    const T vf0 = x[22] + (R[0][0] * vb0 + R[0][1] * vb1 + R[0][2] * vb2);
    const T vf1 = x[23] + (R[1][0] * vb0 + R[1][1] * vb1 + R[1][2] * vb2);
    const T vf2 = x[24] + (R[2][0] * vb0 + R[2][1] * vb1 + R[2][2] * vb2);
    const T fn0 = (kc * sig) * mi_softplus(-zf * (1.0 / sig));
    o.fw[0] = -(mu * fn0) * vf0; o.fw[1] = -(mu * fn0) * vf1; o.fw[2] = fn0 * (1.0 - dn * vf2);
    const T f0 = R[0][0] * o.fw[0] + R[1][0] * o.fw[1] + R[2][0] * o.fw[2];        // R^T f: the force in the body frame
    const T f1 = R[0][1] * o.fw[0] + R[1][1] * o.fw[1] + R[2][1] * o.fw[2];
    const T f2 = R[0][2] * o.fw[0] + R[1][2] * o.fw[1] + R[2][2] * o.fw[2];
    o.tq[0] = fb1 * f2 - fb2 * f1; o.tq[1] = fb2 * f0 - fb0 * f2; o.tq[2] = fb0 * f1 - fb1 * f0;
    o.ja[0] = (u[3 * k + 0] - b_j * jd0 + (0.0 * f0 + Ja1 * f1 + Ja2 * f2)) * (1.0 / p[11]);
    o.ja[1] = (u[3 * k + 1] - b_j * jd1 + (Jb0 * f0 + Jb1 * f1 + Jb2 * f2)) * (1.0 / p[12]);
    o.ja[2] = (u[3 * k + 2] - b_j * jd2 + (Jc0 * f0 + Jc1 * f1 + Jc2 * f2)) * (1.0 / p[13]);
  }
  // trunk update from the summed wrench F (world), Tq (body): writes quaternion, position, angular and linear velocity
  template <class T, class XA>
  __device__ static inline void trunk(const XA& x, const T (&F)[3], const T (&Tq)[3], T* xn, const double* p, double dt) {
    const double g = p[0], imt = 1.0 / p[7], Ix = p[8], Iy = p[9], Iz = p[10];
    const T qw = x[0], qx = x[1], qy = x[2], qz = x[3];
    const T om0 = x[19], om1 = x[20], om2 = x[21];
    const T al0 = F[0] * imt, al1 = F[1] * imt, al2 = F[2] * imt - g;
    const T aw0 = (Tq[0] - (Iz - Iy) * om1 * om2) * (1.0 / Ix), aw1 = (Tq[1] - (Ix - Iz) * om2 * om0) * (1.0 / Iy),
            aw2 = (Tq[2] - (Iy - Ix) * om0 * om1) * (1.0 / Iz);
    const T w0 = om0 + dt * aw0, w1 = om1 + dt * aw1, w2 = om2 + dt * aw2;
    const T v0 = x[22] + dt * al0, v1 = x[23] + dt * al1, v2 = x[24] + dt * al2;
    const double hd = 0.5 * dt;
    xn[0] = qw + hd * (-(qx * w0) - qy * w1 - qz * w2);
    xn[1] = qx + hd * (qw * w0 + qy * w2 - qz * w1);
    xn[2] = qy + hd * (qw * w1 + qz * w0 - qx * w2);
    xn[3] = qz + hd * (qw * w2 + qx * w1 - qy * w0);
    xn[4] = x[4] + dt * v0; xn[5] = x[5] + dt * v1; xn[6] = x[6] + dt * v2;
    xn[19] = w0; xn[20] = w1; xn[21] = w2; xn[22] = v0; xn[23] = v1; xn[24] = v2;
  }
  template <class T, class XA>
  __device__ static inline void joints(int k, const XA& x, const LegOut<T>& o, T* xn, double dt) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const T jdn = x[25 + 3 * k + i] + dt * o.ja[i];
      xn[25 + 3 * k + i] = jdn;
      xn[7 + 3 * k + i] = x[7 + 3 * k + i] + dt * jdn;
    }
  }
  template <class T>
  __device__ static inline void step(const T* x, const T* u, T* xn, const double* p, double dt) { step_acc<T>(x, u, xn, p, dt); }
  // x, u: anything indexable (pointers, or accessors that perturb / seed one entry on the fly)
  template <class T, class XA, class UA>
  __device__ static inline void step_acc(const XA& x, const UA& u, T* xn, const double* p, double dt) {
    T R[3][3];
    rotation<T>(x, R);
    T F[3], Tq[3];
    {
      LegOut<T> l0, l2;
      leg<T>(0, R, x, u, p, l0); joints<T>(0, x, l0, xn, dt);
      leg<T>(2, R, x, u, p, l2); joints<T>(2, x, l2, xn, dt);
#pragma unroll
      for (int i = 0; i < 3; ++i) { F[i] = l0.fw[i] + l2.fw[i]; Tq[i] = l0.tq[i] + l2.tq[i]; }
    }
    {
      LegOut<T> l1, l3;
      leg<T>(1, R, x, u, p, l1); joints<T>(1, x, l1, xn, dt);
      leg<T>(3, R, x, u, p, l3); joints<T>(3, x, l3, xn, dt);
#pragma unroll
      for (int i = 0; i < 3; ++i) { F[i] = F[i] + (l1.fw[i] + l3.fw[i]); Tq[i] = Tq[i] + (l1.tq[i] + l3.tq[i]); }
    }
    trunk<T>(x, F, Tq, xn, p, dt);
  }
  __device__ static inline bool infeasible_velocity(double vn, const double* p) { return !(fabs(vn) <= p[6]); }
};

// 7-joint arm pushing a free ball - the state kinova_gen3.py:52-70 / panda_fr3.py stack: 14 positions (7 joint angles | the
// ball's unit quaternion w,x,y,z | the ball's position) + 13 velocities (7 joint rates | the ball's angular | linear velocity,
// world frame), 7 joint torques: n = 27, m = 7.  The same formulas, in the same operation order, as
// oracle/models_np.py:arm27_kinematics / arm27_step and oracle/ilqr_oracle.c.  params [g, k, sigma, dn, mu, b_joint, m_ball,
// r_ball, r_ee, m_elbow, m_hand, I_shoulder, I_elbow, I_wrist, ee_off].  Gen3-shaped kinematics (joint axes alternate z, y,
// ... in the moving frame, link offsets along the local z axis); the actuators' reflected inertia stands for the joint-space
// inertia (the approximation Quad3D's legs make), the links' weight is kept as two point masses (elbow, hand); the ball is a
// rigid sphere with compliant contacts against the ground and against the hand's sphere (smooth penalty, normal damping,
// load-proportional viscous friction at the contact point); the hand receives the opposite force through J^T.
// Served by the mid-size workgroup-per-problem kernels (ilqr_large.hpp: mid_backward).
// The two contacts' softplus values (hand - ball, ball - ground).  ROW (the workgroup-per-problem rollouts: every lane of a 16-lane
// row runs the step on the same values): even lanes evaluate the first argument, odd lanes the second - ONE evaluation in the
// instruction stream instead of two (~90 of a step's ~750 instructions) - and lanes 0 / 1 share their results with the row;
// softplus is branch-free, so each value has the bits the scalar form gives.
template <bool ROW, class T>
__device__ __forceinline__ void softplus_pair(const T& za, const T& zb, T& sa, T& sb) {
  if constexpr (ROW && std::is_same<T, double>::value) {
    const double s_ = mi_softplus((threadIdx.x & 1) ? zb : za);
    sa = __builtin_amdgcn_update_dpp(s_, s_, 0x150, 0xF, 0xF, true);        // row_newbcast:0 / :1 (one v_mov_b64_dpp each)
    sb = __builtin_amdgcn_update_dpp(s_, s_, 0x151, 0xF, 0xF, true);
  } else {
    sa = mi_softplus(za);
    sb = mi_softplus(zb);
  }
}

struct Arm27 {
  static constexpr int n = 27, m = 7, n_params = 15;
  static constexpr bool kWholeStep = true;
  static constexpr double kH0 = 0.28, kL1 = 0.42, kL2 = 0.31, kL3 = 0.27;
  template <class T>
  __device__ static inline void cross(const T (&a)[3], const T (&b)[3], T (&o)[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
  }
  // The sines and cosines of the seven joint angles are the one part of a step that is independent per joint: the
  // workgroup-per-problem rollout evaluates them on fourteen lanes at once (ilqr_large.hpp: kTrigCooperative; sines on lanes
  // 0..6, cosines on lanes 8..14 of one instruction stream - fast_sin_or_cos, bitwise mi_sin / mi_cos) and hands them to
  // core(); step() evaluates them one after the other.  Same bits either way.
  static constexpr bool kTrigCooperative = true;
  static constexpr int kJoints = 7;
  template <class T>
  __device__ static inline void step(const T* x, const T* u, T* xn, const double* p, double dt) {
    T S[7], C[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) { S[i] = mi_sin(x[i]); C[i] = mi_cos(x[i]); }
    core<T>(S, C, x, u, xn, p, dt);
  }
  template <class T, bool ROW = false>
  __device__ static inline void core(const T (&S)[7], const T (&C)[7], const T* x, const T* u, T* xn, const double* p, double dt) {
    const double g = p[0], kc = p[1], sig = p[2], dn = p[3], mu = p[4], bj = p[5];
    const double mb = p[6], rb = p[7], re = p[8], m_el = p[9], m_hd = p[10];
    // ---- kinematics: hand and elbow points, joint axes and origins in the world frame
    T ex[3] = {T(1.0), T(0.0), T(0.0)}, ey[3] = {T(0.0), T(1.0), T(0.0)}, ez[3] = {T(0.0), T(0.0), T(1.0)};
    T pos[3] = {T(0.0), T(0.0), T(kH0)}, elbow[3], hand[3], axes[7][3], orgs[7][3];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const T s = S[i], c = C[i];
      if (i % 2 == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { axes[i][k] = ez[k]; orgs[i][k] = pos[k]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) { const T a = c * ex[k] + s * ey[k], b = c * ey[k] - s * ex[k]; ex[k] = a; ey[k] = b; }
      } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) { axes[i][k] = ey[k]; orgs[i][k] = pos[k]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) { const T a = c * ex[k] - s * ez[k], b = c * ez[k] + s * ex[k]; ex[k] = a; ez[k] = b; }
      }
      if (i == 2) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { pos[k] = pos[k] + kL1 * ez[k]; elbow[k] = pos[k]; }
      } else if (i == 4) {
#pragma unroll
        for (int k = 0; k < 3; ++k) pos[k] = pos[k] + kL2 * ez[k];
      }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) hand[k] = pos[k] + (p[14] * ex[k] + kL3 * ez[k]);
    T J[7][3], JEz[3];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      T r[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) r[k] = hand[k] - orgs[i][k];
      cross(axes[i], r, J[i]);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      T r[3], t3[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) r[k] = elbow[k] - orgs[i][k];
      cross(axes[i], r, t3);
      JEz[i] = t3[2];
    }
    const T* qd = x + 14;
    T vh[3], d[3], nr[3], om[3], vb[3], pb[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      vh[k] = ((J[0][k] * qd[0] + J[1][k] * qd[1]) + (J[2][k] * qd[2] + J[3][k] * qd[3])) + ((J[4][k] * qd[4] + J[5][k] * qd[5]) + J[6][k] * qd[6]);
      pb[k] = x[11 + k]; om[k] = x[21 + k]; vb[k] = x[24 + k];
      d[k] = pb[k] - hand[k];
    }
    // ---- hand - ball contact
    const T dist = mi_sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    const T idist = mi_rcp(dist);
#pragma unroll
    for (int k = 0; k < 3; ++k) nr[k] = d[k] * idist;
    const T phi = dist - (rb + re);
    T sp_c, sp_g;
    softplus_pair<ROW>(-phi * (1.0 / sig), -(pb[2] - rb) * (1.0 / sig), sp_c, sp_g);
    const T fn0 = (kc * sig) * sp_c;
    T wxn[3], rel[3], vt[3], Fc[3], nxv[3], tc[3];
    cross(om, nr, wxn);
#pragma unroll
    for (int k = 0; k < 3; ++k) rel[k] = vb[k] - rb * wxn[k] - vh[k];
    const T vn = rel[0] * nr[0] + rel[1] * nr[1] + rel[2] * nr[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) vt[k] = rel[k] - vn * nr[k];
    const T fnn = fn0 * (1.0 - dn * vn);
#pragma unroll
    for (int k = 0; k < 3; ++k) Fc[k] = fnn * nr[k] - (mu * fn0) * vt[k];
    cross(nr, vt, nxv);
#pragma unroll
    for (int k = 0; k < 3; ++k) tc[k] = (rb * mu) * fn0 * nxv[k];
    // ---- ball - ground contact
    const T fg0 = (kc * sig) * sp_g;
    const T vcx = vb[0] - rb * om[1], vcy = vb[1] + rb * om[0];
    const T Fg[3] = {-(mu * fg0) * vcx, -(mu * fg0) * vcy, fg0 * (1.0 - dn * vb[2])};
    const T tg[3] = {rb * Fg[1], -(rb * Fg[0]), T(0.0)};
    // ---- joints
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const double iI = 1.0 / (i < 2 ? p[11] : (i < 4 ? p[12] : p[13]));
      T grav = m_hd * J[i][2];
      if (i < 3) grav = grav + m_el * JEz[i];
      grav = g * grav;
      const T jf = J[i][0] * Fc[0] + J[i][1] * Fc[1] + J[i][2] * Fc[2];
      const T acc = (u[i] - bj * qd[i] - grav - jf) * iI;
      const T qdn = qd[i] + dt * acc;
      xn[14 + i] = qdn; xn[i] = x[i] + dt * qdn;
    }
    // ---- ball
    const double ib = 1.0 / (0.4 * mb * rb * rb), imb = 1.0 / mb;
    T omn[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      omn[k] = om[k] + dt * ((tc[k] + tg[k]) * ib);
      T al = (Fc[k] + Fg[k]) * imb;
      if (k == 2) al = al - g;
      const T vbn = vb[k] + dt * al;
      xn[11 + k] = pb[k] + dt * vbn; xn[21 + k] = omn[k]; xn[24 + k] = vbn;
    }
    const double hd = 0.5 * dt;
    const T qw = x[7], qx = x[8], qy = x[9], qz = x[10];
    xn[7] = qw + hd * (-(omn[0] * qx) - omn[1] * qy - omn[2] * qz);
    xn[8] = qx + hd * (qw * omn[0] + (omn[1] * qz - omn[2] * qy));
    xn[9] = qy + hd * (qw * omn[1] + (omn[2] * qx - omn[0] * qz));
    xn[10] = qz + hd * (qw * omn[2] + (omn[0] * qy - omn[1] * qx));
  }
};

// The arm + ball with COUPLED rigid-body joint dynamics (SURVEY (f)4: higher-fidelity articulated dynamics for the n = 27 shape;
// kinova_gen3.py:105-213 builds the real arm from its URDF).  State, kinematics, contacts and integrator are Arm27's; the seven
// joint accelerations solve the manipulator equation
//     M(q) qdd = tau - b qd - J_hand^T F_contact - sum_p m_p J_p^T (a_p + g e_z),      M = diag(I_rotor) + sum_p m_p J_p^T J_p
// over three point masses p (elbow, wrist, hand) that carry the links' inertia: J_p the points' Jacobians (3 / 5 / 7 columns), a_p
// their velocity-product accelerations (centripetal and Coriolis terms: from the links' angular velocities and the bias part of
// their angular accelerations, recursively along the chain), M factorized as L D L^T (no pivoting: positive definite).  params: the
// fifteen of Arm27 (I_* = rotor inertias on M's diagonal) + m_wrist.  The same formulas in the same operation order as
// oracle/models_np.py:arm27c_step and oracle/ilqr_oracle.c:arm27c_step.  Mid-size workgroup-per-problem kernels; the rollout takes the
// seven sines / cosines on fourteen lanes like Arm27's (kTrigCooperative).
struct Arm27C {
  static constexpr int n = 27, m = 7, n_params = 16;
  static constexpr bool kWholeStep = true;
  static constexpr bool kTrigCooperative = true;
#ifdef MI_ARM27C_PIVSPLIT
  static constexpr bool kPivSplit = true;
#endif
  static constexpr int kJoints = 7;
  template <class T>
  __device__ static inline void cross(const T (&a)[3], const T (&b)[3], T (&o)[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
  }
  template <class T>
  __device__ static inline T dot3(const T (&a)[3], const T (&b)[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
  template <class T>
  __device__ static inline void rot_acc(const T (&al)[3], const T (&w)[3], const T (&r)[3], T (&o)[3]) {
    T a1[3], t[3], a2[3];
    cross(al, r, a1); cross(w, r, t); cross(w, t, a2);
#pragma unroll
    for (int k = 0; k < 3; ++k) o[k] = a1[k] + a2[k];
  }
  template <class T>
  __device__ static inline void step(const T* x, const T* u, T* xn, const double* p, double dt) {
    T S[7], C[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) { S[i] = mi_sin(x[i]); C[i] = mi_cos(x[i]); }
    core<T>(S, C, x, u, xn, p, dt);
  }
  template <class T, bool ROW = false>
  __device__ static inline void core(const T (&S)[7], const T (&C)[7], const T* x, const T* u, T* xn, const double* p, double dt) {
    const double g = p[0], kc = p[1], sig = p[2], dn = p[3], mu = p[4], bj = p[5];
    const double mb = p[6], rb = p[7], re = p[8], m_el = p[9], m_hd = p[10], m_wr = p[15];
    // ---- kinematics: Arm27's, + the wrist point (origin of joints 5, 6)
    T ex[3] = {T(1.0), T(0.0), T(0.0)}, ey[3] = {T(0.0), T(1.0), T(0.0)}, ez[3] = {T(0.0), T(0.0), T(1.0)};
    T pos[3] = {T(0.0), T(0.0), T(Arm27::kH0)}, elbow[3], wrist[3], hand[3], axes[7][3], orgs[7][3];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const T s = S[i], c = C[i];
      if (i % 2 == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { axes[i][k] = ez[k]; orgs[i][k] = pos[k]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) { const T a = c * ex[k] + s * ey[k], b = c * ey[k] - s * ex[k]; ex[k] = a; ey[k] = b; }
      } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) { axes[i][k] = ey[k]; orgs[i][k] = pos[k]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) { const T a = c * ex[k] - s * ez[k], b = c * ez[k] + s * ex[k]; ex[k] = a; ez[k] = b; }
      }
      if (i == 2) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { pos[k] = pos[k] + Arm27::kL1 * ez[k]; elbow[k] = pos[k]; }
      } else if (i == 4) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { pos[k] = pos[k] + Arm27::kL2 * ez[k]; wrist[k] = pos[k]; }
      }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) hand[k] = pos[k] + (p[14] * ex[k] + Arm27::kL3 * ez[k]);
    T J[7][3], JW[5][3], JE[3][3];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      T r[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) r[k] = hand[k] - orgs[i][k];
      cross(axes[i], r, J[i]);
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      T r[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) r[k] = wrist[k] - orgs[i][k];
      cross(axes[i], r, JW[i]);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      T r[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) r[k] = elbow[k] - orgs[i][k];
      cross(axes[i], r, JE[i]);
    }
    T qd[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) qd[i] = x[14 + i];
    T vh[3], d[3], nr[3], om[3], vb[3], pb[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      vh[k] = ((J[0][k] * qd[0] + J[1][k] * qd[1]) + (J[2][k] * qd[2] + J[3][k] * qd[3])) + ((J[4][k] * qd[4] + J[5][k] * qd[5]) + J[6][k] * qd[6]);
      pb[k] = x[11 + k]; om[k] = x[21 + k]; vb[k] = x[24 + k];
      d[k] = pb[k] - hand[k];
    }
    // ---- hand - ball and ball - ground contacts: Arm27's
    const T dist = mi_sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    const T idist = mi_rcp(dist);
#pragma unroll
    for (int k = 0; k < 3; ++k) nr[k] = d[k] * idist;
    const T phi = dist - (rb + re);
    T sp_c, sp_g;
    softplus_pair<ROW>(-phi * (1.0 / sig), -(pb[2] - rb) * (1.0 / sig), sp_c, sp_g);
    const T fn0 = (kc * sig) * sp_c;
    T wxn[3], rel[3], vt[3], Fc[3], nxv[3], tc[3];
    cross(om, nr, wxn);
#pragma unroll
    for (int k = 0; k < 3; ++k) rel[k] = vb[k] - rb * wxn[k] - vh[k];
    const T vn = rel[0] * nr[0] + rel[1] * nr[1] + rel[2] * nr[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) vt[k] = rel[k] - vn * nr[k];
    const T fnn = fn0 * (1.0 - dn * vn);
#pragma unroll
    for (int k = 0; k < 3; ++k) Fc[k] = fnn * nr[k] - (mu * fn0) * vt[k];
    cross(nr, vt, nxv);
#pragma unroll
    for (int k = 0; k < 3; ++k) tc[k] = (rb * mu) * fn0 * nxv[k];
    const T fg0 = (kc * sig) * sp_g;
    const T vcx = vb[0] - rb * om[1], vcy = vb[1] + rb * om[0];
    const T Fg[3] = {-(mu * fg0) * vcx, -(mu * fg0) * vcy, fg0 * (1.0 - dn * vb[2])};
    const T tg[3] = {rb * Fg[1], -(rb * Fg[0]), T(0.0)};
    // ---- velocity-product accelerations of the three point masses (w_i = w_{i-1} + a_i qd_i, al_i = al_{i-1} + (w_{i-1} x a_i) qd_i)
    T w[3] = {T(0.0), T(0.0), T(0.0)}, al[3] = {T(0.0), T(0.0), T(0.0)};
    T w2[3], al2[3], w4[3], al4[3];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      T wxa[3];
      cross(w, axes[i], wxa);
#pragma unroll
      for (int k = 0; k < 3; ++k) al[k] = al[k] + wxa[k] * qd[i];
#pragma unroll
      for (int k = 0; k < 3; ++k) w[k] = w[k] + axes[i][k] * qd[i];
      if (i == 2) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { w2[k] = w[k]; al2[k] = al[k]; }
      } else if (i == 4) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { w4[k] = w[k]; al4[k] = al[k]; }
      }
    }
    T r2[3], r4[3], r6[3], aE[3], aW[3], aH[3], t4[3], t6[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { r2[k] = elbow[k] - orgs[2][k]; r4[k] = wrist[k] - elbow[k]; r6[k] = hand[k] - wrist[k]; }
    rot_acc(al2, w2, r2, aE);
    rot_acc(al4, w4, r4, t4);
#pragma unroll
    for (int k = 0; k < 3; ++k) aW[k] = aE[k] + t4[k];
    rot_acc(al, w, r6, t6);
#pragma unroll
    for (int k = 0; k < 3; ++k) aH[k] = aW[k] + t6[k];
    const T gE[3] = {aE[0], aE[1], aE[2] + g}, gW[3] = {aW[0], aW[1], aW[2] + g}, gH[3] = {aH[0], aH[1], aH[2] + g};
    // ---- right-hand side, mass matrix (lower triangle), L D L^T, the two triangular solves
    T rhs[7], Mm[7][7], L[7][7], dd[7], idd[7], y[7], acc[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      T h = m_hd * dot3(J[i], gH);
      if (i < 5) h = h + m_wr * dot3(JW[i < 5 ? i : 0], gW);
      if (i < 3) h = h + m_el * dot3(JE[i < 3 ? i : 0], gE);
      rhs[i] = u[i] - bj * qd[i] - h - dot3(J[i], Fc);
    }
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        T v = m_hd * dot3(J[i], J[j]);
        if (i < 5) v = v + m_wr * dot3(JW[i < 5 ? i : 0], JW[j < 5 ? j : 0]);
        if (i < 3) v = v + m_el * dot3(JE[i < 3 ? i : 0], JE[j < 3 ? j : 0]);
        if (i == j) v = v + (i < 2 ? p[11] : (i < 4 ? p[12] : p[13]));
        Mm[i][j] = v;
      }
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      T v = Mm[j][j];
#pragma unroll
      for (int k = 0; k < j; ++k) v = v - (L[j][k] * L[j][k]) * dd[k];
      dd[j] = v; idd[j] = mi_rcp(v);
#pragma unroll
      for (int i = j + 1; i < 7; ++i) {
        T v2 = Mm[i][j];
#pragma unroll
        for (int k = 0; k < j; ++k) v2 = v2 - (L[i][k] * L[j][k]) * dd[k];
        L[i][j] = v2 * idd[j];
      }
    }
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      T v = rhs[i];
#pragma unroll
      for (int k = 0; k < i; ++k) v = v - L[i][k] * y[k];
      y[i] = v;
    }
#pragma unroll
    for (int i = 6; i >= 0; --i) {
      T v = y[i] * idd[i];
#pragma unroll
      for (int k = i + 1; k < 7; ++k) v = v - L[k][i] * acc[k];
      acc[i] = v;
    }
#pragma unroll
    for (int i = 0; i < 7; ++i) { const T qdn = qd[i] + dt * acc[i]; xn[14 + i] = qdn; xn[i] = x[i] + dt * qdn; }
    // ---- ball: Arm27's
    const double ib = 1.0 / (0.4 * mb * rb * rb), imb = 1.0 / mb;
    T omn[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      omn[k] = om[k] + dt * ((tc[k] + tg[k]) * ib);
      T alb = (Fc[k] + Fg[k]) * imb;
      if (k == 2) alb = alb - g;
      const T vbn = vb[k] + dt * alb;
      xn[11 + k] = pb[k] + dt * vbn; xn[21 + k] = omn[k]; xn[24 + k] = vbn;
    }
    const double hd = 0.5 * dt;
    const T qw = x[7], qx = x[8], qy = x[9], qz = x[10];
    xn[7] = qw + hd * (-(omn[0] * qx) - omn[1] * qy - omn[2] * qz);
    xn[8] = qx + hd * (qw * omn[0] + (omn[1] * qz - omn[2] * qy));
    xn[9] = qy + hd * (qw * omn[1] + (omn[2] * qx - omn[0] * qz));
    xn[10] = qz + hd * (qw * omn[2] + (omn[0] * qy - omn[1] * qx));
  }
};

}  // namespace mi
