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
  template <class T>
  __device__ static inline void step(const T* x, const T* u, T* xn, const double* p, double dt) {
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
      const T F = k * sig * mi_softplus(-phi * inv_sig);
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

}  // namespace mi
