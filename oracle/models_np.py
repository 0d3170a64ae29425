"""Build-owned discrete-time dynamics models, CPU/oracle side (test infrastructure).

The reference gets x_{t+1}=f(x_t,u_t) from Drake's MultibodyPlant
(/root/reference/ilqr.py:223-229), which is a third-party C++ library that is
neither vendored under /root/reference nor installed here.
These closed-form models are therefore the BUILD's own model definitions; the
same formulas, in the same operation order, are implemented for the device in
drake_ddp_amd/csrc/models.hpp and in C in oracle/ilqr_oracle.c.  Parity is
pinned against /root/reference/ilqr.py *driven by these models* — never against
Drake.  All integrate with semi-implicit Euler: v+ = v + dt*a(q,v,u); q+ = q + dt*v+.

Every step function is written against the float-or-Dual primitives of
oracle/dual.py so the identical code yields exact Jacobians (AutoDiff analogue).

Model ids / parameter vectors (must match include/mi_ilqr.h and models.hpp):
  0 PENDULUM       n=2  m=1   [ml2, b, mgl]
  1 ACROBOT        n=4  m=1   [m1, m2, l1, lc1, lc2, Ic1, Ic2, b1, b2, g]
  2 CARTPOLE       n=4  m=1   [mc, mp, l, g]
  3 CARTPOLE_WALL  n=4  m=1   [mc, mp, l, g, wall_face_x, ball_radius, k, sigma]
  4 SYNTH36        n=36 m=12  [ks, c, kc, bu]
"""
import numpy as np

from . import dual as D

PENDULUM, ACROBOT, CARTPOLE, CARTPOLE_WALL, SYNTH36 = 0, 1, 2, 3, 4

MODEL_DIMS = {PENDULUM: (2, 1), ACROBOT: (4, 1), CARTPOLE: (4, 1),
              CARTPOLE_WALL: (4, 1), SYNTH36: (36, 12)}

MODEL_NAMES = {PENDULUM: "pendulum", ACROBOT: "acrobot", CARTPOLE: "cart_pole",
               CARTPOLE_WALL: "cart_pole_with_wall", SYNTH36: "synth36"}

DEFAULT_PARAMS = {
    # m=1, l=0.5, b=0.1, g=9.81 (the shape of pendulum.py's plant; SURVEY.md §8c anchor)
    PENDULUM: [0.25, 0.1, 4.905],
    # textbook acrobot (shape of acrobot.py's plant)
    ACROBOT: [1.0, 1.0, 1.0, 0.5, 1.0, 0.083, 0.33, 0.1, 0.1, 9.81],
    CARTPOLE: [10.0, 1.0, 0.5, 9.81],
    # wall box 0.1 thick welded at x=-0.5 (cart_pole_with_wall.py:79-97) -> face at
    # x=-0.45; ball radius 0.05 (:64); smooth penalty contact stands in for
    # hydroelastics: F = k*sigma*softplus(-phi/sigma)
    CARTPOLE_WALL: [10.0, 1.0, 0.5, 9.81, -0.45, 0.05, 2000.0, 0.01],
    SYNTH36: [4.0, 0.5, 6.0, 0.1],
}


def pendulum_step(x, u, p, dt):
    ml2, b, mgl = p[0], p[1], p[2]
    th, w = x[0], x[1]
    acc = (u[0] - b * w - mgl * D.sin(th)) / ml2
    wn = w + dt * acc
    thn = th + dt * wn
    return [thn, wn]


def acrobot_step(x, u, p, dt):
    m1, m2, l1, lc1, lc2, Ic1, Ic2, b1, b2, g = p[:10]
    q1, q2, v1, v2 = x[0], x[1], x[2], x[3]
    I1 = Ic1 + m1 * lc1 * lc1
    I2 = Ic2 + m2 * lc2 * lc2
    s1 = D.sin(q1)
    s2 = D.sin(q2)
    c2 = D.cos(q2)
    s12 = D.sin(q1 + q2)
    h = m2 * l1 * lc2
    M11 = I1 + I2 + m2 * l1 * l1 + 2.0 * h * c2
    M12 = I2 + h * c2
    M22 = I2
    # Coriolis/centrifugal bias C(q,v)v
    cb1 = -2.0 * h * s2 * v2 * v1 - h * s2 * v2 * v2
    cb2 = h * s2 * v1 * v1
    # gravity torques
    g1 = g * (m1 * lc1 + m2 * l1) * s1 + g * m2 * lc2 * s12
    g2 = g * m2 * lc2 * s12
    r1 = -cb1 - g1 - b1 * v1
    r2 = u[0] - cb2 - g2 - b2 * v2
    det = M11 * M22 - M12 * M12
    a1 = (M22 * r1 - M12 * r2) / det
    a2 = (M11 * r2 - M12 * r1) / det
    v1n = v1 + dt * a1
    v2n = v2 + dt * a2
    return [q1 + dt * v1n, q2 + dt * v2n, v1n, v2n]


def _cartpole_common(x, u, p, dt, wall):
    mc, mp, l, g = p[0], p[1], p[2], p[3]
    px, th, vx, w = x[0], x[1], x[2], x[3]
    s = D.sin(th)
    c = D.cos(th)
    # theta = 0 is the pole hanging down, theta = pi upright
    M11 = mc + mp
    M12 = mp * l * c
    M22 = mp * l * l
    r1 = u[0] + mp * l * w * w * s
    r2 = -mp * g * l * s
    if wall:
        face, rad, k, sig = p[4], p[5], p[6], p[7]
        tip = px + l * s
        phi = tip - rad - face            # signed distance ball surface -> wall face
        F = k * sig * D.softplus(-phi / sig)   # pushes the ball in +x
        r1 = r1 + F
        r2 = r2 + F * l * c               # J^T F with J = [1, l cos(theta)]
    det = M11 * M22 - M12 * M12
    a1 = (M22 * r1 - M12 * r2) / det
    a2 = (M11 * r2 - M12 * r1) / det
    vxn = vx + dt * a1
    wn = w + dt * a2
    return [px + dt * vxn, th + dt * wn, vxn, wn]


def cartpole_step(x, u, p, dt):
    return _cartpole_common(x, u, p, dt, False)


def cartpole_wall_step(x, u, p, dt):
    return _cartpole_common(x, u, p, dt, True)


def synth36_step(x, u, p, dt):
    """Smooth 18-dof chain of coupled pendula; dofs 6..17 actuated, dofs 0..5
    ('floating base') driven only through coupling.  Cheetah-SHAPED (n=36,m=12),
    not a quadruped model (SURVEY.md §7 hard parts)."""
    ks, c, kc, bu = p[0], p[1], p[2], p[3]
    nq = 18
    q = x[:nq]
    v = x[nq:]
    sq = [D.sin(q[i]) for i in range(nq)]
    # link[i] = sin(q[i+1]-q[i]), i = 0..16
    link = [D.sin(q[i + 1] - q[i]) for i in range(nq - 1)]
    out_q = [None] * nq
    out_v = [None] * nq
    for i in range(nq):
        a = -ks * sq[i] - c * v[i]
        if i < nq - 1:
            a = a + kc * link[i]
        if i > 0:
            a = a - kc * link[i - 1]
        if i >= 6:
            a = a + u[i - 6]
        else:
            a = a + bu * (u[2 * i] - u[2 * i + 1])
        vn = v[i] + dt * a
        out_v[i] = vn
        out_q[i] = q[i] + dt * vn
    return out_q + out_v


STEP_FUNCS = {PENDULUM: pendulum_step, ACROBOT: acrobot_step, CARTPOLE: cartpole_step,
              CARTPOLE_WALL: cartpole_wall_step, SYNTH36: synth36_step}


class Model:
    """A discrete-time model x+ = f(x,u): id, parameter vector and time step."""

    def __init__(self, model_id, dt, params=None):
        self.model_id = int(model_id)
        self.n, self.m = MODEL_DIMS[self.model_id]
        self.dt = float(dt)
        self.params = np.array(DEFAULT_PARAMS[self.model_id] if params is None else params,
                               dtype=float)
        self._f = STEP_FUNCS[self.model_id]

    def step(self, x, u):
        """Next state for float inputs -> (n,) float array."""
        return np.array(self._f(list(x), list(u), self.params, self.dt), dtype=float)

    def step_generic(self, x, u):
        """Next state for float-or-Dual inputs -> list."""
        return self._f(list(x), list(u), self.params, self.dt)

    def jac_ad(self, x, u):
        """Exact (fx, fu) by forward-mode duals — AutoDiffXd analogue
        (/root/reference/ilqr.py:233-272)."""
        xu = D.seed(np.concatenate([np.asarray(x, float).ravel(), np.asarray(u, float).ravel()]))
        G = D.gradient(self._f(xu[:self.n], xu[self.n:], self.params, self.dt))
        return G[:, :self.n].copy(), G[:, self.n:].copy()

    def jac_fd(self, x, u, h):
        """Central finite differences with absolute step h — the formula the HIP
        linearization uses (drake_ddp_amd/csrc/ilqr_small.hip)."""
        x = np.asarray(x, float).ravel()
        u = np.asarray(u, float).ravel()
        n, m = self.n, self.m
        inv2h = 1.0 / (2.0 * h)
        G = np.zeros((n, n + m))
        for c in range(n + m):
            xp, up, xm, um = x.copy(), u.copy(), x.copy(), u.copy()
            if c < n:
                xp[c] = x[c] + h
                xm[c] = x[c] - h
            else:
                up[c - n] = u[c - n] + h
                um[c - n] = u[c - n] - h
            G[:, c] = (self.step(xp, up) - self.step(xm, um)) * inv2h
        return G[:, :n].copy(), G[:, n:].copy()
