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
  5 PLANAR_QUAD    n=36 m=12  [g, k, sigma, dn, mu, b_leg, b_tail, k_tail, v_max]   (articulated, contact, can FAIL)
  6 QUAD3D         n=37 m=12  [g, k, sigma, dn, mu, b_joint, v_max, m_trunk, Ixx, Iyy, Izz, I_abad, I_hip, I_knee]
                              (3-D floating base with a unit-quaternion attitude, contact at four feet, can FAIL)
  7 ARM27          n=27 m=7   [g, k, sigma, dn, mu, b_joint, m_ball, r_ball, r_ee, m_elbow, m_hand, I_shoulder, I_elbow, I_wrist, ee_off]
                              (7-joint arm + a free ball with a unit-quaternion attitude: the state kinova_gen3.py:52-70 stacks)
  8 ARM27C         n=27 m=7   the same arm, ball and contacts with COUPLED rigid-body joint dynamics: M(q) qdd + c(q, qd) + g(q) = tau - b qd - J^T F
                              [..the 15 of ARM27 (I_* = rotor inertias on M's diagonal).., m_wrist]

A model may declare a step INFEASIBLE (the analogue of Drake's discrete update throwing): ``Model.step``
then raises RuntimeError, which the reference's line search catches (ilqr.py:315-323).
"""
import numpy as np

from . import dual as D

PENDULUM, ACROBOT, CARTPOLE, CARTPOLE_WALL, SYNTH36, PLANAR_QUAD, QUAD3D, ARM27, ARM27C = 0, 1, 2, 3, 4, 5, 6, 7, 8

MODEL_DIMS = {PENDULUM: (2, 1), ACROBOT: (4, 1), CARTPOLE: (4, 1),
              CARTPOLE_WALL: (4, 1), SYNTH36: (36, 12), PLANAR_QUAD: (36, 12), QUAD3D: (37, 12), ARM27: (27, 7), ARM27C: (27, 7)}

MODEL_NAMES = {PENDULUM: "pendulum", ACROBOT: "acrobot", CARTPOLE: "cart_pole",
               CARTPOLE_WALL: "cart_pole_with_wall", SYNTH36: "synth36", PLANAR_QUAD: "planar_quadruped",
               QUAD3D: "quadruped_3d", ARM27: "arm_and_ball", ARM27C: "arm_and_ball_coupled"}

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
    # gravity; ground penalty k*sigma*softplus(-z/sigma) with normal damping dn and load-proportional viscous
    # friction mu; joint damping of the legs / tail, tail spring; |v| bound beyond which a step is infeasible
    PLANAR_QUAD: [9.81, 4000.0, 0.004, 0.3, 0.15, 0.05, 0.02, 2.0, 60.0],
    # gravity; ground penalty k*sigma*softplus(-z/sigma), normal damping dn, load-proportional viscous friction mu;
    # joint damping; |v| bound beyond which a step is infeasible; trunk mass and principal inertias (mini-cheetah
    # sized: 9 kg); reflected actuator inertias of the ab/ad, hip and knee joints
    QUAD3D: [9.81, 4000.0, 0.004, 0.3, 0.15, 0.3, 60.0, 9.0, 0.07, 0.26, 0.28, 0.06, 0.06, 0.04],
    # gravity; contact penalty k*sigma*softplus(-phi/sigma) (ball-ground and hand-ball), normal damping dn, load-proportional
    # viscous friction mu; joint damping; ball mass and radius; hand (end-effector sphere) radius; the two point masses that
    # carry the links' weight (elbow, hand); reflected actuator inertias of the joint pairs (0,1), (2,3), (4,5,6); lateral
    # offset of the hand point from the last joint's axis
    ARM27: [9.81, 1500.0, 0.005, 0.5, 1.0, 0.5, 0.2, 0.1, 0.05, 1.0, 0.8, 0.6, 0.3, 0.1, 0.04],
    # the same, with the links' inertia carried by three point masses (elbow, wrist, hand) through the joint-space mass matrix and the
    # I_* entries as rotor (armature) inertias on its diagonal; m_wrist is the sixteenth parameter
    ARM27C: [9.81, 1500.0, 0.005, 0.5, 1.0, 0.5, 0.2, 0.1, 0.05, 1.0, 0.8, 0.3, 0.15, 0.05, 0.04, 0.6],
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


# ----------------------------------------------------------------------------------------------------
# PLANAR_QUAD: a planar floating-base articulated body of the mini-cheetah's SHAPE (mini_cheetah.py:41-52:
# 18 positions + 18 velocities, 12 actuators): trunk (x, z, pitch) + four 3-link legs (hip, knee, ankle:
# the 12 actuated joints) + a passive 3-link tail, 16 bodies, compliant ground contact at the four feet and
# the tail tip.  Accelerations come from the articulated-body algorithm (Featherstone) written in
# world-aligned planar coordinates: a body's spatial quantities are (moment about / rotation at its joint
# axis p_i, x, z), transforms between bodies are pure translations.  Build-owned like every model here.
#   q = [x, z, pitch | leg0 hip,knee,ankle | leg1 | leg2 | leg3 | tail 1,2,3],  u -> the 12 leg joints.
# ----------------------------------------------------------------------------------------------------
QUAD_NQ = 18
# body table: (parent, attach offset in the parent's frame (x, z), length, mass, inertia about the COM)
_TH, _SH, _FT, _TL = (0.20, 0.60, 0.60 * 0.20 ** 2 / 12), (0.18, 0.40, 0.40 * 0.18 ** 2 / 12), \
    (0.14, 0.30, 0.30 * 0.14 ** 2 / 12), (0.12, 0.10, 0.10 * 0.12 ** 2 / 12)
QUAD_TRUNK = (4.0, 0.06)                       # mass, inertia
QUAD_HIPS = [(0.19, 0.0), (0.19, 0.0), (-0.19, 0.0), (-0.19, 0.0)]
QUAD_TAIL_AT = (-0.25, 0.02)
QUAD_TAIL_REST = (-1.2, -0.2, -0.2)


def quad_bodies():
    """[(parent, (ax, az) or None = parent's tip, length, mass, Ic)] for bodies 1..15 (body 0 = trunk)."""
    out = []
    for leg in range(4):
        base = 1 + 3 * leg
        out.append((0, QUAD_HIPS[leg]) + _TH)
        out.append((base, None) + _SH)
        out.append((base + 1, None) + _FT)
    out.append((0, QUAD_TAIL_AT) + _TL)
    out.append((13, None) + _TL)
    out.append((14, None) + _TL)
    return out


def quad_accel(x, u, p):
    """Generalized accelerations (18) of the planar quadruped, float-or-Dual."""
    g, kc, sig, dn, mu, b_leg, b_tail, k_tail = p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7]
    nq = QUAD_NQ
    q, v = x[:nq], x[nq:]
    bodies = quad_bodies()
    nb = 16
    th = [None] * nb; om = [None] * nb; px = [None] * nb; pz = [None] * nb; vx = [None] * nb; vz = [None] * nb
    sn = [None] * nb; cs = [None] * nb; rx = [None] * nb; rz = [None] * nb
    dx = [None] * nb; dz = [None] * nb; cbx = [None] * nb; cbz = [None] * nb
    mass = [QUAD_TRUNK[0]] + [b[3] for b in bodies]
    inert = [QUAD_TRUNK[1]] + [b[4] for b in bodies]
    length = [0.0] + [b[2] for b in bodies]
    th[0], om[0], px[0], pz[0], vx[0], vz[0] = q[2], v[2], q[0], q[1], v[0], v[1]
    sn[0], cs[0] = D.sin(th[0]), D.cos(th[0])
    rx[0], rz[0] = 0.0, 0.0
    # ---- pass 1: kinematics (joint axis positions / velocities, COM offsets, velocity-product accelerations)
    for i in range(1, nb):
        par, at = bodies[i - 1][0], bodies[i - 1][1]
        if at is None:                                   # at the parent's tip: (0, -l_par) in its frame
            lp = length[par]
            dx[i], dz[i] = lp * sn[par], -lp * cs[par]
        else:
            dx[i] = cs[par] * at[0] - sn[par] * at[1]
            dz[i] = sn[par] * at[0] + cs[par] * at[1]
        th[i] = th[par] + q[2 + i]
        om[i] = om[par] + v[2 + i]
        sn[i], cs[i] = D.sin(th[i]), D.cos(th[i])
        px[i], pz[i] = px[par] + dx[i], pz[par] + dz[i]
        vx[i], vz[i] = vx[par] - om[par] * dz[i], vz[par] + om[par] * dx[i]
        hl = 0.5 * length[i]
        rx[i], rz[i] = hl * sn[i], -hl * cs[i]
        w2 = om[par] * om[par]
        cbx[i], cbz[i] = -w2 * dx[i], -w2 * dz[i]
    # ---- joint torques: actuation, damping, tail spring
    tau = [None] * nb
    for i in range(1, 13):
        tau[i] = u[i - 1] - b_leg * v[2 + i]
    for k in range(3):
        i = 13 + k
        tau[i] = -b_tail * v[2 + i] - k_tail * (q[2 + i] - QUAD_TAIL_REST[k])
    # ---- articulated inertias (symmetric 3x3: [J, hx, hz; hx, mxx, mxz; hz, mxz, mzz]) and bias forces about p_i
    J = [None] * nb; hx = [None] * nb; hz = [None] * nb; mxx = [None] * nb; mxz = [None] * nb; mzz = [None] * nb
    bn = [None] * nb; bx = [None] * nb; bz = [None] * nb
    for i in range(nb):
        m_, w2 = mass[i], om[i] * om[i]
        J[i] = inert[i] + m_ * (rx[i] * rx[i] + rz[i] * rz[i])
        hx[i], hz[i] = -m_ * rz[i], m_ * rx[i]
        mxx[i], mxz[i], mzz[i] = m_, 0.0, m_
        # centripetal term of the COM minus gravity (force (0, -m g) at the COM)
        bn[i] = m_ * g * rx[i]
        bx[i] = -m_ * w2 * rx[i]
        bz[i] = -m_ * w2 * rz[i] + m_ * g
    # ---- ground contact at the feet (bodies 3, 6, 9, 12) and the tail tip (15)
    for i in (3, 6, 9, 12, 15):
        ex, ez = 2.0 * rx[i], 2.0 * rz[i]                 # tip relative to the joint axis
        tz = pz[i] + ez
        tvx, tvz = vx[i] - om[i] * ez, vz[i] + om[i] * ex
        fn0 = kc * sig * D.softplus(-tz / sig)
        fn = fn0 * (1.0 - dn * tvz)
        ft = -mu * fn0 * tvx
        bn[i] = bn[i] - (ex * fn - ez * ft)
        bx[i] = bx[i] - ft
        bz[i] = bz[i] - fn
    # ---- pass 2: leaves to root
    Dj = [None] * nb; Ux = [None] * nb; Uz = [None] * nb; uu = [None] * nb
    for i in range(nb - 1, 0, -1):
        par = bodies[i - 1][0]
        Dj[i] = J[i]
        Ux[i], Uz[i] = hx[i], hz[i]
        uu[i] = tau[i] - bn[i]
        invD = 1.0 / Dj[i]
        # articulated inertia seen through the joint: only the translational block survives
        exx = mxx[i] - Ux[i] * Ux[i] * invD
        exz = mxz[i] - Ux[i] * Uz[i] * invD
        ezz = mzz[i] - Uz[i] * Uz[i] * invD
        s_ = uu[i] * invD
        fx = bx[i] + exx * cbx[i] + exz * cbz[i] + Ux[i] * s_
        fz = bz[i] + exz * cbx[i] + ezz * cbz[i] + Uz[i] * s_
        # shift to the parent's axis: G = [perp(d) | I], perp(d) = (-dz, dx)
        gx = -exx * dz[i] + exz * dx[i]                  # E perp(d)
        gz = -exz * dz[i] + ezz * dx[i]
        J[par] = J[par] + (-dz[i] * gx + dx[i] * gz)
        hx[par] = hx[par] + gx
        hz[par] = hz[par] + gz
        mxx[par] = mxx[par] + exx
        mxz[par] = mxz[par] + exz
        mzz[par] = mzz[par] + ezz
        bn[par] = bn[par] + tau[i] - dz[i] * fx + dx[i] * fz
        bx[par] = bx[par] + fx
        bz[par] = bz[par] + fz
    # ---- floating base: I_A a = -p_A  (3x3 symmetric, eliminated in the order x, z, pitch... by Cramer-free LDL^T)
    a11, a12, a13, a22, a23, a33 = mxx[0], mxz[0], hx[0], mzz[0], hz[0], J[0]      # unknowns (ax, az, alpha)
    r1, r2, r3 = -bx[0], -bz[0], -bn[0]
    l21 = a12 / a11
    l31 = a13 / a11
    d2 = a22 - l21 * a12
    e23 = a23 - l21 * a13
    l32 = e23 / d2
    d3 = a33 - l31 * a13 - l32 * e23
    y2 = r2 - l21 * r1
    y3 = r3 - l31 * r1 - l32 * y2
    alpha = y3 / d3
    az = (y2 - e23 * alpha) / d2
    ax = (r1 - a12 * az - a13 * alpha) / a11
    # ---- pass 3: root to leaves
    al = [None] * nb; acx = [None] * nb; acz = [None] * nb; qdd = [None] * nq
    al[0], acx[0], acz[0] = alpha, ax, az
    qdd[0], qdd[1], qdd[2] = ax, az, alpha
    for i in range(1, nb):
        par = bodies[i - 1][0]
        apx = acx[par] - al[par] * dz[i] + cbx[i]
        apz = acz[par] + al[par] * dx[i] + cbz[i]
        qi = (uu[i] - (Dj[i] * al[par] + Ux[i] * apx + Uz[i] * apz)) / Dj[i]
        qdd[2 + i] = qi
        al[i], acx[i], acz[i] = al[par] + qi, apx, apz
    return qdd


def planar_quad_step(x, u, p, dt):
    nq = QUAD_NQ
    qdd = quad_accel(x, u, p)
    vn = [x[nq + i] + dt * qdd[i] for i in range(nq)]
    qn = [x[i] + dt * vn[i] for i in range(nq)]
    return qn + vn


def planar_quad_infeasible(xn, p):
    """The step is declared infeasible (Drake's update would throw: ilqr.py:315-323) when a velocity leaves
    [-v_max, v_max] or is not finite - the regime where the penalty contact model means nothing."""
    vmax = p[8]
    for i in range(QUAD_NQ, 2 * QUAD_NQ):
        val = xn[i].v if isinstance(xn[i], D.Dual) else xn[i]
        if not (abs(val) <= vmax):
            return True
    return False


# ----------------------------------------------------------------------------------------------------
# QUAD3D: a 3-D floating-base quadruped with the state layout of mini_cheetah.py:41-52 - 19 positions
# (unit quaternion w,x,y,z | base position | 4 x (ab/ad, hip, knee)) + 18 velocities (angular | linear | joint
# rates), 12 actuators: n = 37, m = 12.  Build-owned like every model here (Drake is absent): the trunk is a 3-D
# rigid body (Euler's equations, body-frame angular velocity, the attitude quaternion integrated as
# q+ = q + dt/2 q (x) (0, w+) and - like the reference's plain-vector iLQR - never renormalized inside a step);
# the legs carry 3-D kinematics (ab/ad about the body x axis, hip and knee about the rotated y axis; link lengths
# of the mini cheetah) and their links are massless: each joint has its actuator's reflected inertia, a foot's
# contact force reaches the trunk as a wrench and the joints through J^T f.  Compliant ground contact at the four
# feet (mini_cheetah.py:92-101 uses hydroelastics; here the smooth penalty of the other contact models).
# The legs' contributions are summed in the order (leg0 + leg2) + (leg1 + leg3): the order a 16-lane DPP row sum
# produces on the device, where one lane per leg evaluates them.
# ----------------------------------------------------------------------------------------------------
Q3_NQ, Q3_NV = 19, 18
Q3_L0, Q3_L1, Q3_L2 = 0.062, 0.209, 0.195
Q3_HIPX, Q3_HIPY = 0.19, 0.049


def quad3d_leg(k, quat_R, om, vlin, pz, q, jd, u, p):
    """Leg k: foot kinematics, contact force, joint accelerations; returns (f_world[3], torque_body[3], jacc[3])."""
    kc, sig, dn, mu, b_j = p[1], p[2], p[3], p[4], p[5]
    Ij = (p[11], p[12], p[13])
    sx = 1.0 if k < 2 else -1.0
    sy = -1.0 if (k % 2) == 0 else 1.0
    a, b, c = q[0], q[1], q[2]
    sa, ca = D.sin(a), D.cos(a)
    sb, cb = D.sin(b), D.cos(b)
    sbc, cbc = D.sin(b + c), D.cos(b + c)
    X = -(Q3_L1 * sb + Q3_L2 * sbc)
    Z = -(Q3_L1 * cb + Q3_L2 * cbc)
    Y = sy * Q3_L0
    fb = [sx * Q3_HIPX + X, sy * Q3_HIPY + (Y * ca - Z * sa), Y * sa + Z * ca]      # foot in the body frame
    Ja = [0.0, -(Y * sa) - Z * ca, Y * ca - Z * sa]
    Jb = [Z, X * sa, -(X * ca)]
    dXc, dZc = -(Q3_L2 * cbc), Q3_L2 * sbc
    Jc = [dXc, -(dZc * sa), dZc * ca]
    R = quat_R
    # foot height and velocity in the world frame
    zf = pz + (R[2][0] * fb[0] + R[2][1] * fb[1] + R[2][2] * fb[2])
    vb = [om[1] * fb[2] - om[2] * fb[1] + (Ja[0] * jd[0] + Jb[0] * jd[1] + Jc[0] * jd[2]),
          om[2] * fb[0] - om[0] * fb[2] + (Ja[1] * jd[0] + Jb[1] * jd[1] + Jc[1] * jd[2]),
          om[0] * fb[1] - om[1] * fb[0] + (Ja[2] * jd[0] + Jb[2] * jd[1] + Jc[2] * jd[2])]
    vf = [vlin[i] + (R[i][0] * vb[0] + R[i][1] * vb[1] + R[i][2] * vb[2]) for i in range(3)]
    fn0 = kc * sig * D.softplus(-zf / sig)
    fw = [-(mu * fn0) * vf[0], -(mu * fn0) * vf[1], fn0 * (1.0 - dn * vf[2])]
    fbd = [R[0][i] * fw[0] + R[1][i] * fw[1] + R[2][i] * fw[2] for i in range(3)]  # R^T f: the force in the body frame
    tq = [fb[1] * fbd[2] - fb[2] * fbd[1], fb[2] * fbd[0] - fb[0] * fbd[2], fb[0] * fbd[1] - fb[1] * fbd[0]]
    jacc = [(u[0] - b_j * jd[0] + (Ja[0] * fbd[0] + Ja[1] * fbd[1] + Ja[2] * fbd[2])) / Ij[0],
            (u[1] - b_j * jd[1] + (Jb[0] * fbd[0] + Jb[1] * fbd[1] + Jb[2] * fbd[2])) / Ij[1],
            (u[2] - b_j * jd[2] + (Jc[0] * fbd[0] + Jc[1] * fbd[1] + Jc[2] * fbd[2])) / Ij[2]]
    return fw, tq, jacc


def quad3d_step(x, u, p, dt):
    g, mt, Ix, Iy, Iz = p[0], p[7], p[8], p[9], p[10]
    qw, qx, qy, qz = x[0], x[1], x[2], x[3]
    pos = x[4:7]
    jq = x[7:19]
    om = x[19:22]
    vl = x[22:25]
    jd = x[25:37]
    s2 = 2.0 / (qw * qw + qx * qx + qy * qy + qz * qz)
    R = [[1.0 - s2 * (qy * qy + qz * qz), s2 * (qx * qy - qw * qz), s2 * (qx * qz + qw * qy)],
         [s2 * (qx * qy + qw * qz), 1.0 - s2 * (qx * qx + qz * qz), s2 * (qy * qz - qw * qx)],
         [s2 * (qx * qz - qw * qy), s2 * (qy * qz + qw * qx), 1.0 - s2 * (qx * qx + qy * qy)]]
    legs = [quad3d_leg(k, R, om, vl, pos[2], jq[3 * k:3 * k + 3], jd[3 * k:3 * k + 3], u[3 * k:3 * k + 3], p) for k in range(4)]
    F = [(legs[0][0][i] + legs[2][0][i]) + (legs[1][0][i] + legs[3][0][i]) for i in range(3)]
    T = [(legs[0][1][i] + legs[2][1][i]) + (legs[1][1][i] + legs[3][1][i]) for i in range(3)]
    al = [F[0] / mt, F[1] / mt, F[2] / mt - g]
    aw = [(T[0] - (Iz - Iy) * om[1] * om[2]) / Ix, (T[1] - (Ix - Iz) * om[2] * om[0]) / Iy, (T[2] - (Iy - Ix) * om[0] * om[1]) / Iz]
    omn = [om[i] + dt * aw[i] for i in range(3)]
    vln = [vl[i] + dt * al[i] for i in range(3)]
    jdn = [jd[3 * k + i] + dt * legs[k][2][i] for k in range(4) for i in range(3)]
    hd = 0.5 * dt
    qn = [qw + hd * (-(qx * omn[0]) - qy * omn[1] - qz * omn[2]),
          qx + hd * (qw * omn[0] + qy * omn[2] - qz * omn[1]),
          qy + hd * (qw * omn[1] + qz * omn[0] - qx * omn[2]),
          qz + hd * (qw * omn[2] + qx * omn[1] - qy * omn[0])]
    pn = [pos[i] + dt * vln[i] for i in range(3)]
    jn = [jq[i] + dt * jdn[i] for i in range(12)]
    return qn + pn + jn + omn + vln + jdn


def quad3d_infeasible(xn, p):
    """Like planar_quad_infeasible: a velocity outside [-v_max, v_max] (or not finite) makes the step infeasible."""
    vmax = p[6]
    for i in range(Q3_NQ, Q3_NQ + Q3_NV):
        val = xn[i].v if isinstance(xn[i], D.Dual) else xn[i]
        if not (abs(val) <= vmax):
            return True
    return False


# ----------------------------------------------------------------------------------------------------
# ARM27: a 7-joint arm that pushes a free ball - the state kinova_gen3.py:52-70 / panda_fr3.py stack: 14 positions (7 joint
# angles | the ball's unit quaternion w,x,y,z | the ball's position) + 13 velocities (7 joint rates | the ball's angular |
# linear velocity, world frame), 7 joint torques: n = 27, m = 7.  Build-owned like every model here (Drake and its URDFs are
# absent).  Kinematics of a Gen3-shaped arm: joint axes alternate z, y, z, y, z, y, z in the moving frame, link offsets
# along the local z axis (shoulder height, upper arm, forearm, hand); the hand point sits ee_off beside the last axis so
# that every joint moves it.  Dynamics: the actuators' reflected inertia dominates the links' own (harmonic drives: the
# approximation Quad3D's legs also make), so each joint is I_j q''_j = u_j - b q'_j + gravity_j + (J^T f)_j; the links' WEIGHT
# is kept as two point masses (elbow, hand).  The ball is a rigid sphere (I = 2/5 m r^2) under gravity, a compliant contact
# with the ground and one with the hand's sphere (smooth penalty along the centre line, normal damping, load-proportional
# viscous friction acting at the contact point, so the ball rolls); the hand receives the opposite force through J^T.
# The attitude quaternion is integrated as q+ = q + dt/2 (0, w+) (x) q and never renormalized inside a step.
# ----------------------------------------------------------------------------------------------------
A27_H0, A27_L1, A27_L2, A27_L3 = 0.28, 0.42, 0.31, 0.27


def _cross(a, b):
    return [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]


def arm27_kinematics(q, p):
    """Hand point, elbow point, joint axes a[7] and joint origins o[7] in the world frame."""
    ex, ey, ez = [1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]
    pos = [0.0, 0.0, A27_H0]
    axes, orgs = [], []
    elbow = None
    for i in range(7):
        s, c = D.sin(q[i]), D.cos(q[i])
        if i % 2 == 0:                        # about the local z axis
            axes.append(list(ez)); orgs.append(list(pos))
            ex, ey = [c * ex[k] + s * ey[k] for k in range(3)], [c * ey[k] - s * ex[k] for k in range(3)]
        else:                                 # about the local y axis
            axes.append(list(ey)); orgs.append(list(pos))
            ex, ez = [c * ex[k] - s * ez[k] for k in range(3)], [c * ez[k] + s * ex[k] for k in range(3)]
        if i == 2:
            pos = [pos[k] + A27_L1 * ez[k] for k in range(3)]
            elbow = list(pos)
        elif i == 4:
            pos = [pos[k] + A27_L2 * ez[k] for k in range(3)]
    hand = [pos[k] + (p[14] * ex[k] + A27_L3 * ez[k]) for k in range(3)]
    return hand, elbow, axes, orgs


def arm27_step(x, u, p, dt):
    g, kc, sig, dn, mu, bj = p[0], p[1], p[2], p[3], p[4], p[5]
    mb, rb, re, m_el, m_hd = p[6], p[7], p[8], p[9], p[10]
    Ij = [p[11], p[11], p[12], p[12], p[13], p[13], p[13]]
    q, qd = x[0:7], x[14:21]
    qw, qx, qy, qz = x[7], x[8], x[9], x[10]
    pb, om, vb = x[11:14], x[21:24], x[24:27]
    hand, elbow, axes, orgs = arm27_kinematics(q, p)
    # Jacobian columns of the hand point (all joints) and of the elbow point (joints 0..2)
    J = [_cross(axes[i], [hand[k] - orgs[i][k] for k in range(3)]) for i in range(7)]
    JEz = [_cross(axes[i], [elbow[k] - orgs[i][k] for k in range(3)])[2] for i in range(3)]
    vh = [((J[0][k] * qd[0] + J[1][k] * qd[1]) + (J[2][k] * qd[2] + J[3][k] * qd[3])) + ((J[4][k] * qd[4] + J[5][k] * qd[5]) + J[6][k] * qd[6])
          for k in range(3)]
    # hand - ball contact
    d = [pb[k] - hand[k] for k in range(3)]
    dist = D.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])
    idist = 1.0 / dist
    nr = [d[k] * idist for k in range(3)]
    phi = dist - (rb + re)
    fn0 = (kc * sig) * D.softplus(-phi / sig)
    wxn = _cross(om, nr)
    rel = [vb[k] - rb * wxn[k] - vh[k] for k in range(3)]       # ball's contact point (at -r_b n from its centre) against the hand
    vn = rel[0] * nr[0] + rel[1] * nr[1] + rel[2] * nr[2]
    vt = [rel[k] - vn * nr[k] for k in range(3)]
    fnn = fn0 * (1.0 - dn * vn)
    Fc = [fnn * nr[k] - (mu * fn0) * vt[k] for k in range(3)]    # on the ball; the hand receives -Fc
    nxv = _cross(nr, vt)
    tc = [(rb * mu) * fn0 * nxv[k] for k in range(3)]            # (-r_b n) x Fc
    # ball - ground contact
    fg0 = (kc * sig) * D.softplus(-(pb[2] - rb) / sig)
    vcx, vcy = vb[0] - rb * om[1], vb[1] + rb * om[0]
    Fg = [-(mu * fg0) * vcx, -(mu * fg0) * vcy, fg0 * (1.0 - dn * vb[2])]
    tg = [rb * Fg[1], -(rb * Fg[0]), 0.0]
    # joints
    qdn, qn = [None] * 7, [None] * 7
    for i in range(7):
        grav = g * (m_hd * J[i][2] + (m_el * JEz[i] if i < 3 else 0.0))
        jf = J[i][0] * Fc[0] + J[i][1] * Fc[1] + J[i][2] * Fc[2]
        acc = (u[i] - bj * qd[i] - grav - jf) / Ij[i]
        qdn[i] = qd[i] + dt * acc
        qn[i] = q[i] + dt * qdn[i]
    # ball
    ib = 1.0 / (0.4 * mb * rb * rb)
    omn = [om[k] + dt * ((tc[k] + tg[k]) * ib) for k in range(3)]
    al = [(Fc[0] + Fg[0]) / mb, (Fc[1] + Fg[1]) / mb, (Fc[2] + Fg[2]) / mb - g]
    vbn = [vb[k] + dt * al[k] for k in range(3)]
    pbn = [pb[k] + dt * vbn[k] for k in range(3)]
    hd = 0.5 * dt
    quatn = [qw + hd * (-(omn[0] * qx) - omn[1] * qy - omn[2] * qz),
             qx + hd * (qw * omn[0] + (omn[1] * qz - omn[2] * qy)),
             qy + hd * (qw * omn[1] + (omn[2] * qx - omn[0] * qz)),
             qz + hd * (qw * omn[2] + (omn[0] * qy - omn[1] * qx))]
    return qn + quatn + pbn + qdn + omn + vbn


# ARM27C: the arm + ball with COUPLED rigid-body joint dynamics (SURVEY (f)4: "higher-fidelity articulated-body" dynamics for the n = 27
# shape; kinova_gen3.py:105-213 builds the real arm from its URDF).  Same kinematics, state, contacts and integrator as ARM27; the
# seven joint accelerations now solve the manipulator equation
#     M(q) qdd = tau - b qd - J_hand^T F_contact - sum_p m_p J_p^T (a_p + g e_z),      M = diag(I_rotor) + sum_p m_p J_p^T J_p,
# over three point masses p (elbow, wrist, hand) that carry the links' inertia: J_p the points' Jacobians (3 / 5 / 7 columns),
# a_p their velocity-product accelerations (Jdot_p qd: centripetal and Coriolis terms, from the links' angular velocities and the
# bias part of their angular accelerations, recursively along the chain), M factorized as L D L^T (7 x 7, no pivoting: M is
# positive definite).  The same formulas in the same operation order in models.hpp (Arm27C) and ilqr_oracle.c.
def _dot3(a, b):
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def arm27c_kinematics(q, p):
    """arm27_kinematics + the wrist point (origin of joints 5, 6)."""
    ex, ey, ez = [1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]
    pos = [0.0, 0.0, A27_H0]
    axes, orgs = [], []
    elbow = wrist = None
    for i in range(7):
        s, c = D.sin(q[i]), D.cos(q[i])
        if i % 2 == 0:
            axes.append(list(ez)); orgs.append(list(pos))
            ex, ey = [c * ex[k] + s * ey[k] for k in range(3)], [c * ey[k] - s * ex[k] for k in range(3)]
        else:
            axes.append(list(ey)); orgs.append(list(pos))
            ex, ez = [c * ex[k] - s * ez[k] for k in range(3)], [c * ez[k] + s * ex[k] for k in range(3)]
        if i == 2:
            pos = [pos[k] + A27_L1 * ez[k] for k in range(3)]
            elbow = list(pos)
        elif i == 4:
            pos = [pos[k] + A27_L2 * ez[k] for k in range(3)]
            wrist = list(pos)
    hand = [pos[k] + (p[14] * ex[k] + A27_L3 * ez[k]) for k in range(3)]
    return hand, wrist, elbow, axes, orgs


def arm27c_step(x, u, p, dt):
    g, kc, sig, dn, mu, bj = p[0], p[1], p[2], p[3], p[4], p[5]
    mb, rb, re, m_el, m_hd, m_wr = p[6], p[7], p[8], p[9], p[10], p[15]
    Ia = [p[11], p[11], p[12], p[12], p[13], p[13], p[13]]
    q, qd = x[0:7], x[14:21]
    qw, qx, qy, qz = x[7], x[8], x[9], x[10]
    pb, om, vb = x[11:14], x[21:24], x[24:27]
    hand, wrist, elbow, axes, orgs = arm27c_kinematics(q, p)
    J = [_cross(axes[i], [hand[k] - orgs[i][k] for k in range(3)]) for i in range(7)]
    JW = [_cross(axes[i], [wrist[k] - orgs[i][k] for k in range(3)]) for i in range(5)]
    JE = [_cross(axes[i], [elbow[k] - orgs[i][k] for k in range(3)]) for i in range(3)]
    vh = [((J[0][k] * qd[0] + J[1][k] * qd[1]) + (J[2][k] * qd[2] + J[3][k] * qd[3])) + ((J[4][k] * qd[4] + J[5][k] * qd[5]) + J[6][k] * qd[6])
          for k in range(3)]
    # hand - ball and ball - ground contacts: ARM27's
    d = [pb[k] - hand[k] for k in range(3)]
    dist = D.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])
    idist = 1.0 / dist
    nr = [d[k] * idist for k in range(3)]
    phi = dist - (rb + re)
    fn0 = (kc * sig) * D.softplus(-phi / sig)
    wxn = _cross(om, nr)
    rel = [vb[k] - rb * wxn[k] - vh[k] for k in range(3)]
    vn = rel[0] * nr[0] + rel[1] * nr[1] + rel[2] * nr[2]
    vt = [rel[k] - vn * nr[k] for k in range(3)]
    fnn = fn0 * (1.0 - dn * vn)
    Fc = [fnn * nr[k] - (mu * fn0) * vt[k] for k in range(3)]
    nxv = _cross(nr, vt)
    tc = [(rb * mu) * fn0 * nxv[k] for k in range(3)]
    fg0 = (kc * sig) * D.softplus(-(pb[2] - rb) / sig)
    vcx, vcy = vb[0] - rb * om[1], vb[1] + rb * om[0]
    Fg = [-(mu * fg0) * vcx, -(mu * fg0) * vcy, fg0 * (1.0 - dn * vb[2])]
    tg = [rb * Fg[1], -(rb * Fg[0]), 0.0]
    # ---- velocity-product accelerations of the three point masses: angular velocity w and bias angular acceleration al of the
    #      links along the chain (w_i = w_{i-1} + a_i qd_i, al_i = al_{i-1} + (w_{i-1} x a_i) qd_i); the points sit on links 2, 4, 6
    w = [0.0, 0.0, 0.0]
    al = [0.0, 0.0, 0.0]
    wl, all_ = [], []
    for i in range(7):
        wxa = _cross(w, axes[i])
        al = [al[k] + wxa[k] * qd[i] for k in range(3)]
        w = [w[k] + axes[i][k] * qd[i] for k in range(3)]
        wl.append(w); all_.append(al)

    def rot_acc(al_, w_, r):
        a1, a2 = _cross(al_, r), _cross(w_, _cross(w_, r))
        return [a1[k] + a2[k] for k in range(3)]
    r2 = [elbow[k] - orgs[2][k] for k in range(3)]
    r4 = [wrist[k] - elbow[k] for k in range(3)]
    r6 = [hand[k] - wrist[k] for k in range(3)]
    aE = rot_acc(all_[2], wl[2], r2)
    t4 = rot_acc(all_[4], wl[4], r4)
    aW = [aE[k] + t4[k] for k in range(3)]
    t6 = rot_acc(all_[6], wl[6], r6)
    aH = [aW[k] + t6[k] for k in range(3)]
    gE, gW, gH = [aE[0], aE[1], aE[2] + g], [aW[0], aW[1], aW[2] + g], [aH[0], aH[1], aH[2] + g]
    # ---- right-hand side and mass matrix (lower triangle)
    rhs = []
    for i in range(7):
        h = m_hd * _dot3(J[i], gH)
        if i < 5:
            h = h + m_wr * _dot3(JW[i], gW)
        if i < 3:
            h = h + m_el * _dot3(JE[i], gE)
        rhs.append(u[i] - bj * qd[i] - h - _dot3(J[i], Fc))
    M = [[None] * 7 for _ in range(7)]
    for i in range(7):
        for j in range(i + 1):
            v = m_hd * _dot3(J[i], J[j])
            if i < 5:
                v = v + m_wr * _dot3(JW[i], JW[j])
            if i < 3:
                v = v + m_el * _dot3(JE[i], JE[j])
            if i == j:
                v = v + Ia[i]
            M[i][j] = v
    # ---- L D L^T and the two triangular solves
    L = [[None] * 7 for _ in range(7)]
    dd, idd = [None] * 7, [None] * 7
    for j in range(7):
        v = M[j][j]
        for k in range(j):
            v = v - (L[j][k] * L[j][k]) * dd[k]
        dd[j] = v
        idd[j] = 1.0 / v
        for i in range(j + 1, 7):
            v = M[i][j]
            for k in range(j):
                v = v - (L[i][k] * L[j][k]) * dd[k]
            L[i][j] = v * idd[j]
    y = [None] * 7
    for i in range(7):
        v = rhs[i]
        for k in range(i):
            v = v - L[i][k] * y[k]
        y[i] = v
    acc = [None] * 7
    for i in range(6, -1, -1):
        v = y[i] * idd[i]
        for k in range(i + 1, 7):
            v = v - L[k][i] * acc[k]
        acc[i] = v
    qdn = [qd[i] + dt * acc[i] for i in range(7)]
    qn = [q[i] + dt * qdn[i] for i in range(7)]
    # ---- ball: ARM27's
    ib = 1.0 / (0.4 * mb * rb * rb)
    omn = [om[k] + dt * ((tc[k] + tg[k]) * ib) for k in range(3)]
    alb = [(Fc[0] + Fg[0]) / mb, (Fc[1] + Fg[1]) / mb, (Fc[2] + Fg[2]) / mb - g]
    vbn = [vb[k] + dt * alb[k] for k in range(3)]
    pbn = [pb[k] + dt * vbn[k] for k in range(3)]
    hd = 0.5 * dt
    quatn = [qw + hd * (-(omn[0] * qx) - omn[1] * qy - omn[2] * qz),
             qx + hd * (qw * omn[0] + (omn[1] * qz - omn[2] * qy)),
             qy + hd * (qw * omn[1] + (omn[2] * qx - omn[0] * qz)),
             qz + hd * (qw * omn[2] + (omn[0] * qy - omn[1] * qx))]
    return qn + quatn + pbn + qdn + omn + vbn


def arm27c_gravity_torques(q, p):
    """Joint torques that hold the coupled arm still away from the ball: g(q) = g sum_p m_p J_p[:, z]."""
    hand, wrist, elbow, axes, orgs = arm27c_kinematics(list(q), p)
    out = []
    for i in range(7):
        v = p[10] * _cross(axes[i], [hand[k] - orgs[i][k] for k in range(3)])[2]
        if i < 5:
            v += p[15] * _cross(axes[i], [wrist[k] - orgs[i][k] for k in range(3)])[2]
        if i < 3:
            v += p[9] * _cross(axes[i], [elbow[k] - orgs[i][k] for k in range(3)])[2]
        out.append(p[0] * v)
    return np.array(out, dtype=float)


def arm27_gravity_torques(q, p):
    """Joint torques that hold the arm still away from the ball (kinova_gen3.py:268-275: the initial guess)."""
    hand, elbow, axes, orgs = arm27_kinematics(list(q), p)
    out = []
    for i in range(7):
        Jz = _cross(axes[i], [hand[k] - orgs[i][k] for k in range(3)])[2]
        JEz = _cross(axes[i], [elbow[k] - orgs[i][k] for k in range(3)])[2] if i < 3 else 0.0
        out.append(p[0] * (p[10] * Jz + p[9] * JEz))
    return np.array(out, dtype=float)


STEP_FUNCS = {PENDULUM: pendulum_step, ACROBOT: acrobot_step, CARTPOLE: cartpole_step,
              CARTPOLE_WALL: cartpole_wall_step, SYNTH36: synth36_step, PLANAR_QUAD: planar_quad_step, QUAD3D: quad3d_step,
              ARM27: arm27_step, ARM27C: arm27c_step}
INFEASIBLE_FUNCS = {PLANAR_QUAD: planar_quad_infeasible, QUAD3D: quad3d_infeasible}


class Model:
    """A discrete-time model x+ = f(x,u): id, parameter vector and time step."""

    def __init__(self, model_id, dt, params=None):
        self.model_id = int(model_id)
        self.n, self.m = MODEL_DIMS[self.model_id]
        self.dt = float(dt)
        self.params = np.array(DEFAULT_PARAMS[self.model_id] if params is None else params,
                               dtype=float)
        self._f = STEP_FUNCS[self.model_id]
        self._bad = INFEASIBLE_FUNCS.get(self.model_id)

    @classmethod
    def custom(cls, n, m, step_fn, params, dt, model_id=-1):
        """A model given as a float-or-Dual Python function step_fn(x, u, p, dt) -> list (the oracle side of a plugin
        model, drake_ddp_amd/plugin.py)."""
        s = cls.__new__(cls)
        s.model_id, s.n, s.m, s.dt = int(model_id), int(n), int(m), float(dt)
        s.params = np.array(params, dtype=float)
        s._f, s._bad = step_fn, None
        return s

    def step(self, x, u):
        """Next state for float inputs -> (n,) float array.  RuntimeError when the model declares the step
        infeasible (the reference's line search catches it: ilqr.py:315-323)."""
        xn = self._f(list(x), list(u), self.params, self.dt)
        if self._bad is not None and self._bad(xn, self.params):
            raise RuntimeError("infeasible simulation step")
        return np.array(xn, dtype=float)

    def step_unchecked(self, x, u):
        """The formula alone (finite differences probe it on both sides of a feasibility boundary)."""
        return np.array(self._f(list(x), list(u), self.params, self.dt), dtype=float)

    def step_generic(self, x, u):
        """Next state for float-or-Dual inputs -> list (the duck-typed Drake system's update)."""
        xn = self._f(list(x), list(u), self.params, self.dt)
        if self._bad is not None and self._bad(xn, self.params):
            raise RuntimeError("infeasible simulation step")
        return xn

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
            G[:, c] = (self.step_unchecked(xp, up) - self.step_unchecked(xm, um)) * inv2h
        return G[:, :n].copy(), G[:, n:].copy()
