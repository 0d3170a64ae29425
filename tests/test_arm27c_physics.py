"""MI_MODEL_ARM27C's joint dynamics against an INDEPENDENT statement of the same physics (CPU, oracle only): the model writes the
manipulator equation with analytic point Jacobians, a recursion for the velocity-product accelerations and an LDL^T solve
(oracle/models_np.py: arm27c_step; the formulas csrc/models.hpp: Arm27C and oracle/ilqr_oracle.c repeat operation by
operation); here the Euler-Lagrange equations of  T = 1/2 sum_p m_p |d/dt r_p(q)|^2 + 1/2 sum_i I_i qd_i^2,  V = g sum_p m_p z_p(q)
are formed from NUMERICAL derivatives of nothing but the three points' positions, and solved with numpy."""
import numpy as np
import pytest

from oracle import models_np as M

P = np.array(M.DEFAULT_PARAMS[M.ARM27C], float)
MASS = {"hand": P[10], "wrist": P[15], "elbow": P[9]}
ROTOR = np.array([P[11], P[11], P[12], P[12], P[13], P[13], P[13]])


def _points(q):
    hand, wrist, elbow, _, _ = M.arm27c_kinematics(list(q), P)
    return {"hand": np.array(hand), "wrist": np.array(wrist), "elbow": np.array(elbow)}


def _jac(q, h=1e-5):
    out = {k: np.zeros((3, 7)) for k in MASS}
    for i in range(7):
        e = np.zeros(7); e[i] = h
        a, b = _points(q + e), _points(q - e)
        for k in MASS:
            out[k][:, i] = (a[k] - b[k]) / (2 * h)
    return out


def _mass_matrix(q):
    J = _jac(q)
    return np.diag(ROTOR) + sum(MASS[k] * J[k].T @ J[k] for k in MASS)


def _kinetic(q, qd):
    return 0.5 * qd @ _mass_matrix(q) @ qd


def _potential(q):
    pts = _points(q)
    return P[0] * sum(MASS[k] * pts[k][2] for k in MASS)


def _euler_lagrange_qdd(q, qd, tau, h=1e-4):
    """M qdd = tau - d/dt(M) qd + dT/dq - dV/dq, every derivative a central difference."""
    Mq = _mass_matrix(q)
    Mdot_qd, dT, dV = np.zeros(7), np.zeros(7), np.zeros(7)
    for k in range(7):
        e = np.zeros(7); e[k] = h
        Mdot_qd += ((_mass_matrix(q + e) - _mass_matrix(q - e)) / (2 * h)) @ qd * qd[k]
        dT[k] = (_kinetic(q + e, qd) - _kinetic(q - e, qd)) / (2 * h)
        dV[k] = (_potential(q + e) - _potential(q - e)) / (2 * h)
    return np.linalg.solve(Mq, tau - Mdot_qd + dT - dV)


def _state(q, qd, ball=(5.0, 5.0, 5.0)):
    """Ball far from the hand and off the ground: no contact force reaches the arm."""
    x = np.zeros(27)
    x[0:7], x[7], x[11:14], x[14:21] = q, 1.0, ball, qd
    return x


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_joint_accelerations_are_the_euler_lagrange_ones(seed):
    rng = np.random.default_rng(seed)
    q, qd, tau = rng.uniform(-1.5, 1.5, 7), rng.uniform(-2, 2, 7), rng.uniform(-5, 5, 7)
    p = P.copy(); p[5] = 0.3                                             # joint damping enters as -b qd
    dt = 1e-3
    xn = np.array(M.arm27c_step(list(_state(q, qd)), list(tau), p, dt))
    qdd_model = (xn[14:21] - qd) / dt
    qdd_el = _euler_lagrange_qdd(q, qd, tau - p[5] * qd)
    err = np.max(np.abs(qdd_model - qdd_el)) / max(1.0, np.max(np.abs(qdd_el)))
    assert err < 1e-6, (err, qdd_model, qdd_el)
    assert np.allclose(xn[0:7], q + dt * xn[14:21], rtol=0, atol=1e-15)  # semi-implicit Euler, like the other models


def test_mass_matrix_couples_the_joints():
    """What ARM27 does not have: off-diagonal inertia (the response of joint j's acceleration to joint i's torque)."""
    q = np.array([0.2, 0.9, -0.3, 1.2, 0.1, 0.6, 0.0])
    base = np.array(M.arm27c_step(list(_state(q, np.zeros(7))), [0.0] * 7, P, 1e-3))[14:21]
    resp = np.zeros((7, 7))
    for i in range(7):
        tau = np.zeros(7); tau[i] = 1.0
        resp[:, i] = (np.array(M.arm27c_step(list(_state(q, np.zeros(7))), list(tau), P, 1e-3))[14:21] - base) / 1e-3
    Minv = np.linalg.inv(_mass_matrix(q))
    assert np.max(np.abs(resp - Minv)) < 1e-6 * np.max(np.abs(Minv))
    off = np.abs(Minv - np.diag(np.diag(Minv)))
    assert off.max() > 0.1 * np.abs(np.diag(Minv)).min()


def test_gravity_compensation_holds_the_arm_still():
    q = np.array([-0.3067, 0.8748, 0.0, 1.2788, 0.0, 0.6632, 0.0])       # kinova_gen3.py's start configuration
    tau = np.array(M.arm27c_gravity_torques(q, P))
    xn = np.array(M.arm27c_step(list(_state(q, np.zeros(7))), list(tau), P, 1e-3))
    assert np.max(np.abs(xn[14:21])) < 1e-14 and np.max(np.abs(xn[0:7] - q)) < 1e-14


def test_free_arm_keeps_its_energy():
    """No damping, no torque, no contact: T + V of the arm over 0.2 s of swinging (2000 steps of 1e-4 s; the semi-implicit Euler
    step is symplectic-like, its energy error oscillates at O(dt))."""
    p = P.copy(); p[5] = 0.0
    q, qd = np.array([0.2, 0.9, -0.3, 1.2, 0.1, 0.6, 0.0]), np.zeros(7)
    x = _state(q, qd, ball=(5.0, 5.0, 50.0))
    e0 = _kinetic(q, qd) + _potential(q)
    scale = abs(_potential(q) - _potential(np.zeros(7)))
    worst = 0.0
    for k in range(2000):
        x = np.array(M.arm27c_step(list(x), [0.0] * 7, p, 1e-4))
        if k % 200 == 199:
            worst = max(worst, abs(_kinetic(x[0:7], x[14:21]) + _potential(x[0:7]) - e0))
    assert np.max(np.abs(x[14:21])) > 0.5                                 # (it did swing)
    assert worst < 2e-3 * scale, (worst, scale)
