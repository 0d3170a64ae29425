"""ORACLE-side workload definitions (test infrastructure): the cost matrices, horizons and
line-search parameters of the five BASELINE.json configs, restated here from the reference's
example scripts so that the golden generator (oracle/gen_golden.py) does not depend on product
code for its inputs.  tests/test_oracle_vs_golden.py::test_oracle_and_product_workloads_agree
holds these equal to drake_ddp_amd/workloads.py, value for value.

Model ids follow include/mi_ilqr.h / oracle/models_np.py.
"""
import numpy as np

PENDULUM, ACROBOT, CARTPOLE, CARTPOLE_WALL, SYNTH36, PLANAR_QUAD, QUAD3D, ARM27, ARM27C = range(9)
SYNTH_TARGET_VEL = 1.0


def _problem(name, model_id, dt, N, x_nom, Q, R, Qf, beta):
    # every example keeps delta = 1e-2, gamma = 0 (pendulum.py:85-86, acrobot.py:118-120, ...)
    return dict(name=name, model_id=model_id, dt=dt, N=N, x_nom=np.asarray(x_nom, dtype=np.float64),
                Q=Q, R=R, Qf=Qf, delta=1e-2, beta=beta, gamma=0.0)


def pendulum_problem():
    """/root/reference/pendulum.py:18-19 (T=2.0, dt=1e-2), :29 target, :32-34 cost, :93-94 (dt*Q, dt*R, Qf)."""
    T, dt = 2.0, 1e-2
    Q = 0.01 * np.diag([0.0, 1.0])
    R = 0.01 * np.eye(1)
    return _problem("pendulum", PENDULUM, dt, int(T / dt), [np.pi, 0.0], dt * Q, dt * R, 100.0 * np.eye(2), 0.95)


def pendulum_batch_x0(B, seed=0):
    rng = np.random.default_rng(seed)
    theta = rng.uniform(-np.pi, np.pi, B)
    omega = rng.uniform(-1.0, 1.0, B)
    return np.column_stack([theta, omega])


def acrobot_problem(N=40):
    """/root/reference/acrobot.py:20 (dt=0.004), :40 target, :43-45 cost, :118-120 beta=0.5; N=40 per the config."""
    dt = 0.004
    Q = 0.01 * np.diag([0.0, 0.0, 1.0, 1.0])
    R = 0.01 * np.eye(1)
    return _problem("acrobot", ACROBOT, dt, N, [np.pi, 0.0, 0.0, 0.0], dt * Q, dt * R, 100.0 * np.eye(4), 0.5)


def acrobot_batch_x0(B, seed=1):
    return np.random.default_rng(seed).uniform(-0.1, 0.1, (B, 4))


def cartpole_problem(N=100):
    """/root/reference/cart_pole.py:21-22,44-46, beta=0.9 (:106-108)."""
    dt = 1e-2
    Q = np.diag([10.0, 10.0, 0.1, 0.1])
    R = 0.001 * np.eye(1)
    return _problem("cart_pole", CARTPOLE, dt, N, [0.0, np.pi, 0.0, 0.0], dt * Q, dt * R,
                    np.diag([100.0, 100.0, 10.0, 10.0]), 0.9)


def cartpole_wall_problem(N=200):
    """/root/reference/cart_pole_with_wall.py:23 (dt), :38 target, :41-43 cost, :148 beta=0.5; N=200 per the config."""
    dt = 1e-2
    Q = np.diag([0.1, 1.0, 0.01, 0.01])
    R = 0.001 * np.eye(1)
    return _problem("cart_pole_with_wall", CARTPOLE_WALL, dt, N, [0.0, np.pi, 0.0, 0.0], dt * Q, dt * R,
                    np.diag([200.0, 200.0, 10.0, 10.0]), 0.5)


def cartpole_wall_batch_x0(B, seed=2):
    x0 = np.zeros((B, 4))
    x0[:, 1] = np.pi + 0.5 + np.random.default_rng(seed).uniform(-0.2, 0.2, B)
    return x0


def synth36_problem(N=40):
    """C5 shape: diagonal weights patterned on /root/reference/mini_cheetah.py:60-69 (6 'base' + 12 'leg'
    dofs), dt=4e-3 (:23), beta=0.5 (:168-169), forward-velocity target (:55-57)."""
    dt = 4e-3
    q_base = np.array([3.0, 3.0, 3.0, 1.0, 1.0, 1.0])
    v_base = np.ones(6)
    q_leg = np.zeros(12)
    v_leg = np.full(12, 0.01)
    Q = np.diag(np.concatenate([q_base, q_leg, 0.01 * v_base, v_leg]))
    R = 0.01 * np.eye(12)
    Qf = np.diag(np.concatenate([5.0 * q_base, 0.1 + q_leg, v_base, v_leg]))
    x_nom = np.zeros(36)
    x_nom[0] = SYNTH_TARGET_VEL * N * dt
    x_nom[18] = SYNTH_TARGET_VEL
    return _problem("synth36", SYNTH36, dt, N, x_nom, dt * Q, dt * R, Qf, 0.5)


def synth36_batch_x0(B, seed=3):
    x0 = np.zeros((B, 36))
    x0[:, :18] = np.random.default_rng(seed).uniform(-0.1, 0.1, (B, 18))
    return x0


def synth36_u_guess(N):
    return np.full((12, N - 1), 0.05)


# ---- 3-D quadruped: floating base with a quaternion attitude + four 3-joint legs + feet contact (csrc/models.hpp: Quad3D), n=37 m=12
QUAD3D_TARGET_VEL = 0.5                             # (mini_cheetah.py:25 asks Drake's model for 1.0 m/s; this model tracks 0.5 and falls after ~1.1 s at 1.0)
_Q3_LEG = np.array([0.0, -0.8, 1.6])                # ab/ad, hip, knee (mini_cheetah.py:41-46)
_Q3_STAND_Z = 0.27711117215837355                   # feet 4.4 mm into the compliant ground: 4 f_n = m g
_Q3_U_STAND = np.array([1.368495, 0.221674, -3.087599, -1.368495, 0.221674, -3.087599,
                        1.368495, 0.221674, -3.087599, -1.368495, 0.221674, -3.087599])


def quad3d_stand():
    """Standing state (mini_cheetah.py:41-52: q0 = [1,0,0,0 | 0,0,z | 4 x (0,-0.8,1.6)], zero velocity)."""
    x = np.zeros(37)
    x[0] = 1.0
    x[6] = _Q3_STAND_Z
    x[7:19] = np.tile(_Q3_LEG, 4)
    return x


def quad3d_problem(N=40, target_vel=None):
    """mini_cheetah.py:54-69,168-173 on the build's 3-D quadruped: Q = diag([3,3,3,3,1,1,1 | 0 x 12 | 0.01 x 6 | 0.01 x 12]),
    R = 0.01 I, Qf = diag([5 x base | 0.1 x 12 | 1 x 6 | 0.01 x 12]), passed as dt*Q, dt*R, Qf; the target is the standing
    state moved forward by target_vel * T with base x velocity target_vel; dt = 4e-3, beta = 0.5, delta = 1e-2."""
    dt = 4e-3
    qb = np.ones(7)
    qb[0:4] += 2.0
    vb = np.ones(6)
    ql, vl = np.zeros(12), 0.01 * np.ones(12)
    Q = np.diag(np.hstack([qb, ql, 0.01 * vb, vl]))
    R = 0.01 * np.eye(12)
    Qf = np.diag(np.hstack([5 * qb, 0.1 + ql, vb, vl]))
    tv = QUAD3D_TARGET_VEL if target_vel is None else target_vel
    x_nom = quad3d_stand()
    x_nom[4] += tv * N * dt                         # base x position (mini_cheetah.py:56)
    x_nom[22] += tv                                 # base x velocity (:57)
    return dict(name="quadruped_3d", model_id=QUAD3D, dt=dt, N=N, x_nom=x_nom,
                Q=dt * Q, R=dt * R, Qf=Qf, delta=1e-2, beta=0.5, gamma=0.0)


def quad3d_batch_x0(B, seed=5):
    """Standing states with the attitude, height and joints perturbed (unit quaternions)."""
    rng = np.random.default_rng(seed)
    x0 = np.tile(quad3d_stand(), (B, 1))
    x0[:, 1:4] += rng.uniform(-0.02, 0.02, (B, 3))
    x0[:, 0:4] /= np.linalg.norm(x0[:, 0:4], axis=1, keepdims=True)
    x0[:, 6] += rng.uniform(0.0, 0.01, B)
    x0[:, 7:19] += rng.uniform(-0.05, 0.05, (B, 12))
    return x0


def quad3d_u_guess(N):
    """Constant standing torques (the u_stand of mini_cheetah.py:47-49,177)."""
    return np.repeat(_Q3_U_STAND[:, None], N - 1, axis=1)


# ---- 7-joint arm + free ball (csrc/models.hpp: Arm27; oracle/models_np.py: arm27_step), n=27 m=7: the shape of kinova_gen3.py / panda_fr3.py
ARM27_PUSH_TORQUE = 2.0                             # N m on the base yaw joint, added to the gravity compensation of the initial guess
_A27_Q_START = np.array([-0.3067, 0.8748, 0.0, 1.2788, 0.0, 0.6632, 0.0])      # hand 4 cm beside the ball, at its height
_A27_BALL_Z = 0.10603644421888478                   # radius 0.1 + 6 mm: the compliant ground carries the ball's weight
_A27_U_GRAV = np.array([0.0, -8.101374106804624, 0.0, -2.4099460650944824, 0.0, -0.37867887819099333, 0.0])


def arm27_start():
    """kinova_gen3.py:52-70: x0 = [q_start | ball quaternion, position | 13 zero velocities]; the ball rests on the ground at
    (0.6, 0, r), the hand beside it (the "side" scenario's push direction is +y)."""
    return np.concatenate([_A27_Q_START, [1.0, 0.0, 0.0, 0.0, 0.6, 0.0, _A27_BALL_Z], np.zeros(13)])


def arm27_problem(N=50):
    """kinova_gen3.py:31-32 (T = 0.5, dt = 1e-2 -> N = 50), :66-87 ("side": ball target 0.15 m along +y; Q = diag([0 x 7 |
    0,0,0,0,100,100,100 | 0.1 x 7 | 0.1 x 6]), R = 0.01 I, Qf = the same with 10 x the ball's velocity weights), passed as
    dt*Q, dt*R, Qf (:262-263); beta = 0.5, delta = 1e-3, gamma = 0 (:258-259)."""
    dt = 1e-2
    q_ball = np.array([0.0, 0.0, 0.0, 0.0, 100.0, 100.0, 100.0])
    Q = np.diag(np.hstack([np.zeros(7), q_ball, 0.1 * np.ones(7), 0.1 * np.ones(6)]))
    R = 0.01 * np.eye(7)
    Qf = np.diag(np.hstack([np.zeros(7), q_ball, 0.1 * np.ones(7), 10 * 0.1 * np.ones(6)]))
    x_nom = arm27_start()
    x_nom[12] += 0.15
    return dict(name="arm_and_ball", model_id=ARM27, dt=dt, N=N, x_nom=x_nom,
                Q=dt * Q, R=dt * R, Qf=Qf, delta=1e-3, beta=0.5, gamma=0.0)


def arm27_batch_x0(B, seed=6):
    """The start state with the ball moved on the ground and the arm's joints perturbed (rng seed 6)."""
    rng = np.random.default_rng(seed)
    x0 = np.tile(arm27_start(), (B, 1))
    x0[:, 11:13] += rng.uniform(-0.01, 0.01, (B, 2))
    x0[:, 0:7] += rng.uniform(-0.01, 0.01, (B, 7))
    return x0


def arm27_u_guess(N):
    """Gravity compensation at the start configuration (kinova_gen3.py:268-275) plus ARM27_PUSH_TORQUE on the base yaw joint:
    the build's point contact has no force (hence no gradient) at a distance, unlike the reference's hydroelastic bodies,
    and pure gravity compensation leaves iLQR in the local optimum that never touches the ball."""
    u = _A27_U_GRAV.copy()
    u[0] += ARM27_PUSH_TORQUE
    return np.repeat(u[:, None], N - 1, axis=1)


# ---- the same problem on the arm with coupled rigid-body joint dynamics (csrc/models.hpp: Arm27C; oracle/models_np.py: arm27c_step)
_A27C_U_GRAV = np.array([0.0, -11.521967177552286, 0.0, -3.933396455272099, 0.0, -0.37867887819099333, 0.0])   # g(q_start) of the three point masses


def arm27c_problem(N=50):
    """arm27_problem on model 8: kinova_gen3.py's horizon, cost, target, delta and beta."""
    return dict(arm27_problem(N), name="arm_and_ball_coupled", model_id=ARM27C)


def arm27c_u_guess(N):
    """Gravity compensation of the coupled arm at the start configuration (kinova_gen3.py:268-275) + the push torque of arm27_u_guess."""
    u = _A27C_U_GRAV.copy()
    u[0] += ARM27_PUSH_TORQUE
    return np.repeat(u[:, None], N - 1, axis=1)


# panda_fr3.py:20-58 switches between scenarios that differ in start pose, ball target and initial guess only.  "side" is the one
# above (kinova_gen3.py's); "forward": the hand 1 cm behind the ball on the -x side (inverse kinematics of the build's arm with
# the base yaw at 0), target 0.2 m along +x, gravity compensation at that pose + J^T (5 N along +x) as the guess.  ("lift"
# needs the reference's whole-arm hydroelastic wrap: a single point contact cannot carry the ball.)
_A27_Q_FORWARD = np.array([0.0, 0.52920677, 0.0, 2.20419733, 0.0, -0.1334041, 0.0])
_A27_U_GRAV_FORWARD = np.array([0.0, -5.53319785, 0.0, -1.78905756, 0.0, -0.82333227, 0.0])
_A27_JT_X_FORWARD = np.array([0.0, -0.17396357, 0.0, -0.53651085, 0.0, -0.25198002, 0.0])     # J^T e_x at that pose


def arm27_scenario(name, N=50):
    """(problem, x0, u_guess) of a panda_fr3.py scenario on the arm + ball model: "side" or "forward"."""
    prob = arm27_problem(N)
    if name == "side":
        return prob, arm27_start(), arm27_u_guess(N)
    if name != "forward":
        raise RuntimeError("Unknown scenario %s" % name)          # panda_fr3.py:47 ("lift": see above)
    x0 = np.concatenate([_A27_Q_FORWARD, [1.0, 0.0, 0.0, 0.0, 0.6, 0.0, _A27_BALL_Z], np.zeros(13)])
    x_nom = x0.copy()
    x_nom[11] += 0.2
    u = _A27_U_GRAV_FORWARD + 5.0 * _A27_JT_X_FORWARD
    return dict(prob, x_nom=x_nom), x0, np.repeat(u[:, None], N - 1, axis=1)


def mpc_shift(x, u, replan):
    """acrobot.py:147-152 / mini_cheetah.py:193-198: drop `replan` controls, repeat the last, restart at x[:, replan]."""
    u_next = np.concatenate([u[..., replan:], np.repeat(u[..., -1:], replan, axis=-1)], axis=-1)
    return np.array(x[..., replan]), u_next


# ---- planar quadruped (articulated body + ground contact, the model that can declare a step infeasible)
QUAD_TARGET_VEL = 0.5
QUAD_STANCE = (0.6, -1.2, 0.6)            # hip, knee, ankle of every leg
QUAD_TAIL = (-1.2, -0.2, -0.2)
QUAD_U_STAND = (0.1743, 1.7505, -0.0176, 0.1743, 1.7505, -0.0176, 0.2331, 1.7024, -0.0176, 0.2331, 1.7024, -0.0176)


def planar_quad_stand():
    """Standing state: feet 4.5 mm into the compliant ground (that carries the weight), zero velocity."""
    q = np.zeros(18)
    q[1] = 0.20 * np.cos(0.6) + 0.18 * np.cos(0.6) + 0.14 - 0.0045
    q[3:15] = np.tile(QUAD_STANCE, 4)
    q[15:18] = QUAD_TAIL
    return np.concatenate([q, np.zeros(18)])


def planar_quad_problem(N=40):
    """Weights patterned on /root/reference/mini_cheetah.py:60-69 (base pose 1, orientation +2, legs 0, base
    velocities 0.01, joint velocities 0.01; Qf = 5x / 0.1+ / 1x), passed as dt*Q, dt*R, Qf (:172-173); target =
    the stance moving forward at QUAD_TARGET_VEL (:55-57); dt = 4e-3 (:23), beta = 0.5, delta = 1e-2 (:168-169)."""
    dt = 4e-3
    q_base = np.array([1.0, 1.0, 3.0])
    v_base = np.ones(3)
    q_jnt = np.zeros(15)
    v_jnt = np.full(15, 0.01)
    Q = np.diag(np.concatenate([q_base, q_jnt, 0.01 * v_base, v_jnt]))
    R = 0.01 * np.eye(12)
    Qf = np.diag(np.concatenate([5.0 * q_base, 0.1 + q_jnt, v_base, v_jnt]))
    x_nom = planar_quad_stand()
    x_nom[0] = QUAD_TARGET_VEL * N * dt
    x_nom[18] = QUAD_TARGET_VEL
    return _problem("planar_quadruped", PLANAR_QUAD, dt, N, x_nom, dt * Q, dt * R, Qf, 0.5)


def planar_quad_batch_x0(B, seed=4):
    """The stance with every joint angle and the trunk's height / pitch moved a little (rng seed 4)."""
    rng = np.random.default_rng(seed)
    x0 = np.tile(planar_quad_stand(), (B, 1))
    x0[:, 1] += rng.uniform(0.0, 0.01, B)
    x0[:, 2] += rng.uniform(-0.03, 0.03, B)
    x0[:, 3:18] += rng.uniform(-0.05, 0.05, (B, 15))
    return x0


def planar_quad_u_guess(N):
    """Constant standing torques (the u_stand of mini_cheetah.py:47-49,177)."""
    return np.repeat(np.array(QUAD_U_STAND)[:, None], N - 1, axis=1)
