"""Model descriptors: what the `system` argument of the reference class
(/root/reference/ilqr.py:21) becomes.  A Drake System cannot execute on the GPU
(SURVEY.md §8b), so the solver is handed a descriptor naming one of the device
dynamics functions (drake_ddp_amd/csrc/models.hpp) plus its parameters and time
step.  The descriptor answers the Drake calls the example scripts make on their
plant around the solver (IsDifferenceEquationSystem, time_step, ...)."""
import numpy as np

from . import _capi

PENDULUM, ACROBOT, CARTPOLE, CARTPOLE_WALL, SYNTH36, PLANAR_QUAD, QUAD3D, ARM27, ARM27C = 0, 1, 2, 3, 4, 5, 6, 7, 8
_DIMS = {PENDULUM: (2, 1), ACROBOT: (4, 1), CARTPOLE: (4, 1), CARTPOLE_WALL: (4, 1), SYNTH36: (36, 12), PLANAR_QUAD: (36, 12), QUAD3D: (37, 12),
         ARM27: (27, 7), ARM27C: (27, 7)}
_DEFAULTS = {
    PENDULUM: [0.25, 0.1, 4.905],
    ACROBOT: [1.0, 1.0, 1.0, 0.5, 1.0, 0.083, 0.33, 0.1, 0.1, 9.81],
    CARTPOLE: [10.0, 1.0, 0.5, 9.81],
    CARTPOLE_WALL: [10.0, 1.0, 0.5, 9.81, -0.45, 0.05, 2000.0, 0.01],
    SYNTH36: [4.0, 0.5, 6.0, 0.1],
    PLANAR_QUAD: [9.81, 4000.0, 0.004, 0.3, 0.15, 0.05, 0.02, 2.0, 60.0],
    QUAD3D: [9.81, 4000.0, 0.004, 0.3, 0.15, 0.3, 60.0, 9.0, 0.07, 0.26, 0.28, 0.06, 0.06, 0.04],
    ARM27: [9.81, 1500.0, 0.005, 0.5, 1.0, 0.5, 0.2, 0.1, 0.05, 1.0, 0.8, 0.6, 0.3, 0.1, 0.04],
    ARM27C: [9.81, 1500.0, 0.005, 0.5, 1.0, 0.5, 0.2, 0.1, 0.05, 1.0, 0.8, 0.3, 0.15, 0.05, 0.04, 0.6],
}
_NAMES = {"pendulum": PENDULUM, "acrobot": ACROBOT, "cart_pole": CARTPOLE,
          "cart_pole_with_wall": CARTPOLE_WALL, "synth36": SYNTH36, "planar_quadruped": PLANAR_QUAD,
          "quadruped_3d": QUAD3D, "arm_and_ball": ARM27, "arm_and_ball_coupled": ARM27C}


class _InputPort:
    def __init__(self, m):
        self._m = m

    def size(self):
        return self._m

    def get_index(self):
        return 0


class ModelSystem:
    """Discrete-time model x+ = f(x,u) evaluated on the device."""

    def __init__(self, model, dt, params=None):
        self.model_id = _NAMES[model] if isinstance(model, str) else int(model)
        if self.model_id not in _DIMS:
            raise ValueError(f"unknown model {model!r}")
        self.n, self.m = _DIMS[self.model_id]
        self.dt = float(dt)
        self.params = np.array(_DEFAULTS[self.model_id] if params is None else params, dtype=np.float64)
        if self.params.size > _capi.MAX_PARAMS:
            raise ValueError("too many model parameters")

    @classmethod
    def registered(cls, model_id, n, m, dt, params, m_user=0):
        """A model registered at run time (drake_ddp_amd/plugin.py, mi_ilqr_register_model).  m_user: the model's own number
        of controls when the device model carries padding controls (plugin.device_controls): `m` is what callers see,
        `m_dev` what the device arrays are sized with."""
        s = cls.__new__(cls)
        s.model_id, s.n, s.m, s.dt = int(model_id), int(n), int(m_user) if m_user else int(m), float(dt)
        s.m_dev = int(m)
        s.params = np.array(params, dtype=np.float64)
        if s.params.size > _capi.MAX_PARAMS:
            raise ValueError("too many model parameters")
        return s

    # --- the Drake calls made on the plant by the reference ctor / scripts ---
    def IsDifferenceEquationSystem(self):          # ilqr.py:37
        return (True, self.dt)

    def time_step(self):                            # ilqr.py:725
        return self.dt

    def GetSubsystemByName(self, name):             # ilqr.py:725
        return self

    def get_input_port(self, index=0):              # ilqr.py:43
        return _InputPort(self.m)

    def get_actuation_input_port(self):             # pendulum.py:76
        return _InputPort(self.m)

    def num_multibody_states(self):                 # mini_cheetah.py:183
        return self.n


def Pendulum(dt=1e-2, **kw):
    return ModelSystem(PENDULUM, dt, **kw)


def Acrobot(dt=0.004, **kw):
    return ModelSystem(ACROBOT, dt, **kw)


def CartPole(dt=1e-2, **kw):
    return ModelSystem(CARTPOLE, dt, **kw)


def CartPoleWithWall(dt=1e-2, **kw):
    return ModelSystem(CARTPOLE_WALL, dt, **kw)


def Synth36(dt=4e-3, **kw):
    return ModelSystem(SYNTH36, dt, **kw)


def PlanarQuadruped(dt=4e-3, **kw):
    """Planar floating-base quadruped with ground contact; the one model that can declare a step infeasible."""
    return ModelSystem(PLANAR_QUAD, dt, **kw)


def Quadruped3D(dt=4e-3, **kw):
    """3-D floating-base quadruped with the state layout of mini_cheetah.py:41-52 (unit quaternion, n = 37, m = 12) and
    compliant feet contact; can declare a step infeasible like the planar one."""
    return ModelSystem(QUAD3D, dt, **kw)


def ArmAndBall(dt=1e-2, **kw):
    """7-joint arm pushing a free ball: the state kinova_gen3.py:52-70 / panda_fr3.py stack (7 joint angles | the ball's unit
    quaternion and position | 13 velocities; n = 27, m = 7), served by the mid-size workgroup-per-problem kernels."""
    return ModelSystem(ARM27, dt, **kw)


def ArmAndBallCoupled(dt=1e-2, **kw):
    """The arm + ball with COUPLED rigid-body joint dynamics (joint-space mass matrix of three point masses + rotor inertias,
    centripetal / Coriolis and gravity terms; csrc/models.hpp: Arm27C) - same state, contacts and kernels as ArmAndBall."""
    return ModelSystem(ARM27C, dt, **kw)
