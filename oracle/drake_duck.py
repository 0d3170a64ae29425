"""Duck-typed stand-ins for the Drake System/Context/Port objects that
/root/reference/ilqr.py calls (every call site listed in SURVEY.md §8b),
backed by the build-owned models of oracle/models_np.py.  Test infrastructure:
lets oracle/gen_golden.py run the UNMODIFIED reference solver in this container.
"""
import numpy as np


class _Vec:
    def __init__(self, n):
        self.data = [0.0] * n

    def size(self):
        return len(self.data)

    def value(self):                      # ilqr.py:229
        return np.array(self.data, dtype=float).reshape(-1, 1)

    def CopyToVector(self):               # ilqr.py:265
        out = np.empty(len(self.data), dtype=object)
        for i, q in enumerate(self.data):
            out[i] = q
        return out


class _State:
    def __init__(self, n):
        self.vec = _Vec(n)

    def get_vector(self):
        return self.vec


class _Context:
    def __init__(self, n, m):
        self.state = _State(n)
        self.u = [0.0] * m

    def get_discrete_state_vector(self):  # ilqr.py:57
        return self.state.vec

    def SetDiscreteState(self, x):        # ilqr.py:223,259
        self.state.vec.data = list(np.asarray(x).ravel())

    def get_discrete_state(self):         # ilqr.py:227,263
        return self.state


class _Port:
    def __init__(self, m):
        self.m = m

    def size(self):                       # ilqr.py:58
        return self.m

    def FixValue(self, context, u):       # ilqr.py:224,260
        context.u = list(np.asarray(u).ravel())


class _Plant:
    def __init__(self, dt):
        self.dt = dt

    def time_step(self):                  # ilqr.py:725
        return self.dt


class DuckSystem:
    def __init__(self, model):
        self.model = model

    def IsDifferenceEquationSystem(self):  # ilqr.py:37
        return (True, self.model.dt)

    def CreateDefaultContext(self):        # ilqr.py:42,47
        return _Context(self.model.n, self.model.m)

    def get_input_port(self, index):       # ilqr.py:43,48
        return _Port(self.model.m)

    def ToAutoDiffXd(self):                # ilqr.py:46
        return DuckSystem(self.model)

    def CalcForcedDiscreteVariableUpdate(self, context, state):  # ilqr.py:228,264
        state.vec.data = self.model.step_generic(context.state.vec.data, context.u)

    def GetSubsystemByName(self, name):    # ilqr.py:725
        return _Plant(self.model.dt)
