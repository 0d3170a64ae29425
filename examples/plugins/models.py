"""Two models that are NOT compiled into libmi_ilqr.so, written the way a user of the open model interface writes them
(drake_ddp_amd/plugin.py, include/mi_ilqr.h: mi_ilqr_register_model): the C++ body of the discrete update for the device.
(The tests keep the same updates as float-or-dual Python functions for the oracle: tests/plugin_steps.py.)

  vdp     n = 2, m = 1   controlled Van der Pol oscillator, params [mu]
                         (an n = 2 model: served by the time-parallel rollout and Riccati scan, like the pendulum)
  kink2   n = 2, m = 1   a mass against a one-sided spring: non-smooth dynamics on the same kernels, params [c, k]
  chain3  n = 6, m = 2   three coupled pendula, the outer two actuated, params [ks, c, kc]
                         (a shape no built-in model has: served by the generic scalar passes of the wave-per-problem kernel)
  synth36p n = 36, m = 12 the built-in 36-state chain re-stated as a plugin (whole-step form) for the matrix-core family
  chainx   any (n, m)     nq coupled pendula with the last m actuated + ne first-order "filter" states (n = 2 nq + ne): the
                         mid-size workgroup-per-problem family (n <= 32, any m <= 16) at whatever shape a test asks for -
                         (12, 4) a quadrotor's, (14, 7) a 7-joint arm's, (27, 7) kinova_gen3.py's arm + free body
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

VDP_BODY = """    // q'' = mu (1 - q^2) q' - q + u, semi-implicit Euler
    const T q = x[0], v = x[1];
    const T a = p[0] * (1.0 - q * q) * v - q + u[0];
    const T vn = v + dt * a;
    xn[1] = vn; xn[0] = q + dt * vn;"""
VDP_DEFAULTS = [1.0]

# A pendulum-like mass with a one-sided spring (a wall at q = 0): the force has a KINK where the mass touches the wall - an n = 2
# model whose time-parallel rollout (ilqr_small.hpp: Newton on the trajectory) must notice a kink inside a lane's chunk of
# steps (its guard re-steps every step of the chunk for models the remainder was not measured on) and fall back.
KINK2_BODY = """    const double c = p[0], k = p[1];
    const T q = x[0], v = x[1];
    T a = u[0] - c * v - 2.0 * mi_sin(q);
    if (value_of(q) < 0.0) a = a - k * q;
    const T vn = v + dt * a;
    xn[1] = vn; xn[0] = q + dt * vn;"""
KINK2_DEFAULTS = [0.2, 400.0]

CHAIN3_BODY = """    const double ks = p[0], c = p[1], kc = p[2];
    const T l01 = mi_sin(x[1] - x[0]), l12 = mi_sin(x[2] - x[1]);
    const T a0 = -ks * mi_sin(x[0]) - c * x[3] + kc * l01 + u[0];
    const T a1 = -ks * mi_sin(x[1]) - c * x[4] + kc * l12 - kc * l01;
    const T a2 = -ks * mi_sin(x[2]) - c * x[5] - kc * l12 + u[1];
    const T v0 = x[3] + dt * a0, v1 = x[4] + dt * a1, v2 = x[5] + dt * a2;
    xn[3] = v0; xn[4] = v1; xn[5] = v2;
    xn[0] = x[0] + dt * v0; xn[1] = x[1] + dt * v1; xn[2] = x[2] + dt * v2;"""
CHAIN3_DEFAULTS = [4.0, 0.3, 3.0]


# The built-in synthetic 36-state chain (csrc/models.hpp: Synth36) written AGAIN as a plugin - whole-step form only, no
# per-dof hooks - to exercise the open interface on the workgroup-per-problem (matrix-core) family: the plugin must solve
# like the built-in model.
SYNTH36P_BODY = """    const double ks = p[0], c = p[1], kc = p[2], bu = p[3];
    constexpr int nq = 18;
    for (int i = 0; i < nq; ++i) {
      const T qi = x[i], vi = x[nq + i];
      T a = -ks * mi_sin(qi) - c * vi;
      if (i < nq - 1) a = a + kc * mi_sin(x[i + 1] - qi);
      if (i > 0) a = a - kc * mi_sin(qi - x[i - 1]);
      if (i >= 6) a = a + u[i - 6];
      else a = a + bu * (u[2 * i] - u[2 * i + 1]);
      const T vn = vi + dt * a;
      xn[nq + i] = vn; xn[i] = qi + dt * vn;
    }"""
SYNTH36P_DEFAULTS = [4.0, 0.5, 6.0, 0.1]


# A chain of nq coupled pendula with its last 12 actuated - the matrix-core family at other state dimensions than the two
# the library ships (n = 2 nq, e.g. 34 and 40: split tile layout with one resp. two four-row groups in the thin last row tile).
def chain_body(nq):
    return """    const double ks = p[0], c = p[1], kc = p[2];
    constexpr int nq = %d;
    for (int i = 0; i < nq; ++i) {
      const T qi = x[i], vi = x[nq + i];
      T a = -ks * mi_sin(qi) - c * vi;
      if (i < nq - 1) a = a + kc * mi_sin(x[i + 1] - qi);
      if (i > 0) a = a - kc * mi_sin(qi - x[i - 1]);
      if (i >= nq - 12) a = a + u[i - (nq - 12)];
      const T vn = vi + dt * a;
      xn[nq + i] = vn; xn[i] = qi + dt * vn;
    }""" % nq


CHAIN_DEFAULTS = [4.0, 0.5, 6.0]


# nq coupled pendula, the last m of them actuated, + ne first-order states e_j' = -a e_j + sin(q_{j mod nq}) + u_{j mod m}:
# n = 2 nq + ne, any m <= nq - the shapes of the mid-size family (params [ks, c, kc, a]).
def chainx_body(nq, m, ne):
    return """    const double ks = p[0], c = p[1], kc = p[2], ae = p[3];
    constexpr int nq = %d, mu = %d, ne = %d;
    for (int i = 0; i < nq; ++i) {
      const T qi = x[i], vi = x[nq + i];
      T a = -ks * mi_sin(qi) - c * vi;
      if (i < nq - 1) a = a + kc * mi_sin(x[i + 1] - qi);
      if (i > 0) a = a - kc * mi_sin(qi - x[i - 1]);
      if (i >= nq - mu) a = a + u[i - (nq - mu)];
      const T vn = vi + dt * a;
      xn[nq + i] = vn; xn[i] = qi + dt * vn;
    }
    for (int j = 0; j < ne; ++j) {
      const T e = x[2 * nq + j];
      xn[2 * nq + j] = e + dt * (mi_sin(x[j %% nq]) - ae * e + u[j %% mu]);
    }""" % (nq, m, ne)


CHAINX_DEFAULTS = [4.0, 0.5, 6.0, 2.0]


def chainx_spec(nq, m, ne=0):
    return ("chainx_%d_%d_%d" % (nq, m, ne), 2 * nq + ne, m, chainx_body(nq, m, ne), CHAINX_DEFAULTS, "large")


def build_chainx(nq, m, ne=0, verbose=False):
    from drake_ddp_amd import plugin
    return plugin.build_model(*chainx_spec(nq, m, ne), verbose=verbose)


# the shapes the tests and __graft_entry__.build() use: (nq, m, ne) -> (n, m) = (12, 4), (14, 7), (27, 7), (32, 16), (9, 4), (7, 3), (16, 1)
CHAINX_SHAPES = [(6, 4, 0), (7, 7, 0), (10, 7, 7), (16, 16, 0), (4, 4, 1), (3, 3, 1), (8, 1, 0)]
# (4, 1): a chain whose step is a few hundred cycles and whose whole state fits a CU's vector L1 several times over - the
# fast-step case of the cluster hand-shake's tests (tests/test_gpu_mpc_quadrupeds_boundary.py: chainx4)
FAST_STEP_SHAPES = [(2, 1, 0)]


# n > 32 with a number of controls that is not a multiple of 4: the plugin pads the device model's controls (plugin.device_controls),
# the classes of drake_ddp_amd/ilqr.py hide it: (n, m) = (36, 7) -> 8 device controls, (37, 3) -> 4
PADDED_SHAPES = [(18, 7, 0), (16, 3, 5)]
# 32 < n <= 40 with m in {4, 8, 16}: every layout class of the n > 32 backward pass other than the bench's (36, 12) - x's tail and
# u sharing a tile exactly ((40, 8)), a tile of its own for u because they would not fill one ((36, 4), (36, 8), (40, 4)) or
# because n % 4 != 0 ((35, 4), (38, 8), (39, 12)), and sixteen pivots ((33, 16), (36, 16), (40, 16))
LARGE_SHAPES = [(18, 4, 0), (18, 8, 0), (20, 4, 0), (20, 8, 0), (17, 4, 1), (19, 8, 0), (19, 12, 1), (16, 16, 1), (18, 16, 0), (20, 16, 0)]


def chain_spec(nq):
    return ("chain%d" % nq, 2 * nq, 12, chain_body(nq), CHAIN_DEFAULTS, "large")


def build_chain(nq, verbose=False):
    from drake_ddp_amd import plugin
    return plugin.build_model(*chain_spec(nq), verbose=verbose)


def build_all(verbose=False):
    """Compile the plugins (one hipcc each, in parallel: 10-40 s the first time) and return their ModelSystem factories."""
    from drake_ddp_amd import plugin
    specs = [("vdp", 2, 1, VDP_BODY, VDP_DEFAULTS, "small"), ("kink2", 2, 1, KINK2_BODY, KINK2_DEFAULTS, "small"),
             ("chain3", 6, 2, CHAIN3_BODY, CHAIN3_DEFAULTS, "small"),
             ("synth36p", 36, 12, SYNTH36P_BODY, SYNTH36P_DEFAULTS, "large"), chain_spec(17), chain_spec(20)]
    specs += [chainx_spec(*sh) for sh in CHAINX_SHAPES + PADDED_SHAPES + LARGE_SHAPES + FAST_STEP_SHAPES]
    return plugin.build_models(specs, verbose=verbose)


if __name__ == "__main__":
    import numpy as np
    from drake_ddp_amd.ilqr import BatchedIterativeLQR
    make = build_all(verbose=True)
    dt, N, B = 0.02, 100, 256
    s = BatchedIterativeLQR(make["vdp"](dt), N, B, delta=1e-3, beta=0.8)
    s.SetTargetState(np.zeros(2)); s.SetRunningCost(dt * np.eye(2), dt * 0.1 * np.eye(1)); s.SetTerminalCost(10.0 * np.eye(2))
    rng = np.random.default_rng(0)
    s.SetInitialState(rng.uniform(-2, 2, (B, 2))); s.SetInitialGuess(np.zeros((1, N - 1)))
    x, u, t, L = s.Solve()
    print(f"Van der Pol plugin model: {B} problems, {int(s.iterations.sum())} iLQR iterations, all converged: {bool((s.status == 0).all())}, "
          f"kernel {s.stats.kernel_ms:.3f} ms, |x_N| <= {np.abs(x[:, :, -1]).max():.3f}")
