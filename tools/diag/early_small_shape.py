"""A small plugin shape with forced clusters: early rounds on / off against one workgroup per problem - where do fx, fu, K differ?"""
import sys, os, subprocess, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
body = r'''
import sys, os, numpy as np
ROOT = %r
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "examples", "plugins"))
import models as PM
from drake_ddp_amd import plugin
from drake_ddp_amd.ilqr import BatchedIterativeLQR
nq, m, ne = 2, 1, 0
name, n, m_, bodyc, defaults, _ = PM.chainx_spec(nq, m, ne)
make = plugin.build_models([(name + "_large", n, m_, bodyc, defaults, "large")], verbose=False)
dt, B, N = 0.02, 2, int(os.environ.get("PROBE_N", "24"))
sys_ = make[name + "_large"](dt)
rng = np.random.default_rng(n * 17 + m)
x_nom = np.zeros(n); x0 = 0.4 * rng.standard_normal((B, n)); ug = 0.2 * rng.standard_normal((B, m, N - 1))
Q = dt * np.diag(10.0 ** rng.uniform(-1, 0.5, n)); R = dt * 0.05 * np.eye(m); Qf = np.diag(10.0 ** rng.uniform(0, 1, n))
s = BatchedIterativeLQR(sys_, N, B, delta=1e-3, beta=0.7, gamma=0.0, jacobian_mode="fd", max_iters=int(os.environ.get("PROBE_IT", "2")))
s.SetTargetState(x_nom); s.SetRunningCost(Q, R); s.SetTerminalCost(Qf); s.SetInitialState(x0); s.SetInitialGuess(ug)
try:
    s.Solve()
except RuntimeError:
    pass
np.savez(sys.argv[1], x=s.x_bar, u=s.u_bar, K=s.K, fx=s.fx, fu=s.fu, it=s.iterations, st=s.status, ls=s.ls_trials, cs=s.cluster_stats)
'''
outs = {}
for tag, env in (("single", {"MI_ILQR_CLUSTER": "1"}), ("early", {"MI_ILQR_CLUSTER": "4"}), ("early0", {"MI_ILQR_CLUSTER": "4", "MI_ILQR_EARLY": "0"})):
    f = "/tmp/ess_%s.npz" % tag
    r = subprocess.run([sys.executable, "-c", body % ROOT, f], capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
    assert r.returncode == 0, r.stderr[-1500:]
    outs[tag] = np.load(f)
print("iterations", {k: v["it"].tolist() for k, v in outs.items()}, "trials", {k: v["ls"].tolist() for k, v in outs.items()})
print("cluster stats early:", outs["early"]["cs"].tolist(), "| max|K| single %.2f early %.2f" % (np.abs(outs["single"]["K"]).max(), np.abs(outs["early"]["K"]).max()))
for tag in ("early", "early0"):
    for k in ("x", "u", "K", "fx", "fu"):
        d = np.abs(outs[tag][k] - outs["single"][k])
        per_t = d.reshape(d.shape[0], -1, d.shape[-1]).max(axis=1)       # (B, time)
        print(tag, k, "max diff %.2e" % d.max(), "| per time step (problem 1):", np.array2string(per_t[1], precision=1, max_line_width=250))
