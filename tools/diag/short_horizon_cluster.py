"""36-state chain at the shortest horizons, clusters forced against one workgroup per problem: where do they differ?"""
import sys, os, subprocess, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
body = r'''
import sys, os, numpy as np
ROOT = %r
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from drake_ddp_amd import workloads as W
from test_gpu_parity import make_solver
N, B = int(os.environ["PROBE_N"]), int(os.environ.get("PROBE_B", "70"))
p = dict(W.synth36_problem(), N=N)
x0 = W.synth36_batch_x0(B)
ug = 0.1 * np.random.default_rng(1).standard_normal((B, 12, N - 1))
s = make_solver(p, B=B, jac="fd", hist_cap=16)
s.SetInitialState(x0); s.SetInitialGuess(ug)
try:
    s.Solve()
except RuntimeError as e:
    print("raised", e)
np.savez(sys.argv[1], x=s.x_bar, u=s.u_bar, K=s.K, fx=s.fx, fu=s.fu, it=s.iterations, st=s.status, ls=s.ls_trials, L=s.cost, h=s.history, cs=s.cluster_stats)
'''
for N in (3, 4):
    outs = {}
    for tag, env in (("single", {"MI_ILQR_CLUSTER": "1"}), ("c2", {"MI_ILQR_CLUSTER": "2", "MI_ILQR_EARLY": "0", "MI_ILQR_LS_GROUPS": "0"}), ("c2early", {"MI_ILQR_CLUSTER": "2"})):
        f = "/tmp/shc_%s.npz" % tag
        r = subprocess.run([sys.executable, "-c", body % ROOT, f], capture_output=True, text=True, timeout=300, env=dict(os.environ, PROBE_N=str(N), **env))
        assert r.returncode == 0, r.stderr[-1500:]
        outs[tag] = np.load(f)
    a = outs["single"]
    for tag in ("c2", "c2early"):
        b = outs[tag]
        bad = np.nonzero(np.abs(a["L"] - b["L"]) > 1e-12 * np.abs(a["L"]))[0]
        print(f"N={N} {tag}: problems with another cost: {bad.tolist()[:20]} of {len(a['L'])}; iterations differ in {int((a['it'] != b['it']).sum())}, status {np.unique(b['st']).tolist()}, helpers {b['cs'][:, 0].tolist()[:12]}")
        for k in ("x", "u", "K", "fx", "fu"):
            d = np.abs(a[k] - b[k]).reshape(len(a["L"]), -1).max(axis=1)
            print(f"   {k}: max diff {d.max():.2e} (problem {int(d.argmax())}); problems off: {np.nonzero(d > 1e-13)[0].tolist()[:20]}")
        for q in bad[:3]:
            print(f"   problem {q}: single it {a['it'][q]} L {a['L'][q]:.12g} hist {a['h'][q, :a['it'][q], :3].tolist()} | {tag} it {b['it'][q]} L {b['L'][q]:.12g} hist {b['h'][q, :b['it'][q], :3].tolist()}")
