import sys, os, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from drake_ddp_amd import workloads as W
from test_gpu_parity import make_solver
from common import load_golden, make_oracle, rel_err
from oracle import c_oracle, models_np as M

def c3():
    a = W.acrobot_problem(); B = 512
    x0 = W.acrobot_batch_x0(B)
    s = make_solver(a, B=B, jac="fd")
    s.SetInitialState(x0); s.SetInitialGuess(np.zeros((1, a["N"] - 1)))
    s.Solve()
    first_it = s.iterations.copy(); first_L = s.cost.copy()
    st = s.MPCRun(50, 2)
    log = s.mpc_log
    t0 = time.time()
    r = c_oracle.mpc_batch(M.Model(a["model_id"], a["dt"]), a, x0, np.zeros((1, a["N"] - 1)), 50, 2)
    print("C3 oracle s", time.time() - t0, "threads", r["threads"])
    print("C3 first iters equal", np.mean(first_it == r["first"][:, 1]), "first L rel", np.max(np.abs(first_L - r["first"][:, 0]) / np.abs(r["first"][:, 0])))
    same = (log[:, :, -1] == r["log"][:, :, -1])
    print("C3 resolve-iters equal frac", same.mean(), "problems fully equal", same.all(axis=1).mean())
    relL = np.abs(log[:, :, -2] - r["log"][:, :, -2]) / np.abs(r["log"][:, :, -2])
    print("C3 cost rel: max over same", relL[same].max(), "max overall", relL.max(), "median", np.median(relL))
    full = same.all(axis=1)
    print("C3 x0 log abs diff (fully-equal problems)", np.max(np.abs(log[full][:, :, :4] - r["log"][full][:, :, :4])))
    print("C3 final x diff", np.max(np.abs(s.x_bar[full] - r["x_bar"][full])), "K rel", rel_err(s.K[full], r["K"][full]))
    print("C3 status", np.bincount(s.status), "ls equal", np.mean(s.ls_trials == 0))
    nf = ~full
    if nf.any():
        print("C3 non-equal problems", nf.sum(), "their max cost rel", relL[nf].max(), "final-cost rel", np.max(np.abs(log[nf, -1, -2] - r["log"][nf, -1, -2]) / np.abs(r["log"][nf, -1, -2])))

def c5():
    q = W.synth36_problem(); B = 64
    x0 = W.synth36_batch_x0(B); ug = W.synth36_u_guess(q["N"])
    step = np.zeros(36); step[0] = W.SYNTH_TARGET_VEL * q["dt"] * 4
    s = make_solver(q, B=B, jac="fd")
    s.SetInitialState(x0); s.SetInitialGuess(ug)
    s.Solve()
    first_it = s.iterations.copy(); first_L = s.cost.copy()
    st = s.MPCRun(100, 4, target_step=step)
    log = s.mpc_log
    t0 = time.time()
    r = c_oracle.mpc_batch(M.Model(q["model_id"], q["dt"]), q, x0, ug, 100, 4, target_step=step)
    print("C5 oracle s", time.time() - t0, "threads", r["threads"])
    print("C5 first iters equal", np.mean(first_it == r["first"][:, 1]), "first L rel", np.max(np.abs(first_L - r["first"][:, 0]) / np.abs(r["first"][:, 0])))
    same = (log[:, :, -1] == r["log"][:, :, -1])
    print("C5 resolve-iters equal frac", same.mean(), "problems fully equal", same.all(axis=1).mean())
    relL = np.abs(log[:, :, -2] - r["log"][:, :, -2]) / np.abs(r["log"][:, :, -2])
    print("C5 cost rel: max over same", relL[same].max(), "max overall", relL.max(), "median", np.median(relL))
    full = same.all(axis=1)
    print("C5 x0 log abs diff", np.max(np.abs(log[full][:, :, :36] - r["log"][full][:, :, :36])))
    print("C5 final x diff", np.max(np.abs(s.x_bar[full] - r["x_bar"][full])), "K rel", rel_err(s.K[full], r["K"][full]))

def c4_sens():
    for name in ("cartpole_wall_c4_0", "cartpole_wall_c4_1"):
        g, prob = load_golden(name)
        s = make_solver(prob, jac="fd", single=True, hist_cap=256)
        s.SetInitialState(g["x0"]); s.SetInitialGuess(g["u_guess"])
        s.Solve()
        it = int(s.iterations[0]); h = s.history[0][:it]
        outs = []
        for k in range(3):
            o = make_oracle(prob, jacobian="fd", fd_step=1e-5)
            x0 = np.array(g["x0"], float)
            if k == 1: x0[1] = np.nextafter(x0[1], np.inf)
            if k == 2: x0[1] = np.nextafter(x0[1], -np.inf)
            o.set_problem(x0, prob["x_nom"], prob["Q"], prob["R"], prob["Qf"], g["u_guess"])
            xo, uo, Lo, hist = o.solve()
            outs.append(np.array(hist))
        A, Bp, Bm = outs
        print(name, "iters hip", it, "oracle", len(A), len(Bp), len(Bm), "golden(AD)", len(g["hist"]))
        k = min(it, len(A), len(Bp), len(Bm))
        for i in range(k):
            dh = abs(h[i, 0] - A[i, 0]) / abs(A[i, 0]); dp = abs(Bp[i, 0] - A[i, 0]) / abs(A[i, 0]); dm = abs(Bm[i, 0] - A[i, 0]) / abs(A[i, 0])
            print(f"  it {i:3d} ls hip/A/B+/B- {int(h[i,2])}/{int(A[i,2])}/{int(Bp[i,2])}/{int(Bm[i,2])}  dL hip {dh:.2e}  ulp+ {dp:.2e}  ulp- {dm:.2e}")
        print("  final L hip", s.cost[0], "A", A[-1, 0], "B+", Bp[-1, 0], "B-", Bm[-1, 0])

def lockstep_fd():
    for name in ("cartpole_wall_c4_0", "cartpole_wall_c4_1"):
        g, prob = load_golden(name)
        s = make_solver(prob, jac="fd")
        o = make_oracle(prob, jacobian="fd", fd_step=1e-5)
        n, m, N = g["x_bar"].shape[0], g["u_bar"].shape[0], prob["N"]
        st = dict(x_bar=np.zeros((n, N)), u_bar=np.array(g["u_guess"], float).reshape(m, N - 1),
                  K=np.zeros((m, n, N - 1)), kappa=np.zeros((m, N - 1)), dV_coeff=np.zeros(N - 1))
        L = np.inf
        s.SetInitialState(g["x0"][None])
        worst = dict(L=0, x=0, u=0, fx=0, fu=0, K=0, kap=0, dV=0); bad_ls = 0
        for it in range(40):
            s.set_state(**{k: v[None] for k, v in st.items()})
            Lg, eps, ls = s.stage_forward(L)
            s.stage_backward()
            o.set_problem(g["x0"], prob["x_nom"], prob["Q"], prob["R"], prob["Qf"], st["u_bar"])
            o.x_bar, o.K, o.kappa, o.dV = st["x_bar"].copy(), st["K"].copy(), st["kappa"].copy(), st["dV_coeff"].copy()
            Lo, eps_o, ls_o = o.forward(L)
            o.backward()
            if ls[0] != ls_o: bad_ls += 1; print("   ls differs at", it, ls[0], ls_o)
            worst["L"] = max(worst["L"], abs(Lg[0] - Lo) / abs(Lo))
            worst["x"] = max(worst["x"], rel_err(s.x_bar[0], o.x_bar)); worst["u"] = max(worst["u"], rel_err(s.u_bar[0], o.u_bar))
            worst["fx"] = max(worst["fx"], rel_err(s.fx[0], o.fx)); worst["fu"] = max(worst["fu"], rel_err(s.fu[0], o.fu))
            worst["K"] = max(worst["K"], rel_err(s.K[0], o.K)); worst["kap"] = max(worst["kap"], rel_err(s.kappa[0], o.kappa))
            worst["dV"] = max(worst["dV"], rel_err(s.dV_coeff[0], o.dV))
            st = dict(x_bar=s.x_bar[0], u_bar=s.u_bar[0], K=s.K[0], kappa=s.kappa[0], dV_coeff=s.dV_coeff[0])
            L = Lg[0]
        print(name, "lockstep FD worst", {k: f"{v:.2e}" for k, v in worst.items()}, "ls mismatches", bad_ls)

def nonmatching():
    # the three property tests that drop non-matching problems: how far off are the dropped ones?
    for seed in range(4):
        rng = np.random.default_rng(100 + seed)
        model_id = [0, 1, 2, 0][seed]; n = [2, 4, 4, 2][seed]
        N = int(rng.integers(20, 90)); B = int(rng.integers(40, 150)); dt = [0.02, 0.01, 0.02, 0.03][seed]
        prob = dict(model_id=model_id, dt=dt, N=N, x_nom=np.concatenate([[np.pi], np.zeros(n - 1)]) if model_id != 2 else np.array([0, np.pi, 0, 0.0]),
                    Q=dt * np.diag(rng.uniform(0.0, 2.0, n)), R=dt * np.diag(rng.uniform(0.05, 0.5, 1)),
                    Qf=np.diag(rng.uniform(1.0, 50.0, n)), delta=1e-3, beta=float(rng.choice([0.5, 0.7, 0.9])), gamma=float(rng.choice([0.0, 0.1])))
        x0 = rng.uniform(-1.0, 1.0, (B, n))
        if model_id == 2: x0[:, 1] += np.pi
        ug = rng.uniform(-0.5, 0.5, (B, 1, N - 1))
        s = make_solver(prob, B=B, jac="fd", hist_cap=8)
        s.SetInitialState(x0); s.SetInitialGuess(ug)
        x, u, _, L = s.Solve()
        r = c_oracle.solve_batch(M.Model(model_id, dt), prob, x0, ug)
        ok = (r["status"] == 0) & (s.status == 0)
        same = ok & (s.iterations == r["iters"]) & (s.ls_trials == r["ls"])
        rel = np.abs(L - r["cost"]) / np.abs(r["cost"])
        print("rand seed", seed, "B", B, "N", N, "ok", ok.mean(), "same", same.mean(), "status hip", np.bincount(s.status, minlength=3), "oracle", np.bincount(r["status"], minlength=3),
              "rel cost same max", rel[same].max() if same.any() else None, "non-same rel costs", np.sort(rel[~same])[-5:] if (~same).any() else None,
              "iters nonsame hip/or", list(zip(s.iterations[~same][:6], r["iters"][~same][:6])))
    for N in [4, 5, 64, 66, 130, 254, 257, 258, 300]:
        rng = np.random.default_rng(N); dt = 2.0 / 200
        prob = dict(model_id=0, dt=dt, N=N, x_nom=np.array([np.pi, 0.0]), Q=dt * 0.01 * np.diag([0.0, 1.0]), R=dt * 0.01 * np.eye(1), Qf=100.0 * np.eye(2), delta=1e-3, beta=0.8, gamma=0.0)
        B = 48
        x0 = np.stack([rng.uniform(-np.pi, np.pi, B), rng.uniform(-1, 1, B)], axis=1); ug = rng.uniform(-0.2, 0.2, (B, 1, N - 1))
        s = make_solver(prob, B=B, jac="fd"); s.SetInitialState(x0); s.SetInitialGuess(ug)
        x, u, _, L = s.Solve()
        r = c_oracle.solve_batch(M.Model(0, dt), prob, x0, ug)
        ok = (r["status"] == 0) & (s.status == 0)
        same = ok & (s.iterations == r["iters"]) & (s.ls_trials == r["ls"])
        rel = np.abs(L - r["cost"]) / np.abs(r["cost"])
        print("horizon", N, "ok", ok.mean(), "same", same.mean(), "nonsame rel", rel[~same], "status", np.bincount(s.status, minlength=3), np.bincount(r["status"], minlength=3))

for f in sys.argv[1:]:
    t0 = time.time()
    globals()[f]()
    print("==", f, "took", round(time.time() - t0, 1), "s", flush=True)
