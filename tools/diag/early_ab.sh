# early linearization on / off x placement order, same box: bash tools/diag/early_ab.sh
OUT=gpurun_out/early; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_mpc_quadrupeds_boundary.py -q -x -k "cluster" > $OUT/cluster_tests.log 2>&1; tail -2 $OUT/cluster_tests.log
python - <<'PY'
import sys, numpy as np
sys.path.insert(0, "tests")
from drake_ddp_amd import workloads as W
from test_gpu_parity import make_solver
for name, p, x0, ug in (("arm27", W.arm27_problem(), W.arm27_batch_x0(64), W.arm27_u_guess(50)), ("synth36", W.synth36_problem(), W.synth36_batch_x0(8), W.synth36_u_guess(40)),
                        ("quad3d", W.quad3d_problem(), W.quad3d_batch_x0(64), W.quad3d_u_guess(40))):
    s = make_solver(p, B=len(x0), jac="fd"); s.SetInitialState(x0); s.SetInitialGuess(ug); s.Solve()
    cs = s.cluster_stats
    print(name, "iters", s.iterations.sum(), "trials", s.ls_trials.sum(), "| helpers", cs[:, 0].min(), cs[:, 0].max(), "regular rounds", cs[:, 1].sum(), "same-L2", cs[:, 2].sum(), "early opened", cs[:, 3].sum(), "early hit", cs[:, 4].sum(), flush=True)
PY
for early in 1 0; do
for ord in 0 2; do
echo "== MI_ILQR_EARLY=$early MI_ILQR_CLUSTER_ORDER=$ord"
export MI_ILQR_EARLY=$early MI_ILQR_CLUSTER_ORDER=$ord
python tools/cyc_large_models.py 2>/dev/null
MI_ILQR_CLUSTER=4 python tools/cyc_large_models.py 2>/dev/null | head -1
python tools/cyc_large_models.py 8 2>/dev/null | head -1
MI_CYC_ARMS_ONLY=1 python tools/cyc_mid_models.py 2>/dev/null | grep -v "B     1 \|B  1024\|B   256"
done; done
