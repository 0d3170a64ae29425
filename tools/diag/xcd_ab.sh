# same-box A/B for the XCD-aware cluster placement: bash tools/diag/xcd_ab.sh <libA> <libB> ...
OUT=gpurun_out/xcd; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_mpc_quadrupeds_boundary.py -q -x -k "cluster" > $OUT/cluster_tests.log 2>&1; tail -2 $OUT/cluster_tests.log
python - <<'PY'
import sys, numpy as np
sys.path.insert(0, "tests")
from drake_ddp_amd import workloads as W
from test_gpu_parity import make_solver
for name, p, x0, ug in (("arm27", W.arm27_problem(), W.arm27_batch_x0(64), W.arm27_u_guess(50)), ("synth36", W.synth36_problem(), W.synth36_batch_x0(8), W.synth36_u_guess(40))):
    s = make_solver(p, B=len(x0), jac="fd"); s.SetInitialState(x0); s.SetInitialGuess(ug); s.Solve()
    cs = s.cluster_stats
    print(name, "cluster stats: helpers", cs[:, 0].min(), cs[:, 0].max(), "rounds", cs[:, 1].sum(), "same-L2 rounds", cs[:, 2].sum(), flush=True)
PY
for lib in "$@"; do
for cl in 0 2 4; do
echo "== $lib MI_ILQR_CLUSTER=$cl"
MI_ILQR_CLUSTER=$cl MI_ILQR_LIB=$PWD/$lib python tools/cyc_large_models.py 2>/dev/null
done
MI_ILQR_LIB=$PWD/$lib python tools/cyc_large_models.py 8 2>/dev/null | head -1
MI_ILQR_LIB=$PWD/$lib MI_CYC_ARMS_ONLY=1 python tools/cyc_mid_models.py 2>/dev/null | grep -v "B     1 \|B  1024"
done
bash tools/diag/ab_libs.sh "$@" 2>/dev/null | grep -v "C1 \|C2 \|C3 \|C4 " | head -40
