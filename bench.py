#!/usr/bin/env python
"""bench.py — iLQR iterations/s of the batched HIP solver on BASELINE.json's config C2
(pendulum swing-up n=2 m=1 N=200, B=1024 seeded initial states per GPU, fp64).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

With --gpus N > 1 and no launcher environment (WORLD_SIZE unset) the script launches its own N ranks
(one process per GPU, torch.distributed.run on 127.0.0.1, RCCL) and rank 0 prints the line.

A "step" is ONE batched Solve() from a cold start (fresh solver state, the resident
u_guess re-armed) of this rank's shard; inputs are resident in HBM before the timed
region.  Steps are enqueued back to back on the solver's stream (one solve runs at a time;
each keeps its own kernel events and statistics record) and collected per group of 32.
value = (sum over ranks and steps of iLQR iterations) / (max-over-ranks wall).
After the headline the same line carries `configs` (BASELINE.json's other configs C1, C3, C4, C5 at
their full sizes, sharded over the ranks: iterations/s, ms per solve, kernel ms, algorithmic GB/s and
roofline fraction, fp64 FLOP/s for C5) and `boundary_inclusive` (C2 through the class surface:
Solve() with the host copy-in and the x_bar/u_bar/cost copy-out).  Prints exactly one JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# N > 1: how long the library's own RCCL communicator may take to come up and pass its self-check before the run falls back
NATIVE_DEADLINE_S = float(os.environ.get("MI_BENCH_NATIVE_DEADLINE_S", "120"))

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def pmc_traffic_committed(batch):
    """Fallback: HBM bytes per launch from the committed rocprofv3 PMC passes of THIS command
    (profiles/rNN_pmc_c2.json)."""
    import glob
    if batch != 1024:
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_c2.json")))
    if not files:
        return None, None
    with open(files[-1]) as fh:
        d = json.load(fh)
    return d.get("hbm_bytes_per_launch_corrected"), "committed " + os.path.basename(files[-1])


PMC_GROUPS = {
    # separate passes: FETCH_SIZE and WRITE_SIZE do not fit one pass (MI355X_MICROARCH.md, rocprofv3 PMC slots)
    "fetch": ["FETCH_SIZE"],
    "write": ["WRITE_SIZE"],
    "fp64": ["SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_TRANS_F64"],
    "waves": ["GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU"],
}


def pmc_live(batch, groups=("fetch", "write", "fp64", "waves")):
    """Hardware counters of the solve kernel, measured now: one rocprofv3 --pmc pass per counter group (kernel
    trace only) over a short nested run of this script; per counter the average over the kernel's launches after
    the first.  {} on any failure."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {}
    vals = {}
    env = dict(os.environ, TMPDIR="/tmp", MI_BENCH_NESTED="1")
    for grp in groups:
        out = tempfile.mkdtemp(prefix="mi_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--pmc"] + PMC_GROUPS[grp] + ["--kernel-trace", "--output-format", "csv", "-d", out, "--",
                   sys.executable, os.path.abspath(__file__), "--steps", "3", "--warmup", "1",
                   "--batch", str(batch), "--no-cpu-baseline", "--no-configs"]
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                           timeout=180, check=True)
            rows = {}
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for r in csv.DictReader(fh):
                        if "ilqr_small_kernel" in r["Kernel_Name"]:
                            rows.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
            for ctr, v in rows.items():
                v.sort()
                if len(v) >= 2:
                    vals[ctr] = sum(x for _, x in v[1:]) / len(v[1:])          # skip the first (cold) launch
        except Exception:
            pass
        finally:
            shutil.rmtree(out, ignore_errors=True)
    return vals


def pmc_traffic_live(batch, vals=None):
    """HBM bytes per launch of the solve kernel from FETCH_SIZE / WRITE_SIZE (KB; FETCH_SIZE is doubled per
    MI355X_MICROARCH.md: gfx950 tallies wide coalesced reads at half their bytes - an upper bound for our
    8 B/lane loads).  None on any failure."""
    vals = pmc_live(batch, ("fetch", "write")) if vals is None else vals
    if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
        return None
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0


def pmc_traffic(batch, vals=None):
    if os.environ.get("MI_BENCH_NESTED"):
        return None, None
    live = pmc_traffic_live(batch, vals)
    if live is not None:
        return live, "live rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH x2 gfx950 correction"
    return pmc_traffic_committed(batch)


def committed_issue_counters():
    """profiles/rNN_pmc_issue.json of the latest round (tools/pmc_issue.py on an MI355X): per BASELINE config the fp64
    flops per iLQR iteration counted by the SQ instruction counters, and the wave-slot occupancy of its launches."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_issue.json")))
    for f in reversed(files):
        try:
            with open(f) as fh:
                d = json.load(fh)
            if d.get("configs"):
                return d["configs"], os.path.basename(f)
        except Exception:
            pass
    return {}, None


def cpu_baseline(prob, x0, sample, budget_s=8.0):
    """The oracle timed on this box's host cores on a bounded sample of the same workload:
    (i) oracle/ilqr_oracle.c (plain-C restatement, OpenMP over problems, all cores) — the
    number reported as cpu_baseline.value; (ii) oracle/ilqr_np.py (NumPy restatement, the
    reference's own implementation style, 1 thread) as `numpy_value`.  Reported beside the
    GPU number; never `value`."""
    from oracle import c_oracle, models_np as M
    from oracle.ilqr_np import OracleILQR
    N = prob["N"]
    model = M.Model(prob["model_id"], prob["dt"])
    cores = len(os.sched_getaffinity(0))
    native = True
    try:
        c_oracle.solve_batch(model, prob, x0[:8], None, want_arrays=False, native=True)     # build for this host + warm
    except Exception:
        native = False                                                                      # (no compiler here: the checker build)
        c_oracle.solve_batch(model, prob, x0[:8], None, want_arrays=False)
    # pick the thread count that serves this box best (1024 sub-millisecond tasks do not always
    # scale to every hardware thread); the count actually used is what `cores` reports
    best_nt, best_rate = 1, 0.0
    for nt in sorted({1, 8, 16, 32, 64, 128, cores}):
        if nt > cores:
            continue
        c_oracle.solve_batch(model, prob, x0, None, nthreads=nt, want_arrays=False, native=native)
        t_ = time.perf_counter()
        r_ = c_oracle.solve_batch(model, prob, x0, None, nthreads=nt, want_arrays=False, native=native)
        rate = r_["iters"].sum() / (time.perf_counter() - t_)
        if rate > best_rate:
            best_nt, best_rate = nt, rate
    cores = best_nt
    reps, iters, t0 = 0, 0, time.perf_counter()
    while True:
        r = c_oracle.solve_batch(model, prob, x0, None, nthreads=cores, want_arrays=False, native=native)
        iters += int(r["iters"].sum())
        reps += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    out = {"value": iters / dt, "unit": "iterations/s", "cores": int(r["threads"]), "kind": "port",
           "sample": f"{reps} repetitions of the full {len(x0)}-problem C2 batch ({iters} iterations, {dt:.1f} s), "
                     f"oracle/ilqr_oracle.c, OpenMP {int(r['threads'])} threads",
           "ms_per_solve": 1e3 * dt / reps,
           "build": ("gcc " + " ".join(c_oracle.NATIVE_FLAGS) + " (compiled on this host)") if native
                    else "oracle/Makefile: gcc -O2 -fopenmp -ffp-contract=off (prebuilt checker library)"}
    # NumPy restatement, single thread, a few problems
    it2, done, t1 = 0, 0, time.perf_counter()
    for b in range(min(sample, len(x0))):
        o = OracleILQR(model, N, prob["delta"], prob["beta"], prob["gamma"], jacobian="fd", fd_step=1e-5)
        o.set_problem(x0[b], prob["x_nom"], prob["Q"], prob["R"], prob["Qf"], np.zeros((1, N - 1)))
        it2 += len(o.solve()[3])
        done += 1
        if time.perf_counter() - t1 > budget_s:
            break
    out["numpy_value"] = it2 / (time.perf_counter() - t1)
    out["numpy_sample"] = f"first {done} problems, {it2} iterations, oracle/ilqr_np.py, 1 thread"
    return out


def measured_fp64_peak():
    """The dense fp64 rate of an MI355X as MEASURED by tools/ubench/fp64_peak.hip (every SIMD busy with independent
    v_fma_f64 chains / v_mfma_f64_16x16x4; kernel time from HIP events) and committed as profiles/rNN_fp64_peak.json - the
    guide has no fp64 figure.  Falls back to 16 FMA / clk / SIMD (tools/ubench/mfma_cu.hip) x 1024 SIMDs x 2.4 GHz."""
    import glob
    for f in reversed(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_fp64_peak.json")))):
        try:
            with open(f) as fh:
                d = json.load(fh)
            return float(d["fp64_peak_TFLOPs"]), "measured: profiles/" + os.path.basename(f) + " (tools/ubench/fp64_peak.hip)"
        except Exception:
            pass
    return 78.6, "16 FMA/clk/SIMD (tools/ubench/mfma_cu.hip) x 1024 SIMDs x 2.4 GHz; no profiles/r*_fp64_peak.json found"


FP64_PEAK_TFLOPS, FP64_PEAK_SOURCE = measured_fp64_peak()


def backward_flops_per_iteration(n, m, N):
    """fp64 flops of one backward pass (ilqr.py:651-667 per step): fx^T Vxx (2n^3), (.) fx (2n^3), fu^T Vxx (2n^2 m),
    Quu (2nm^2), Qux (2n^2 m), the m x m inverse (~2/3 m^3), K (2m^2 n), Qux^T K (2n^2 m), vectors (~6nm + 4n^2)."""
    per_step = 4.0 * n ** 3 + 6.0 * n * n * m + 2.0 * n * m * m + 2.0 * m * m * n + (2.0 / 3.0) * m ** 3 + 6.0 * n * m + 4.0 * n * n
    return per_step * (N - 1)


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks here (one process per GPU,
    torch.distributed.run, rendezvous on 127.0.0.1) and let rank 0 print the JSON line."""
    import socket
    import subprocess
    import torch
    backend = os.environ.get("MI_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if ndev < 1:
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the hot path)")
    if backend == "nccl" and ndev < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {ndev} GPU(s) visible on this node")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "8"))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


class Ranks:
    """The launch contract's rank plumbing: fence = barrier + device sync; reductions over ranks."""

    def __init__(self, world, rank, backend, dist, torch):
        self.world, self.rank, self.backend, self.dist, self.torch = world, rank, backend, dist, torch

    def fence(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def reduce(self, values, op):
        if self.world == 1:
            return [float(v) for v in values]
        t = self.torch.tensor([float(v) for v in values], dtype=self.torch.float64,
                              device="cuda" if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op={"max": self.dist.ReduceOp.MAX, "sum": self.dist.ReduceOp.SUM}[op])
        return [float(v) for v in t.cpu()]


def make_solver(prob, B, dev_index, **kw):
    from drake_ddp_amd.ilqr import BatchedIterativeLQR
    from drake_ddp_amd.models import ModelSystem
    s = BatchedIterativeLQR(ModelSystem(prob["model_id"], prob["dt"]), prob["N"], B, delta=prob["delta"], beta=prob["beta"],
                            gamma=prob["gamma"], jacobian_mode="fd", fd_step=1e-5, device=dev_index, hist_cap=8, **kw)
    s.SetTargetState(prob["x_nom"])
    s.SetRunningCost(prob["Q"], prob["R"])
    s.SetTerminalCost(prob["Qf"])
    return s


def run_config(rk, dev_index, name, prob, x0_all, u_guess, reps, mpc=None):
    """One of BASELINE.json's configs at its full (global) batch, sharded contiguously over the ranks.
    `mpc` = (resolves, replan, (index, step) of the moving target or None): the cold solve plus the whole
    receding-horizon loop on the device (mi_ilqr_mpc_run) is one repetition; otherwise a repetition is
    one cold-start batched solve.  Inputs are resident before the timed region; one host synchronization
    per repetition.  Returns the aggregated record (rank 0's is printed)."""
    from drake_ddp_amd.dist import shard_range
    n, m, N = x0_all.shape[1], prob["R"].shape[0], prob["N"]
    Bg = len(x0_all)
    lo, hi = shard_range(Bg, rk.rank, rk.world)
    B = hi - lo
    s = make_solver(prob, B, dev_index) if B > 0 else None
    step = None
    if mpc is not None and mpc[2] is not None:
        step = np.zeros(n)
        step[mpc[2][0]] = mpc[2][1]

    def once():
        it = kms = ab = 0.0
        conv = mx = 0
        if s is None:
            return it, kms, ab, conv, mx
        if mpc is not None:
            s.Reset()
            s.SetTargetState(np.array(prob["x_nom"], float))
            s.SetInitialState(x0_all[lo:hi])
            s.SetInitialGuess(u_guess)
            s._push_problem()
            st = s.solve_resident()
            it += st.total_iters; kms += st.kernel_ms; ab += st.algorithmic_bytes
            st = s.MPCRun(mpc[0], mpc[1], target_step=step)
        else:
            s.rearm(cold=True)
            st = s.solve_resident()
        it += st.total_iters; kms += st.kernel_ms; ab += st.algorithmic_bytes
        return it, kms, ab, st.n_converged, st.max_iters_seen

    if s is not None and mpc is None:
        s.SetInitialState(x0_all[lo:hi])
        s.SetInitialGuess(u_guess)
        s._push_problem()
    once()                                           # warm-up (module load, first-touch)
    rk.fence()
    t0 = time.perf_counter()
    it = kms = ab = 0.0
    for _ in range(reps):
        a_, b_, c_, conv, mx = once()
        it += a_; kms += b_; ab += c_
    rk.fence()
    wall = time.perf_counter() - t0
    wall, kms_max, mx = rk.reduce([wall, kms, mx], "max")
    it, ab, conv = rk.reduce([it, ab, conv], "sum")
    solves = reps * (1 + (mpc[0] if mpc is not None else 0))
    gbps = ab / (kms_max * 1e-3) / 1e9 / max(rk.world, 1) if kms_max > 0 else 0.0     # per-GPU average rate
    out = {"name": name, "batch": Bg, "batch_per_gpu": -(-Bg // rk.world),
           "n": n, "m": m, "N": N, "solves": solves, "iterations": it / reps, "iterations_per_s": it / wall,
           "ms_per_solve": 1e3 * wall / solves, "kernel_ms_per_solve": kms_max / solves,
           "max_iterations_per_problem": int(mx), "converged": int(conv),
           "algorithmic_GBps_per_gpu": gbps, "hbm_frac": gbps / HBM_PEAK_GBS}
    cc, src = committed_issue_counters()
    key = name.split()[0]
    if key in cc and kms_max > 0:
        # the binding roofline of these kernels: fp64 arithmetic issued / fp64 vector+matrix peak.  Flops per iteration
        # are the SQ counters' (profiles/, same kernels, same workload); iterations and kernel time are this run's.
        tf = cc[key]["fp64_flops_per_iteration"] * it / (kms_max * 1e-3) / 1e12 / max(rk.world, 1)
        out["roofline_compute"] = {"bound": "fp64", "achieved": tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_PEAK_TFLOPS,
                                   "fp64_flops_per_iteration": cc[key]["fp64_flops_per_iteration"],
                                   "wave_slots_occupied": None if "shard" in name else cc[key]["wave_slots_occupied"],
                                   "counters": "committed " + src, "peak_source": FP64_PEAK_SOURCE}
    elif kms_max > 0:
        # no silent omission (round 5 lost C5's entry that way): say in the line that this config has no counter record
        out["roofline_compute"] = {"error": "no entry for %r in %s (tools/pmc_issue.py writes it; its `missing` key says why)" % (key, src or "profiles/r*_pmc_issue.json")}
    if n >= 16:                                      # C5: the backward pass is matrix-core work
        tf = backward_flops_per_iteration(n, m, N) * it / (kms_max * 1e-3) / 1e12 / max(rk.world, 1)
        out["backward_fp64_TFLOPs_per_gpu"] = tf
        out["fp64_frac"] = tf / FP64_PEAK_TFLOPS
    del s
    return out


def all_configs(rk, dev_index):
    """C1, C3, C4, C5 of BASELINE.json (C2 is the headline), SURVEY.md 8(d) inputs."""
    from drake_ddp_amd import workloads as W
    out = []
    p = W.pendulum_problem()
    out.append(run_config(rk, dev_index, "C1 pendulum swing-up, single problem (pendulum.py literal)", p,
                          np.zeros((1, 2)), np.zeros((1, p["N"] - 1)), reps=20))
    a = W.acrobot_problem()
    out.append(run_config(rk, dev_index, "C3 acrobot MPC N=40: 1 + 50 receding-horizon re-solves x batch 512, device loop", a,
                          W.acrobot_batch_x0(512), np.zeros((1, a["N"] - 1)), reps=3, mpc=(50, 2, None)))
    c = W.cartpole_wall_problem()
    out.append(run_config(rk, dev_index, "C4 cart-pole with wall N=200, batch 256, central-FD Jacobians", c,
                          W.cartpole_wall_batch_x0(256), np.zeros((1, c["N"] - 1)), reps=5))
    q = W.synth36_problem()
    out.append(run_config(rk, dev_index, "C5 n=36 m=12 N=40 MPC: 1 + 100 re-solves x batch 64, moving target, device loop", q,
                          W.synth36_batch_x0(64), W.synth36_u_guess(q["N"]), reps=2,
                          mpc=(100, 4, (0, W.SYNTH_TARGET_VEL * q["dt"] * 4))))
    pq = W.planar_quad_problem()
    out.append(run_config(rk, dev_index, "C5q planar quadruped (articulated-body dynamics + ground contact) n=36 m=12 N=40 MPC: "
                          "1 + 100 re-solves x batch 64, moving target, device loop", pq,
                          W.planar_quad_batch_x0(64), W.planar_quad_u_guess(pq["N"]), reps=2,
                          mpc=(100, 4, (0, W.QUAD_TARGET_VEL * pq["dt"] * 4))))
    q3 = W.quad3d_problem()
    out.append(run_config(rk, dev_index, "C5q3d 3-D quadruped (quaternion floating base, feet contact) n=37 m=12 N=40 MPC: "
                          "1 + 100 re-solves x batch 64, moving target, device loop", q3,
                          W.quad3d_batch_x0(64), W.quad3d_u_guess(q3["N"]), reps=2,
                          mpc=(100, 4, (4, W.QUAD3D_TARGET_VEL * q3["dt"] * 4))))
    a27 = W.arm27_problem()
    out.append(run_config(rk, dev_index, "C6 arm + ball (7-joint arm pushing a free body: kinova_gen3.py's shape) n=27 m=7 N=50 MPC: "
                          "1 + 20 re-solves x batch 64, device loop, mid-size kernels", a27,
                          W.arm27_batch_x0(64), W.arm27_u_guess(a27["N"]), reps=2, mpc=(20, 5, None)))
    a27c = W.arm27c_problem()
    out.append(run_config(rk, dev_index, "C6b arm + ball, coupled joint dynamics (M(q) qdd = tau - ...: dense 7x7 mass matrix per step) n=27 m=7 N=50 MPC: "
                          "cold solve + 20 warm re-solves in one launch B=64 (fd)", a27c,
                          W.arm27_batch_x0(64), W.arm27c_u_guess(a27c["N"]), reps=2, mpc=(20, 5, None)))
    if rk.world == 1:
        out.append(throughput_entry(dev_index))
        out.append(run_config(rk, dev_index, "C5 shard of an 8-GPU run: batch 8 on this GPU", q,
                              W.synth36_batch_x0(64)[:8], W.synth36_u_guess(q["N"]), reps=2,
                              mpc=(100, 4, (0, W.SYNTH_TARGET_VEL * q["dt"] * 4))))
    return out


def throughput_entry(dev_index, B=262144, reps=5):
    """The lane-per-problem ("throughput") kernels - the one family that STREAMS its state through HBM (ilqr_batch.hpp): acrobot
    n = 4, m = 1, N = 40, B problems, cold-start batched solves from resident inputs.  `roofline.traffic`: HBM bytes per launch
    from the committed FETCH_SIZE / WRITE_SIZE passes of the same workload (tools/pmc_throughput.py; the counters calibrated on an
    8 B / lane stream: FETCH_SIZE counts half of the bytes read, WRITE_SIZE all of the bytes written)."""
    from drake_ddp_amd import workloads as W
    a = W.acrobot_problem()
    s = make_solver(a, B, dev_index, kernel_mode="throughput")
    s.SetInitialState(np.tile(W.acrobot_batch_x0(512), (B // 512, 1)))
    s.SetInitialGuess(np.zeros((1, a["N"] - 1)))
    s._push_problem()
    s.rearm(cold=True); s.solve_resident()
    t0 = time.perf_counter()
    it = kms = ab = 0.0
    for _ in range(reps):
        s.rearm(cold=True)
        st = s.solve_resident()
        it += st.total_iters; kms += st.kernel_ms; ab += st.algorithmic_bytes
    wall = time.perf_counter() - t0
    traffic = src = None
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_throughput.json")))
    if paths:
        path = paths[-1]                                       # the latest round's counter passes (tools/pmc_throughput.py)
        for r in json.load(open(path))["runs"]:
            if r["B"] == B and r["kp"] == "none":
                traffic, src = r["hbm_bytes_per_launch"], f"committed profiles/{os.path.basename(path)} (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE, calibrated on tools/ubench/stream8)"
    k_s = kms / reps * 1e-3
    achieved = ab / reps / k_s / 1e9
    out = {"name": f"throughput: acrobot N=40, batch {B}, lane-per-problem kernels (state streamed through HBM), cold-start solves",
           "batch": B, "batch_per_gpu": B, "n": 4, "m": 1, "N": a["N"], "solves": reps, "iterations": it / reps, "iterations_per_s": it / wall,
           "ms_per_solve": 1e3 * wall / reps, "kernel_ms_per_solve": kms / reps, "max_iterations_per_problem": int(st.max_iters_seen),
           "converged": int(st.n_converged), "algorithmic_GBps_per_gpu": achieved, "hbm_frac": achieved / HBM_PEAK_GBS,
           "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "traffic": traffic, "traffic_source": src, "kernel": "ilqr_batch_kernel<Acrobot,FD,KP=false>", "kernel_ms": kms / reps,
                        "hbm_traffic_frac": (traffic / k_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
                        "note": "algorithmic bytes exceed the real traffic: the linearization is fused into the backward sweep (fx, fu never written or re-read)"}}
    del s
    return out


def boundary_inclusive(prob, x0, dev_index, reps=5):
    """C2 through the class surface, host buffers in and out: SetInitialState/SetInitialGuess + Solve()
    (copy-in of x0 and u_guess, the solve, copy-out of x_bar, u_bar, cost) - the PCIe-inclusive rate.
    Never `value`.  Two ways of calling it: a guess per problem into freshly allocated pageable arrays (what a
    caller who knows nothing about the device does), and the reference's own call shape - ONE (m,N-1) guess for
    the batch - with the results read into the solver's page-locked buffers (pinned_results=True)."""
    B, N = len(x0), prob["N"]

    def run(pinned, ug):
        s = make_solver(prob, B, dev_index, pinned_results=pinned)
        # (median over the timed calls: the first pageable copies of a process pay one-off staging set-up in the runtime -
        #  3-6 ms on some hosts - that three warm-up calls do not always absorb)
        its, walls = [], []
        for r in range(reps + 3):
            t0 = time.perf_counter()
            s.Reset()
            s.SetInitialState(x0)
            s.SetInitialGuess(ug)
            x, u, _, L = s.Solve()
            t1 = time.perf_counter()
            if r >= 3:
                its.append(s.stats.total_iters)
                walls.append(t1 - t0)
        wall = float(np.median(walls))
        return {"iterations_per_s": float(np.mean(its)) / wall, "ms_per_solve": 1e3 * wall, "ms_per_solve_max": 1e3 * max(walls),
                "bytes_in": int(x0.nbytes + ug.nbytes), "bytes_out": int(x.nbytes + u.nbytes + L.nbytes)}

    out = {"workload": "C2 through Solve(): host x0 + u_guess in, x_bar + u_bar + cost out"}
    out.update(run(False, np.zeros((B, 1, N - 1))))
    out["pinned_results_shared_guess"] = run(True, np.zeros((1, N - 1)))
    out["default_call"] = "pinned_results=True is the constructor default of both classes since round 4; `pinned_results_shared_guess` is what a caller of the reference's own call shape gets"
    out["C1_single_problem_through_Solve"] = class_surface_single(dev_index)
    out["C3_through_Solve_and_MPCRun"] = class_surface_mpc(dev_index)
    return out


def class_surface_single(dev_index, reps=30):
    """C1 through the DROP-IN class, everything included: pendulum.py:85-100's call sequence on
    IterativeLinearQuadraticRegulator - setters, Solve() (copy-in, launch, iteration log and results out, the stopwatch
    attributes), Python overhead and all: the latency an MPC user of pendulum.py sees per solve."""
    from drake_ddp_amd import workloads as W
    from drake_ddp_amd.ilqr import IterativeLinearQuadraticRegulator
    from drake_ddp_amd.models import ModelSystem
    p = W.pendulum_problem()
    ilqr = IterativeLinearQuadraticRegulator(ModelSystem(p["model_id"], p["dt"]), p["N"], delta=p["delta"], beta=p["beta"], gamma=p["gamma"],
                                             verbose=False, device=dev_index)
    ilqr.SetTargetState(p["x_nom"]); ilqr.SetRunningCost(p["Q"], p["R"]); ilqr.SetTerminalCost(p["Qf"])
    u0 = np.zeros((1, p["N"] - 1))
    ts, it = [], 0
    for r in range(reps + 3):
        ilqr.Reset()
        t0 = time.perf_counter()
        ilqr.SetInitialState(np.zeros(2))
        ilqr.SetInitialGuess(u0)
        x, u, _, L = ilqr.Solve()
        if r >= 3:
            ts.append(time.perf_counter() - t0)
            it += int(ilqr.stats.total_iters)
    ts = np.array(ts)
    return {"workload": "pendulum.py literal: SetInitialState + SetInitialGuess + Solve() of IterativeLinearQuadraticRegulator, cold start",
            "ms_per_solve_median": 1e3 * float(np.median(ts)), "ms_per_solve_min": 1e3 * float(ts.min()), "kernel_ms": float(ilqr.stats.kernel_ms),
            "iterations_per_solve": it / reps, "cost": float(L)}


def class_surface_mpc(dev_index, reps=3):
    """C3 through the class surface: acrobot.py:131-162's loop on BatchedIterativeLQR - setters + Solve() for the first plan
    (host arrays in and out), then MPCRun(50, 2) and a read of x_bar / u_bar: what a caller of the MPC scripts pays."""
    from drake_ddp_amd import workloads as W
    a = W.acrobot_problem()
    B = 512
    x0 = W.acrobot_batch_x0(B)
    s = make_solver(a, B, dev_index)
    ts, it = [], 0
    for r in range(reps + 1):
        s.Reset()
        t0 = time.perf_counter()
        s.SetInitialState(x0)
        s.SetInitialGuess(np.zeros((1, a["N"] - 1)))
        x, u, _, L = s.Solve()
        n0 = int(s.stats.total_iters)
        st = s.MPCRun(50, 2)
        x, u = s.x_bar, s.u_bar
        if r >= 1:
            ts.append(time.perf_counter() - t0)
            it += n0 + int(st.total_iters)
    wall = float(np.sum(ts))
    return {"workload": "acrobot MPC B=512: SetInitialState + SetInitialGuess + Solve() + MPCRun(50, 2) + x_bar, u_bar out",
            "iterations_per_s": it / wall, "ms_per_solve": 1e3 * wall / (reps * 51), "ms_per_loop": 1e3 * wall / reps}


def concurrent_batches(prob, x0, dev_index, handles=(2, 4), groups=12, per_group=20):
    """Independent C2 batches in flight at once: `h` solver handles (each with its own stream and its own
    1024 problems), every one pipelining cold-start solves like the headline does.  A single batch of 1024 leaves
    half of the SIMD-time idle (its launch lasts as long as its slowest problem); a second, independent batch
    fills it - what a server with several clients sees.  Reported beside the headline, never `value`: the
    headline's step is ONE batch at a time."""
    B, N = len(x0), prob["N"]
    out = []
    for h in handles:
        ss = []
        for _ in range(h):
            s = make_solver(prob, B, dev_index)
            s.SetInitialState(x0)
            s.SetInitialGuess(np.zeros((1, N - 1)))
            s._push_problem()
            s.set_timing(0)
            ss.append(s)
        best = None
        for g in range(groups):
            t0 = time.perf_counter()
            for _ in range(per_group):
                for s in ss:
                    s.rearm(cold=True)
                    s.solve_resident_async()
            it = sum(st.total_iters for s in ss for st in s.collect(per_group))
            dt = time.perf_counter() - t0
            if g >= groups // 2:                 # the first groups bring the clock up
                best = max(best or 0.0, it / dt)
        out.append({"handles": h, "batch_per_handle": B, "iterations_per_s": best})
        del ss
    return {"workload": "C2, independent batches of 1024 on their own streams, pipelined cold-start solves", "runs": out}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="problems per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="headline only (skip C1/C3/C4/C5 and the boundary-inclusive run)")
    ap.add_argument("--cpu-sample", type=int, default=24)
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the hot path)")
    # MI_BENCH_BACKEND=gloo lets the N>1 code path be exercised on a single-GPU box (all ranks
    # share device 0); the driver's multi-GPU runs use the default: nccl (= RCCL over xGMI).
    backend = os.environ.get("MI_BENCH_BACKEND", "nccl")
    if backend == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py --gpus {world}: only {torch.cuda.device_count()} GPU(s) visible on this node")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=backend)
    rk = Ranks(world, rank, backend, dist, torch)

    from drake_ddp_amd import workloads as W

    prob = W.pendulum_problem()
    B = args.batch
    N = prob["N"]
    x0_all = W.pendulum_batch_x0(B * world, seed=0)       # global batch; rank r owns a contiguous block
    x0 = x0_all[rank * B:(rank + 1) * B]

    # Order of the run: the CPU baseline (host cores only), then every other config and the boundary-inclusive
    # figure, then the headline.
    cpu_base = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_base = cpu_baseline(prob, x0_all, min(args.cpu_sample, B))
    configs = boundary = concurrent = None
    if not args.no_configs and not os.environ.get("MI_BENCH_NESTED"):
        configs = all_configs(rk, dev_index)
        if rank == 0:
            boundary = boundary_inclusive(prob, x0, dev_index)
            if world == 1:
                concurrent = concurrent_batches(prob, x0, dev_index)
        rk.fence()

    s = s_weak = make_solver(prob, B, dev_index)
    s.SetInitialState(x0)
    s.SetInitialGuess(np.zeros((1, N - 1)))
    s._push_problem()                                      # inputs resident in HBM from here on

    pending = []
    # The path's one collective.  By default through the LIBRARY's own RCCL communicator (mi_ilqr_allreduce_min_start / _wait: the C
    # caller's path, torch.distributed only ships the 128-byte id) - after it has proved itself on this run's ranks: every rank
    # creates it, the communicator's own rank count (ncclCommCount) must be the world size and a reduction of rank-dependent
    # values must return their minimum, on EVERY rank (agreed through torch.distributed); anything else falls back to
    # torch.distributed's all_reduce(MIN) with the reason in config.collective_fallback_reason - a communicator that does not answer within
    # NATIVE_DEADLINE_S included.  MI_BENCH_NATIVE_RCCL=try attempts it under any torch backend, =1 insists (no fall-back: a
    # failure ends the run with the library's message), =0 skips the attempt.
    native, native_why, comm_ranks, native_stuck = None, None, (world if world > 1 else None), False
    want_native = os.environ.get("MI_BENCH_NATIVE_RCCL", "auto")
    if world > 1 and want_native == "1":                                 # (any torch backend: it only ships the communicator's id)
        from drake_ddp_amd.dist import NativeComm
        native = NativeComm.from_torch(dev_index)
        comm_ranks = native.count()[0]
    elif world > 1 and want_native != "0":
        if backend != "nccl" and want_native != "try":
            native_why = f"backend {backend}: the ranks share a device, and RCCL refuses two ranks on one device"
        else:
            # (the attempt runs in a worker thread with a deadline: a communicator that never comes up - a bootstrap that
            #  cannot reach a peer - must cost this run NATIVE_DEADLINE_S, not the whole bench; the main thread then agrees on
            #  the fall-back with the other ranks and the run ends through os._exit, the stuck thread still inside librccl)
            import threading
            from drake_ddp_amd.dist import NativeComm
            box = {"ok": 0.0, "err": f"no answer from librccl within {NATIVE_DEADLINE_S} s (communicator bootstrap or the self-check's reduction)"}

            def attempt():
                try:
                    if os.environ.get("MI_BENCH_NATIVE_TEST_STALL"):      # (the contract test's stand-in for a bootstrap that hangs)
                        time.sleep(3600)
                    torch.cuda.set_device(dev_index)                      # (the current device is per thread)
                    c_ = NativeComm(rank, world, dev_index, ident=ident)
                    n_, r_ = c_.count()
                    got = c_.allreduce_min([float(rank + 1), -float(rank), 7.0])
                    if n_ != world or r_ != rank or list(got) != [1.0, -float(world - 1), 7.0]:
                        box.update(ok=0.0, err=f"self-check: count {n_}, rank {r_}, min {list(got)}")
                    else:
                        box.update(ok=1.0, err="", comm=c_)
                except Exception as e:                                   # noqa: BLE001 - any failure means "use torch.distributed"
                    box.update(ok=0.0, err=f"{type(e).__name__}: {e}")
            # the id travels on the MAIN thread (torch.distributed collectives keep one order on every rank)
            ident = NativeComm.torch_exchange(NativeComm.unique_id() if rank == 0 else None)
            th = threading.Thread(target=attempt, daemon=True)
            th.start()
            th.join(NATIVE_DEADLINE_S)
            native_stuck = th.is_alive()
            ok, err = (0.0, box["err"]) if native_stuck else (box["ok"], box["err"])
            all_ok = -rk.reduce([-ok], "max")[0]                          # min over ranks
            if all_ok < 1.0:
                native, native_why = None, ("library communicator not usable on every rank" + (f" (this rank: {err})" if err else ""))
            else:
                native = box["comm"]
                comm_ranks = native.count()[0]
    RING = 32      # solves the library lets us keep in flight (per-launch events + statistics records)
    # HIP events ride on one launch in TIME_EVERY: a profiled dispatch serializes the pipelined stream by ~5 us
    # (tools/ubench/gap.hip); roofline.kernel_ms is the average over the timed launches of the timed region
    TIME_EVERY = 4
    s.set_timing(TIME_EVERY)

    def run_steps(count, s=None):
        """`count` cold-start solves of the whole (per-rank) batch, enqueued back to back on the handle's
        stream in groups of RING: launch latency overlaps the previous solve; every solve still runs in
        full and leaves its own statistics record.  With N > 1 ranks the path's one collective - the
        RCCL all-reduce(min) of each solve's best cost - is issued per group, one 8*RING-byte
        reduction of the group's best costs, asynchronously (it overlaps the next group) and completed
        inside the timed region."""
        from drake_ddp_amd.dist import allreduce_min_vec_async
        s = s if s is not None else s_weak
        out = []
        done = 0
        while done < count:
            k = min(RING, count - done)
            for _ in range(k):
                s.rearm(cold=True)
                s.solve_resident_async()
            grp = s.collect(k)
            if world > 1:
                if native is not None:
                    drain()                                 # (one reduction in flight per communicator)
                    pending.append(native.start([st.best_cost for st in grp]))
                else:
                    pending.append(allreduce_min_vec_async([st.best_cost for st in grp], dev_index))
            out += grp
            done += k
        return out

    def drain():
        while pending:
            pending.pop().wait()

    # The shader clock needs ~10 ms of continuous work to reach its sustained level (tools/warm_sweep.py: the first
    # 20-step group after an idle spell runs 4-5 % slower than the fourth); W warm-up steps of 0.16 ms do not get
    # there, so CLOCK_RAMP_STEPS more untimed steps precede them.  Reported in the line as `clock_ramp_steps`.
    CLOCK_RAMP_STEPS = 96
    # The same K steps from an IDLE clock first (no ramp, no warm-up beyond the module load above): `value_cold_clock`,
    # what a caller who solves one batch now and then sees.  The device idles for half a second before it.
    rk.fence()
    time.sleep(0.5)
    rk.fence()
    tc0 = time.perf_counter()
    cold_steps = run_steps(args.steps)
    drain()
    rk.fence()
    cold_elapsed = rk.reduce([time.perf_counter() - tc0], "max")[0]
    cold_iters = rk.reduce([sum(st.total_iters for st in cold_steps)], "sum")[0]
    run_steps(CLOCK_RAMP_STEPS)
    run_steps(args.warmup)
    drain()
    s.set_timing(TIME_EVERY)                               # (restarts the one-in-k count: the first timed step is timed)
    rk.fence()
    t0 = time.perf_counter()
    per_step = run_steps(args.steps)
    drain()
    rk.fence()
    elapsed = time.perf_counter() - t0
    iters = sum(st.total_iters for st in per_step)
    ls_trials = sum(st.total_ls_trials for st in per_step)
    timed = [st.kernel_ms for st in per_step if st.kernel_ms > 0.0]   # the launches that carried their events
    kernel_ms, n_timed = sum(timed), len(timed)
    alg_bytes = sum(st.algorithmic_bytes for st in per_step)
    last = per_step[-1]
    elapsed = rk.reduce([elapsed], "max")[0]
    iters_all = rk.reduce([iters], "sum")[0]
    onehot = [0.0] * world
    onehot[rank] = float(iters)
    iters_per_rank = rk.reduce(onehot, "sum")              # every rank's own iteration sum over the K timed steps
    del s, s_weak

    # N > 1: the STRONG-scaling figure beside the weak one - north_star's "a batch of 1024 ... at 1/2/4/8 GPUs": the single-GPU run's
    # own 1024 problems (the same draw), a contiguous shard per rank, the same K steps between the same fences.  A launch lasts
    # as long as its slowest problem at any batch size, so this is expected to stay near the single-GPU value: it is reported
    # so that a scaling record shows it, not because the path has anything to gain from it.
    strong = None
    if world > 1:
        from drake_ddp_amd.dist import shard_range
        lo, hi = shard_range(B, rank, world)
        ss = make_solver(prob, hi - lo, dev_index)
        ss.SetInitialState(W.pendulum_batch_x0(B, seed=0)[lo:hi])
        ss.SetInitialGuess(np.zeros((1, N - 1)))
        ss._push_problem()
        ss.set_timing(TIME_EVERY)
        run_steps(args.warmup + 8, ss)
        drain()
        rk.fence()
        ts0 = time.perf_counter()
        st_steps = run_steps(args.steps, ss)
        drain()
        rk.fence()
        s_el = rk.reduce([time.perf_counter() - ts0], "max")[0]
        s_it = sum(st.total_iters for st in st_steps)
        onehot = [0.0] * world
        onehot[rank] = float(s_it)
        s_per_rank = rk.reduce(onehot, "sum")
        strong = {"value": sum(s_per_rank) / s_el, "unit": "iterations/s", "global_batch": B, "batch_per_gpu": hi - lo if rank == 0 else None,
                  "ms_per_step": 1e3 * s_el / args.steps, "iterations_per_rank": s_per_rank,
                  "note": "C2's 1024 problems (the single-GPU run's own draw) sharded contiguously over the ranks; same K steps, same fences"}
        del ss

    if rank == 0:
        k_ms = kernel_ms / n_timed                          # avg launch duration of the dominant kernel (HIP events)
        bytes_per_launch = alg_bytes / args.steps
        achieved = bytes_per_launch / (k_ms * 1e-3) / 1e9
        ctr = pmc_live(B) if (world == 1 and not os.environ.get("MI_BENCH_NESTED")) else {}
        traffic, traffic_src = pmc_traffic(B, ctr) if world == 1 else pmc_traffic_committed(B)
        # the BINDING roofline: fp64 arithmetic issued (SQ instruction counters of this kernel, live when one GPU runs
        # the bench, else the committed pass) against the fp64 vector peak
        cc, cc_src = committed_issue_counters()
        if all(k in ctr for k in PMC_GROUPS["fp64"]):
            flops = 64.0 * (2 * ctr["SQ_INSTS_VALU_FMA_F64"] + ctr["SQ_INSTS_VALU_ADD_F64"] + ctr["SQ_INSTS_VALU_MUL_F64"] + ctr["SQ_INSTS_VALU_TRANS_F64"])
            flops_src = "live rocprofv3 --pmc SQ_INSTS_VALU_{FMA,ADD,MUL,TRANS}_F64 x 64 lanes (FMA = 2)"
        elif "C2" in cc and B == 1024:
            flops = cc["C2"]["fp64_flops_per_iteration"] * iters / args.steps
            flops_src = "committed " + cc_src + " (flops per iteration) x this run's iterations"
        else:
            flops, flops_src = None, None
        slots = slots_src = None
        if "GRBM_GUI_ACTIVE" in ctr and "SQ_WAVE_CYCLES" in ctr and ctr["GRBM_GUI_ACTIVE"] > 0:
            slots = 4.0 * ctr["SQ_WAVE_CYCLES"] / ((ctr["GRBM_GUI_ACTIVE"] / 8.0) * 1024.0)
            slots_src = "live: 4 x SQ_WAVE_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)"
        elif "C2" in cc and B == 1024:
            slots, slots_src = cc["C2"]["wave_slots_occupied"], "committed " + cc_src
        compute = None
        if flops is not None:
            tf = flops / (k_ms * 1e-3) / 1e12
            compute = {"bound": "fp64", "achieved": tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_PEAK_TFLOPS,
                       "peak_source": FP64_PEAK_SOURCE, "fp64_flops_per_launch": flops, "source": flops_src, "wave_slots_occupied": slots, "wave_slots_source": slots_src,
                       "valu_busy_of_resident_wave_time": (ctr["SQ_ACTIVE_INST_VALU"] / ctr["SQ_WAVE_CYCLES"]) if ("SQ_ACTIVE_INST_VALU" in ctr and ctr.get("SQ_WAVE_CYCLES")) else None,
                       "note": "what binds this kernel: ONE wave per SIMD (a resident wave spends ~6 cycles per VALU instruction: 4 in the VALU, the rest scalar / LDS / back edges it cannot hide - DESIGN section 8), and the launch lasts as long "
                               "as its slowest problem (12 iterations against a mean of 6), so about half of the wave slots idle"}
        out = {
            "metric": "iLQR iterations/sec (batch, whole node)",
            "value": iters_all / elapsed,
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "clock_ramp_steps": CLOCK_RAMP_STEPS,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "ms_per_solve": 1e3 * elapsed / args.steps,
            "us_per_solve_per_problem": 1e6 * elapsed / args.steps / B,
            "higher_is_better": True,
            "scaling": "weak",
            "value_strong": None if strong is None else strong["value"],
            "strong_scaling": strong,
            "iterations_per_rank": iters_per_rank,
            "communicator_ranks": comm_ranks,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "C2 pendulum swing-up n=2 m=1 N=200, batch=1024 random initial states per GPU "
                                   "(rng seed 0), fp64, central-FD Jacobians h=1e-5, cold-start Solve per step",
                       "batch_per_gpu": B, "global_batch": B * world, "N": N, "n": 2, "m": 1,
                       "parallelism": f"batch-shard x{world}",
                       "collective": None if world == 1 else ("librccl all-reduce(min) via mi_ilqr_allreduce_min_start (the library's own communicator; ncclCommCount = %d)" % comm_ranks
                                                              if native is not None else f"torch.distributed all_reduce(MIN), backend {backend}"),
                       "collective_fallback_reason": native_why},
            "value_cold_clock": cold_iters / cold_elapsed,
            "ms_per_step_cold_clock": 1e3 * cold_elapsed / args.steps,
            "iterations_per_step_rank0": iters / args.steps,
            "max_iterations_per_problem": int(last.max_iters_seen),
            "converged_rank0": int(last.n_converged),
            "ls_trials_per_step_rank0": ls_trials / args.steps,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "ilqr_small_kernel<Pendulum,FD,SOLVE>", "kernel_ms": k_ms,
                         "kernel_ms_source": f"HIP events carried by {n_timed} of the {args.steps} timed launches (one in {TIME_EVERY})",
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "hbm_traffic_frac": (traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         "limiter": "the instruction stream of one wave per SIMD, NOT HBM: `bound`/`frac` are SURVEY 8(d)'s algorithmic-bytes accounting; the state is LDS-resident, the real "
                                    "HBM traffic is `hbm_traffic_frac` of the peak and what binds is `roofline_compute` (fp64 issued / fp64 peak at one wave per SIMD)",
                         "note": "instruction-count-bound: one wave per problem, rollout and Riccati sweep as time-parallel scans; state is LDS-resident, the launch lasts as long as its slowest problem"},
        }
        out["roofline_compute"] = compute
        out["cpu_baseline"] = cpu_base
        out["order"] = ("cpu_baseline, configs, boundary_inclusive, then the headline: 0.5 s idle, K steps from the idle clock "
                        "(value_cold_clock), clock_ramp_steps untimed steps (the shader clock reaches its sustained level), the W "
                        "warm-up steps, the K timed steps (value)")
        out["configs"] = configs
        out["boundary_inclusive"] = boundary
        out["concurrent_batches"] = concurrent
        print(json.dumps(out), flush=True)
    if native_stuck:                                       # a thread of this process is still inside librccl: no orderly teardown
        rk.fence()
        sys.stdout.flush()
        os._exit(0)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
