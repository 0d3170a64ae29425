"""bench.py's launch contract: `--gpus N` starts N ranks by itself when no launcher did, the N > 1
path runs (gloo stand-in for RCCL on a single-GPU box: all ranks share device 0), and the one JSON
line carries the headline, `roofline`, `cpu_baseline`, `configs` and `boundary_inclusive`."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=900):
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout,
                          env=dict(os.environ, **(env or {})), cwd=ROOT)


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def test_bench_fails_loudly_without_a_gpu_or_with_too_few():
    """No CPU fallback: without a device (this container) every form of the command exits non-zero with a
    message; on a GPU box asking for more GPUs than the node has does."""
    import torch
    if torch.cuda.is_available():
        r = _run(["--gpus", str(torch.cuda.device_count() + 1), "--steps", "1", "--warmup", "0"])
        assert r.returncode != 0 and "GPU(s) visible" in (r.stderr + r.stdout)
    else:
        for a in (["--steps", "1"], ["--gpus", "2", "--steps", "1"]):
            r = _run(a)
            assert r.returncode != 0 and "needs an MI355X" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_bench_gpus_2_launches_its_own_ranks():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-configs"],
             env={"MI_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 2048 and d["value"] > 0
    one = _json_line(_run(["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-configs"]).stdout)
    assert one["n_gpus"] == 1 and one["config"]["global_batch"] == 1024
    # weak scaling: every rank solves 1024 problems per step (its block of the 2048-problem global draw), so
    # rank 0's iterations per step stay within a few percent of the single-rank batch's
    assert abs(d["iterations_per_step_rank0"] - one["iterations_per_step_rank0"]) < 0.05 * one["iterations_per_step_rank0"]
    assert d["converged_rank0"] == 1024


@pytest.mark.gpu
def test_bench_line_carries_every_config_and_the_boundary():
    r = _run(["--steps", "5", "--warmup", "2", "--cpu-sample", "4"])
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "configs", "boundary_inclusive"):
        assert k in d, k
    assert d["dtype"] == "f64" and d["vs_baseline"] is None and "workload" in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["kernel_ms"] > 0
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1 and d["cpu_baseline"]["value"] > 0
    names = [c["name"][:2] for c in d["configs"]]
    assert names[:5] == ["C1", "C3", "C4", "C5", "C5"]
    for c in d["configs"]:
        assert c["iterations_per_s"] > 0 and c["ms_per_solve"] > 0 and c["kernel_ms_per_solve"] > 0 and 0 < c["hbm_frac"] < 1
        assert c["converged"] == c["batch"] or c["name"].startswith("C4")       # (C4: long contact solves may hit no cap, all converge too)
    assert d["configs"][3]["backward_fp64_TFLOPs_per_gpu"] > 0
    b = d["boundary_inclusive"]
    assert b["ms_per_solve"] > d["ms_per_step"] and b["bytes_out"] > b["bytes_in"]
    assert b["pinned_results_shared_guess"]["ms_per_solve"] > d["ms_per_step"] and b["pinned_results_shared_guess"]["bytes_in"] < b["bytes_in"]
    assert d["clock_ramp_steps"] >= 0 and "order" in d and "kernel_ms_source" in rf
    cb = d["concurrent_batches"]["runs"]
    assert [r_["handles"] for r_ in cb] == [2, 4] and all(r_["iterations_per_s"] > 0.8 * d["value"] for r_ in cb)


_ONE_RANK = {}


def _one_rank_configs():
    """The single-rank run with every config (cached for the sharded runs to be compared with)."""
    if "d" not in _ONE_RANK:
        r = _run(["--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
        assert r.returncode == 0, r.stderr[-3000:]
        _ONE_RANK["d"] = _json_line(r.stdout)
    return _ONE_RANK["d"]


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 8])
def test_bench_sharded_configs_do_exactly_the_whole_batchs_work(world):
    """`bench.py --gpus N` WITH every config (C1, C3 ... C6b through run_config's shard path: C3 at 512 / N problems per rank, C5 at
    64 / N, C1's single problem on rank 0 alone), N ranks over gloo sharing this box's GPU - the form of the run the driver's
    8-GPU node executes over RCCL.  Problems are independent and every kernel's result is independent of its batch,
    so the ranks together must do EXACTLY the single-rank run's work: the same iterations per config (rank-summed), every problem
    converged, batch_per_gpu = ceil(B / N); the line says how many ranks and which collective."""
    one = _one_rank_configs()
    r = _run(["--gpus", str(world), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], env={"MI_BENCH_BACKEND": "gloo"}, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == world and d["scaling"] == "weak" and d["config"]["global_batch"] == 1024 * world
    assert d["config"]["parallelism"] == f"batch-shard x{world}" and "all_reduce(MIN)" in d["config"]["collective"]
    # round 6: the N > 1 line checks itself - the communicator's rank count, every rank's own iteration sum, the strong-scaling
    # figure (C2's 1024 problems sharded over the ranks) beside the weak one, and why the library's own communicator was not
    # used (gloo ranks share this box's one GPU; on the driver's node it is, after a self-check)
    assert d["communicator_ranks"] == world and "share a device" in d["config"]["collective_fallback_reason"]
    ipr = d["iterations_per_rank"]
    assert len(ipr) == world and all(v > 0 for v in ipr) and abs(sum(ipr) - d["value"] * d["ms_per_step"] * 1e-3 * d["steps"]) < 1e-6 * sum(ipr)
    sg = d["strong_scaling"]
    assert d["value_strong"] == sg["value"] > 0 and sg["global_batch"] == 1024 and len(sg["iterations_per_rank"]) == world
    # the shards of the single-GPU run's own batch do exactly its work: K steps x the single-rank run's iterations per step
    assert abs(sum(sg["iterations_per_rank"]) - d["steps"] * one["iterations_per_step_rank0"]) < 0.5
    ref = {c["name"]: c for c in one["configs"]}
    names = [c["name"] for c in d["configs"]]
    assert [n[:3].strip() for n in names] == ["C1", "C3", "C4", "C5", "C5q", "C5q", "C6", "C6b"], names     # (the B = 8 shard line is a 1-rank entry)
    for c in d["configs"]:
        o = ref[c["name"]]
        assert c["batch"] == o["batch"] and c["batch_per_gpu"] == -(-c["batch"] // world), c["name"]
        assert c["iterations"] == o["iterations"], (c["name"], c["iterations"], o["iterations"])
        assert c["converged"] == o["converged"] and (c["converged"] == c["batch"] or c["name"].startswith("C4")), (c["name"], c["converged"], o["converged"])
        assert c["max_iterations_per_problem"] == o["max_iterations_per_problem"]
        assert c["iterations_per_s"] > 0 and c["ms_per_solve"] > 0
    assert d["boundary_inclusive"] is not None and d["concurrent_batches"] is None
    # headline: weak scaling, rank 0 solves its 1024-problem block of the 1024 N drawn with one seed (another draw than the single-rank batch)
    assert abs(d["iterations_per_step_rank0"] - one["iterations_per_step_rank0"]) < 0.05 * one["iterations_per_step_rank0"] and d["converged_rank0"] == 1024


@pytest.mark.gpu
def test_bench_native_rccl_communicator_two_ranks():
    """MI_BENCH_NATIVE_RCCL=1: the best-cost reduction through the LIBRARY's communicator (mi_ilqr_comm_create over librccl, the C
    caller's path), the ranks and the 128-byte id shipped by torch.distributed.  With two GPUs it runs; on a one-GPU box RCCL
    refuses two ranks on one device - the run must then stop with the library's own message, not hang or fall back."""
    import torch
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-configs"],
             env={"MI_BENCH_BACKEND": "gloo", "MI_BENCH_NATIVE_RCCL": "1"}, timeout=600)
    if torch.cuda.device_count() >= 2:
        assert r.returncode == 0, r.stderr[-3000:]
        assert "mi_ilqr_allreduce_min_start" in _json_line(r.stdout)["config"]["collective"]
    else:
        assert r.returncode != 0
        assert "ncclCommInitRank failed" in r.stderr and "mi_ilqr_comm_create" in r.stderr, r.stderr[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("stall", [False, True])
def test_bench_falls_back_when_the_library_communicator_fails_or_never_answers(stall):
    """The default N > 1 run tries the library's own communicator first.  Two ranks on this box's one GPU: RCCL refuses the second
    rank (an error), or - MI_BENCH_NATIVE_TEST_STALL - the attempt never returns (a bootstrap that hangs); either way every rank
    agrees on torch.distributed, the reason is in the line and the run finishes."""
    import torch
    if torch.cuda.device_count() >= 2 and not stall:
        pytest.skip("two GPUs: the communicator comes up (test_bench_native_rccl_communicator_two_ranks)")
    env = {"MI_BENCH_BACKEND": "gloo", "MI_BENCH_NATIVE_RCCL": "try"}
    if stall:
        env.update(MI_BENCH_NATIVE_TEST_STALL="1", MI_BENCH_NATIVE_DEADLINE_S="5")
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-configs"], env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert "torch.distributed all_reduce(MIN)" in d["config"]["collective"]
    why = d["config"]["collective_fallback_reason"]
    assert ("no answer from librccl within" in why) if stall else ("mi_ilqr_comm_create: RCCL error" in why), why
    assert d["n_gpus"] == 2 and d["value"] > 0
