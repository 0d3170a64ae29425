"""HBM counter evidence for the lane-per-problem kernels (ilqr_batch.hpp) - the one kernel family of the repo that STREAMS its
state through HBM:  python tools/pmc_throughput.py <tag>   (on the GPU box; writes gpurun_out/<tag>_pmc_throughput.json)
  1. calibration: tools/ubench/stream8 (8 B / lane coalesced reads / writes of 2 GiB, past the Infinity Cache) under
     `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, kernel trace only): bytes moved per counted KB for
     THIS access width (MI355X_MICROARCH.md calibrates 16 B / lane reads only);
  2. the acrobot (n = 4, m = 1, N = 40) at B = 16 384 and 262 144, every step a key-point (KP = false instantiation) and
     adaptiveJerk key-points (KP = true: per-lane lists, masked gathers), the same two passes + a timing pass;
  3. per workload: calibrated HBM bytes per launch, / SURVEY 8(d)'s algorithmic bytes, GB/s against the 8 TB/s peak and the
     6.3 TB/s the guide calls achievable."""
import csv, glob, json, os, shutil, subprocess, sys, tempfile
tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
env = dict(os.environ, TMPDIR="/tmp")
PEAK, ACHIEVABLE = 8000.0, 6300.0


def counters(cmd, counter, match):
    out = tempfile.mkdtemp(prefix="mi_pmc_", dir="/tmp")
    subprocess.run(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "--"] + cmd, cwd="/tmp", env=env,
                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    per = {}
    for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if match in r["Kernel_Name"] and r["Counter_Name"] == counter:
                per.setdefault(r["Kernel_Name"].split("(")[0].replace("void ", ""), []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    shutil.rmtree(out, ignore_errors=True)
    return {k: [v for _, v in sorted(vs)] for k, vs in per.items()}


# ---- 1. calibration
exe = os.path.join(ROOT, "tools", "ubench", "stream8")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", exe + ".hip", "-o", exe])
MiB = 2048
timing = subprocess.run([exe, str(MiB)], stdout=subprocess.PIPE, env=env).stdout.decode().strip().splitlines()[-1]
cal = {"bytes_per_kernel": MiB << 20, "timing": timing}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    got = counters([exe, str(MiB)], c, "8")
    for k, v in got.items():
        name = k.split("<")[0]
        cal.setdefault(name, {})[c + "_KB"] = sum(v[1:]) / max(1, len(v[1:]))
b = float(MiB << 20)
cal["read_bytes_per_FETCH_KB"] = b / cal["read8"]["FETCH_SIZE_KB"]
cal["write_bytes_per_WRITE_KB"] = b / cal["write8"]["WRITE_SIZE_KB"]
cal["copy8_check"] = {"read": cal["copy8"]["FETCH_SIZE_KB"] * cal["read_bytes_per_FETCH_KB"] / b, "write": cal["copy8"]["WRITE_SIZE_KB"] * cal["write_bytes_per_WRITE_KB"] / b}
print("calibration:", json.dumps(cal), flush=True)

# ---- 2. / 3. the solves
runs = []
for B in (16384, 262144):
    for kp in ("none", "adaptiveJerk"):
        cmd = [sys.executable, os.path.join(ROOT, "tools", "run_throughput.py"), str(B), kp]
        plain = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, cwd=ROOT, timeout=900).stdout.decode().strip().splitlines()
        rec = json.loads([l for l in plain if l.startswith("{")][-1])
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            got = counters(cmd, c, "ilqr_batch_kernel")
            assert len(got) == 1, got.keys()
            k, v = next(iter(got.items()))
            rec["kernel"] = k
            rec[c + "_KB_per_launch"] = sum(v[1:]) / max(1, len(v[1:]))      # (three launches per run; the first is left out)
        rd = rec["FETCH_SIZE_KB_per_launch"] * cal["read_bytes_per_FETCH_KB"]
        wr = rec["WRITE_SIZE_KB_per_launch"] * cal["write_bytes_per_WRITE_KB"]
        t = rec["kernel_ms"] * 1e-3
        rec.update({"hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
                    "real_over_algorithmic": (rd + wr) / rec["algorithmic_bytes"],
                    "algorithmic_GBps": rec["algorithmic_bytes"] / t / 1e9, "real_GBps": (rd + wr) / t / 1e9,
                    "algorithmic_frac_of_8TBps": rec["algorithmic_bytes"] / t / 1e9 / PEAK, "real_frac_of_8TBps": (rd + wr) / t / 1e9 / PEAK,
                    "real_frac_of_achievable_6.3TBps": (rd + wr) / t / 1e9 / ACHIEVABLE,
                    "M_iterations_per_s": rec["iterations"] / t / 1e6})
        runs.append(rec)
        print(json.dumps(rec), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python tools/run_throughput.py B kp  (one pass per counter; "
                      "kernel_ms from an un-profiled run's HIP events)",
           "workload": "acrobot n=4 m=1 N=40, fp64, central differences, cold-start batched solve, lane-per-problem kernels",
           "calibration": cal, "runs": runs}, open(os.path.join(ROOT, "gpurun_out", tag + "_pmc_throughput.json"), "w"), indent=1)
