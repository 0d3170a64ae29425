"""Copy the artefacts of tools/refresh_profiles.sh <tag> from gpurun_out/ (scratch) into profiles/ (tracked):
    python tools/install_profiles.py r01l [--drop r01k]
Builds profiles/<tag>_pmc_c2.json (what bench.py falls back to) and the per-launch counter CSV."""
import csv, glob, json, os, shutil, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
tag = sys.argv[1]
out, prof = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
for name in ("bench_c2.json", "bench_c2_kernel_stats.csv", "all_configs.jsonl", "all_configs_kernel_stats.csv", "pmc_issue.json", "c2_clock_ramp.txt", "c2_modes_batch_sweep.txt"):
    if os.path.exists(os.path.join(out, f"{tag}_{name}")):
        shutil.copy(os.path.join(out, f"{tag}_{name}"), os.path.join(prof, f"{tag}_{name}"))
raw = json.load(open(os.path.join(out, f"{tag}_pmc_raw.json")))
bench = json.loads(open(os.path.join(out, f"{tag}_bench_c2.json")).read().strip().splitlines()[-1])
N, n, m, B = 200, 2, 1, 1024
rec = {
    "round": int(tag[1:3]), "revision": tag[3:],
    "workload": "C2 pendulum B=1024 N=200 fp64 FD, cold-start solve",
    "kernel": "ilqr_small_kernel<Pendulum,0,0>",
    "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python bench.py "
               "--no-cpu-baseline --steps 5 --warmup 1 (two separate passes; MI_BENCH_NESTED=1: headline only)",
    **raw,
    "gfx950_read_correction": "FETCH_SIZE x2 per MI355X_MICROARCH.md (wide coalesced reads are tallied at 64 B per 128-B request); "
                              "round 5 calibrated 8 B/lane streams too (tools/ubench/stream8, profiles/r05_pmc_throughput.json: 2048 B read per "
                              "counted KB, WRITE_SIZE 1024 B per KB)",
    "expected_bytes_per_launch": {"read": B * 8 * (m * (N - 1) + n) + 8 * 11,
                                  "write": B * 8 * (n * N + (m + m * n + m + 1 + n * n + n * m) * (N - 1))},
    "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
}
json.dump(rec, open(os.path.join(prof, f"{tag}_pmc_c2.json"), "w"), indent=1)
rows, hdr = [], None
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(out, f"{tag}_pmc_{c}", "**", "*counter_collection.csv"), recursive=True):
        r = list(csv.reader(open(f)))
        hdr = r[0]
        ki, ci = hdr.index("Kernel_Name"), hdr.index("Counter_Name")
        rows += [x for x in r[1:] if "ilqr_small_kernel" in x[ki] and x[ci] == c]
with open(os.path.join(prof, f"{tag}_pmc_c2_fetch_write.csv"), "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(hdr)
    w.writerows(rows)
if "--drop" in sys.argv:
    old = sys.argv[sys.argv.index("--drop") + 1]
    for f in glob.glob(os.path.join(prof, f"{old}_*")):
        os.remove(f)
print("installed", sorted(os.path.basename(f) for f in glob.glob(os.path.join(prof, f"{tag}_*"))))
