#!/bin/bash
# Regenerate the measurement artefacts that profiles/ holds (run on the GPU box through gpurun):
#   tools/refresh_profiles.sh <tag>      e.g. r01f
# Outputs land in gpurun_out/<tag>_*; copy the ones to be judged into profiles/.
set -u
TAG=${1:-rXX}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
python bench.py > $OUT/${TAG}_bench_c2.json 2> $OUT/${TAG}_bench_c2.err
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -- python $REPO/bench.py --no-cpu-baseline --no-configs > $OUT/${TAG}_prof.log 2>&1 )
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && MI_BENCH_NESTED=1 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_$C -- python $REPO/bench.py --no-cpu-baseline --steps 5 --warmup 1 > $OUT/${TAG}_pmc_$C.log 2>&1 )
done
python - > $OUT/${TAG}_all_configs.jsonl <<PY
import json
d = json.loads(open("$OUT/${TAG}_bench_c2.json").read().strip().splitlines()[-1])
for c in d["configs"]:
    print(json.dumps(c))
print(json.dumps({"boundary_inclusive": d["boundary_inclusive"]}))
PY
find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_bench_c2_kernel_stats.csv
python - <<PY
import csv, glob, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = []
    for f in glob.glob("$OUT/${TAG}_pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        rows += [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "ilqr_small_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c]
    out[c] = sum(rows[1:]) / max(1, len(rows[1:]))
json.dump({"FETCH_SIZE_KB_per_launch": out["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": out["WRITE_SIZE"],
           "hbm_bytes_per_launch_corrected": (2 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024,
           "hbm_bytes_per_launch_uncorrected": (out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024}, open("$OUT/${TAG}_pmc_raw.json", "w"), indent=1)
PY
( cd /tmp && MI_RUN_SHARD=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_all -- python $REPO/tools/run_configs.py > $OUT/${TAG}_run_configs.log 2>&1 )
find $OUT/${TAG}_prof_all -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_all_configs_kernel_stats.csv
python tools/pmc_issue.py ${TAG} > $OUT/${TAG}_pmc_issue.log 2>&1
python tools/warm_sweep.py 2>/dev/null | grep events > $OUT/${TAG}_c2_clock_ramp.txt
( echo "# python tools/bsweep_modes.py on one MI355X ($TAG): C2 pendulum N=200 fp64 FD, cold-start solve, kernel time from HIP events of one"
  echo "# blocking solve.  latency = wave-per-problem (time-parallel passes), throughput = lane-per-problem."
  python tools/bsweep_modes.py 2>/dev/null | grep "B=" ) > $OUT/${TAG}_c2_modes_batch_sweep.txt
tail -1 $OUT/${TAG}_bench_c2.json | cut -c1-200
head -3 $OUT/${TAG}_bench_c2_kernel_stats.csv
cat $OUT/${TAG}_pmc_raw.json
