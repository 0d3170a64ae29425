"""Issue-side hardware counters of the solve kernels (run on the GPU box):  python tools/pmc_issue.py <tag>
rocprofv3 --pmc passes (one counter group per pass, kernel trace only) over tools/run_configs.py, which runs the
five BASELINE configs; per kernel, the counters are averaged over its launches after the first.  Writes
gpurun_out/<tag>_pmc_issue.json.  Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count
quad-cycles summed over waves; SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE, SQ_VALU_MFMA_BUSY_CYCLES count cycles."""
import csv, glob, json, os, subprocess, sys, tempfile, shutil
tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = [
    ["GRBM_GUI_ACTIVE", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"],
    ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_SALU", "SQ_WAIT_INST_ANY"],
    ["SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"],
    ["SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_TRANS_F64"],
    ["SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA", "SQ_INSTS_VMEM"],
]
env = dict(os.environ, TMPDIR="/tmp")
acc = {}
for grp in GROUPS:
    out = tempfile.mkdtemp(prefix="mi_pmc_", dir="/tmp")
    cmd = ["rocprofv3", "--pmc"] + grp + ["--kernel-trace", "--output-format", "csv", "-d", out, "--",
                                           sys.executable, os.path.join(ROOT, "tools", "run_configs.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    rows = []
    for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    if not rows:
        print("no counters for", grp, r.stdout.decode()[-400:], file=sys.stderr)
    per = {}
    for row in rows:
        k = row["Kernel_Name"].split("(")[0]
        if "ilqr_" not in k:
            continue
        k = "%s  grid=%s x %s threads" % (k.replace("void ", ""), int(row["Grid_Size"]) // max(1, int(row["Workgroup_Size"])), row["Workgroup_Size"])
        per.setdefault((k, row["Counter_Name"]), []).append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
    for (k, c), v in per.items():
        v.sort()
        vals = [x for _, x in v][1:] or [x for _, x in v]
        acc.setdefault(k, {})[c] = sum(vals) / len(vals)
        acc[k]["launches_averaged"] = len(vals)
    shutil.rmtree(out, ignore_errors=True)
N_SIMD = 1024
N_XCD = 8           # GRBM_GUI_ACTIVE comes back summed over the eight XCDs (checked against the kernel's duration)
for k, c in acc.items():
    g = c.get("GRBM_GUI_ACTIVE", 0) / N_XCD
    c["kernel_cycles (GRBM_GUI_ACTIVE / 8 XCDs)"] = g
    if g:
        c["derived"] = {
            "wave_slots_occupied: SQ_WAVE_CYCLES*4 / (kernel_cycles * 1024 SIMDs)  [1.0 = one wave on every SIMD for the whole launch]": 4 * c.get("SQ_WAVE_CYCLES", 0) / (g * N_SIMD),
            "valu_issue_fraction_of_resident_time (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES)": c.get("SQ_ACTIVE_INST_VALU", 0) / max(1.0, c.get("SQ_WAVE_CYCLES", 0)),
            "waiting_fraction_of_resident_time (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES)": c.get("SQ_WAIT_INST_ANY", 0) / max(1.0, c.get("SQ_WAVE_CYCLES", 0)),
            "valu_busy_fraction_of_chip: SQ_ACTIVE_INST_VALU*4 / (kernel_cycles * 1024 SIMDs)": 4 * c.get("SQ_ACTIVE_INST_VALU", 0) / (g * N_SIMD),
            "fp64_valu_share_of_valu_instructions": (c.get("SQ_INSTS_VALU_FMA_F64", 0) + c.get("SQ_INSTS_VALU_ADD_F64", 0) + c.get("SQ_INSTS_VALU_MUL_F64", 0)
                                                     + c.get("SQ_INSTS_VALU_TRANS_F64", 0)) / max(1.0, c.get("SQ_INSTS_VALU", 0)),
            "mfma_busy_cycles_per_mfma_instruction": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(1.0, c.get("SQ_INSTS_MFMA", 0)),
            "cycles_per_valu_instruction_of_a_resident_wave (4 * SQ_WAVE_CYCLES / SQ_INSTS_VALU)": 4 * c.get("SQ_WAVE_CYCLES", 0) / max(1.0, c.get("SQ_INSTS_VALU", 0)),
            "lds_bank_conflict_fraction (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE)": c.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, c.get("SQ_LDS_IDX_ACTIVE", 0)),
            "mfma_busy_fraction_of_chip: SQ_VALU_MFMA_BUSY_CYCLES / (kernel_cycles * 1024 SIMDs)": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (g * N_SIMD),
        }
# fp64 work per launch from the instruction counters (wave-instructions x 64 lanes, i.e. an upper bound where exec
# masks are partial; an FMA = 2 flops, one v_mfma_f64_16x16x4 = 2048 flops, one v_mfma_f64_4x4x4_4b = 512) and, per BASELINE
# config, per iLQR iteration: the iteration counts come from one un-profiled run of the same script (the solves are
# deterministic).  SQ_INSTS_MFMA does not tell the two MFMA shapes apart: their ratio is the static one of the kernel's ISA
# (both sit in the same backward-pass loops; tools/isa_mix.py reads the built objects).
def mfma_small_fraction(kernel_name):
    import re
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_mix
    m = re.search(r"ilqr_\w+_kernel<mi::(\w+)", kernel_name)
    model = {"Synth36": "synth36", "PlanarQuad": "planar_quad", "Quad3D": "quad3d", "Arm27": "arm27", "Arm27C": "arm27c", "Acrobot": "acrobot",
             "CartPoleT": "cartpole_wall", "Pendulum": "pendulum"}.get(m.group(1) if m else "", None)
    obj = os.path.join(ROOT, "drake_ddp_amd", "lib", "obj", "k_%s.o" % model) if model else None
    if not obj or not os.path.exists(obj):
        return 0.0
    try:
        fns = isa_mix.functions(isa_mix.device_asm(obj))
    except Exception:
        return 0.0
    want = kernel_name.split("  grid=")[0].replace(" ", "")
    n16 = n4 = 0
    for name, body in fns.items():
        if isa_mix.demangle(name).split("(")[0].replace("void ", "").replace(" ", "") == want:
            n16, n4 = body.count("v_mfma_f64_16x16x4"), body.count("v_mfma_f64_4x4x4")
    return n4 / float(n16 + n4) if n16 + n4 else 0.0


for k, c in acc.items():
    f4 = mfma_small_fraction(k) if c.get("SQ_INSTS_MFMA", 0) else 0.0
    c["mfma_4x4x4_fraction_of_mfma_instructions (static ISA)"] = f4
    c["fp64_flops_per_launch"] = 64.0 * (2 * c.get("SQ_INSTS_VALU_FMA_F64", 0) + c.get("SQ_INSTS_VALU_ADD_F64", 0) + c.get("SQ_INSTS_VALU_MUL_F64", 0)
                                         + c.get("SQ_INSTS_VALU_TRANS_F64", 0)) + (2048.0 * (1.0 - f4) + 512.0 * f4) * c.get("SQ_INSTS_MFMA", 0)
r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_configs.py")], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600)
lines = [json.loads(l) for l in r.stdout.decode().splitlines() if l.startswith("{")]
iters = {l["config"].split()[0] + ("/8" if "shard" in l["config"] else ""): l.get("iters_total", l.get("iters_per_solve")) for l in lines}
KERNELS = {   # config -> (model substring, workgroups of the launch or None); every kernel mode of that model (and grid) is the
    # config's.  Only the two pendulum configs share a model and need the grid to tell them apart; the workgroup-per-problem
    # kernels are matched by MODEL alone - their grid depends on the cluster size the launch picks (round 5 lost C5's entry to
    # a "grid=64" filter when its launch became 256 workgroups; run_configs.py keeps the B = 8 shard out of these passes).
    "C1": ("Pendulum", 1), "C2": ("Pendulum", 1024), "C3": ("Acrobot", None), "C4": ("CartPoleT<true>", None),
    "C5": ("Synth36", None), "C5q": ("PlanarQuad", None), "C5q3d": ("Quad3D", None), "C6": ("Arm27,", None), "C6b": ("Arm27C,", None)}
per_config = {}
missing = {}
for cfg, (model, grid) in KERNELS.items():
    ks = [k for k in acc if model in k and (grid is None or ("grid=%d x" % grid) in k)]
    key = cfg
    if not ks or key not in iters:
        missing[cfg] = "no kernel matching %r%s among the profiled launches" % (model, "" if grid is None else " with grid=%d" % grid) if not ks \
            else "tools/run_configs.py printed no line for %s" % cfg
        continue
    fl = sum(acc[k]["fp64_flops_per_launch"] for k in ks)
    wc = sum(acc[k].get("SQ_WAVE_CYCLES", 0) for k in ks)
    kc = sum(acc[k].get("kernel_cycles (GRBM_GUI_ACTIVE / 8 XCDs)", 0) for k in ks)
    per_config[cfg] = {
        "kernels": ks, "iterations_per_run": iters[key], "fp64_flops_per_run": fl, "fp64_flops_per_iteration": fl / max(1.0, iters[key]),
        "wave_slots_occupied": 4 * wc / max(1.0, kc * N_SIMD)}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
path = os.path.join(ROOT, "gpurun_out", tag + "_pmc_issue.json")
json.dump({"command": "rocprofv3 --pmc <group> --kernel-trace --output-format csv -- python tools/run_configs.py (one pass per group)",
           "groups": GROUPS, "kernels": acc, "configs": per_config, "missing": missing}, open(path, "w"), indent=1)
for k, why in missing.items():
    print("MISSING %-10s %s" % (k, why))
for k, c in per_config.items():
    print("%-10s fp64 flops/iteration %.4g   wave slots %.3f" % (k, c["fp64_flops_per_iteration"], c["wave_slots_occupied"]))
for k, c in acc.items():
    print(k)
    for name, v in c.get("derived", {}).items():
        print("   %-120s %.4f" % (name, v))
if missing:
    sys.exit("pmc_issue: no counter entry for " + ", ".join(sorted(missing)))
