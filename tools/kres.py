#!/usr/bin/env python
"""Print per-kernel register/scratch/LDS usage (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys
src = sys.argv[1] if len(sys.argv) > 1 else "drake_ddp_amd/csrc/mi_ilqr.hip"
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-fast-math",
                      "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/_kres.o"],
                     capture_output=True, text=True).stderr
cur = {}
rows = []
for line in out.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+: +(\w[\w ]*?): +(\S+)", line) or re.search(r":\d+:\d+: remark: +(\w[\w ]*?): +(\S+)", line)
    m = re.search(r"(Function Name|Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs Spill|VGPRs Spill): +(\S+)", line)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k in ("Function Name", "Name"):
        cur = {"name": v}; rows.append(cur)
    else:
        cur[k] = v
for r in rows:
    if flt and flt not in r["name"]: continue
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    print(f"{name[:90]:90s} sgpr={r.get('TotalSGPRs')} vgpr={r.get('VGPRs')} agpr={r.get('AGPRs')} scratch={r.get('ScratchSize [bytes/lane]')} occ={r.get('Occupancy [waves/SIMD]')} spillV={r.get('VGPRs Spill')}")
