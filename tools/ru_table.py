#!/usr/bin/env python
"""Table of the kernels' register / spill / scratch figures from a `python -m drake_ddp_amd.build --resource-usage` log
(drake_ddp_amd/lib/resource_usage.txt), optionally as a diff against an older log.

    python tools/ru_table.py [log] [--against older_log] [--filter substring]
"""
import re
import subprocess
import sys

KEYS = ("VGPRs", "AGPRs", "SGPRs Spill", "VGPRs Spill", "ScratchSize [bytes/lane]")


def parse(path):
    out, cur = {}, None
    for line in open(path, errors="replace"):
        m = re.search(r"remark:\s+(Function Name|" + "|".join(re.escape(k) for k in KEYS) + r"):\s+(\S+)", line)
        if not m:
            continue
        if m.group(1) == "Function Name":
            cur = out.setdefault(m.group(2), {})
        elif cur is not None:
            cur[m.group(1)] = int(m.group(2))
    return out


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, (s.replace("mi::", "").replace("(KArgs)", "").replace("void ", "") for s in r)))


def row(d):
    return "vgpr %3d agpr %3d sgpr-spill %4d vgpr-spill %3d scratch %4d" % tuple(d.get(k, -1) for k in KEYS)


if __name__ == "__main__":
    args = sys.argv[1:]
    flt = args[args.index("--filter") + 1] if "--filter" in args else ""
    old = parse(args[args.index("--against") + 1]) if "--against" in args else None
    pos = [a for i, a in enumerate(args) if not a.startswith("--") and (i == 0 or args[i - 1] not in ("--filter", "--against"))]
    new = parse(pos[0] if pos else "drake_ddp_amd/lib/resource_usage.txt")
    names = demangle(sorted(new))
    for k in sorted(new, key=lambda k_: names[k_]):
        if flt and flt not in names[k]:
            continue
        if old is None:
            print(f"{names[k][:72]:72s} {row(new[k])}")
        elif old.get(k) != new[k]:
            print(f"{names[k][:72]:72s} {row(old[k]) if k in old else '(new)'}\n{'':72s} {row(new[k])}")
