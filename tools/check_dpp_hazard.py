#!/usr/bin/env python
"""Static check of the DPP read-after-VALU-write hazard in the built kernels (no GPU needed).

A DPP instruction must not read a VGPR that a VALU instruction wrote fewer than two wait states earlier (every instruction is
one wait state, s_nop N is N + 1).  The compiler's hazard recognizer guarantees that for the instructions it emits - not for
inline assembly, and the Gauss-Jordan elimination's column update is inline assembly (ilqr_large.hpp: v_fmac_f64_dpp with a
row_newbcast source).  This walks the ISA of every kernel object and reports each DPP instruction whose DPP source overlaps the
destination of a VALU instruction within the window.  Straight-line check: a label (branch target) between writer and reader
is treated like any instruction (conservative in neither direction: a loop back-edge is checked against the fall-through
predecessor only - the elimination is fully unrolled, no DPP read follows a branch target within two slots in these kernels,
and the tool reports the number of DPP reads it saw right after a label so that this stays visible).

    python tools/check_dpp_hazard.py [objects ...]        (default: drake_ddp_amd/lib/obj/k_*.o)
Exit status 1 when a violation is found.
"""
import glob
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_mix

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def regs(tok):
    """'v[4:5]' / 'v7' / '-v[2:3]' / '|v3|' -> set of VGPR numbers (empty for anything else)."""
    tok = tok.strip().strip("-|").replace("neg(", "").replace("abs(", "").rstrip(")")
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def check(body):
    bad, seen, after_label = [], 0, 0
    window = []                                   # (wait states since, dst regs, text) of recent VALU writers
    since_label = 99
    for line in body.splitlines():
        line = line.split("//")[0].strip()
        if not line:
            continue
        if line.endswith(":") or re.match(r"^[0-9a-f]+ <", line):
            since_label = 0
            continue
        parts = line.split(None, 1)
        op, ops = parts[0], (parts[1] if len(parts) > 1 else "")
        operands = [o.strip() for o in ops.split(",")]
        is_dpp = "_dpp" in op or "row_newbcast" in ops or "row_shr" in ops or "row_shl" in ops or "quad_perm" in ops or "row_bcast" in ops or "row_ror" in ops or "wave_" in ops or "row_mirror" in ops or "row_half_mirror" in ops or "row_share" in ops
        if is_dpp and op.startswith("v_"):
            seen += 1
            if since_label < 2:
                after_label += 1
            src = regs(operands[1].split()[0]) if len(operands) > 1 else set()
            for ws, dst, text in window:
                if ws < 2 and (dst & src):
                    bad.append((text, line, ws))
        states = 1
        if op == "s_nop":
            states = int(ops.strip() or "0", 0) + 1
        window = [(ws + states, dst, text) for ws, dst, text in window if ws + states < 2]
        if op.startswith("v_") and not op.startswith("v_cmp") and operands:
            d = regs(operands[0])
            if d:
                window.append((0, d, line))
        since_label += 1
    return bad, seen, after_label


def main(paths):
    total_bad = total = labels = 0
    for obj in paths:
        fns = isa_mix.functions(isa_mix.device_asm(obj))
        for name, body in fns.items():
            bad, seen, al = check(body)
            total += seen
            labels += al
            for w, r, ws in bad:
                total_bad += 1
                print(f"{os.path.basename(obj)}: {isa_mix.demangle(name)[:90]}\n    writer: {w}\n    reader: {r}   ({ws} wait state(s) between)")
    print(f"{total} DPP instructions in {len(paths)} objects: {total_bad} hazard violation(s); {labels} DPP reads within two slots of a label")
    return 1 if total_bad else 0


if __name__ == "__main__":
    args = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "drake_ddp_amd", "lib", "obj", "k_*.o")))
    sys.exit(main(args))
