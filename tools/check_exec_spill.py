#!/usr/bin/env python
"""Static check of the built kernel objects for a miscompile of this hipcc that round 5 ran into: a VGPR spill
copy - v_accvgpr_write_b32 (VGPRs are spilled to the accumulation registers first) or a scratch store - placed at the top of a
control-flow join block BEFORE the `s_or_b64 exec, exec, s[..]` that re-activates the lanes.  The block is the target of an
`s_cbranch_execz`, or is fallen into from a divergent loop that leaves with EXEC = 0: the copy then executes with no lane active
and writes nothing, and the reload further down returns whatever the slot held before.  (Seen in ilqr_large_kernel<Arm27C, 0,
MODE_MPC>: the helper workgroups' `last_round` was saved that way, the stale value made them repeat a finished round for ever.)

    python tools/check_exec_spill.py [objects...]      (default: every kernel object of the library)

Prints one line per suspicious site and exits non-zero if there is any."""
import glob
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_mix as I

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SPILL = re.compile(r"^(v_accvgpr_write_b32|scratch_store_\w+)\b")
EXEC_RESTORE = re.compile(r"^s_or_b64 exec, exec, s\[")
IGNORES_EXEC = re.compile(r"^(v_writelane_b32|v_readlane_b32|v_readfirstlane_b32|s_|ds_nop)")
BRANCH = re.compile(r"^(s_cbranch_\w+|s_branch)\s+(\d+)")


def sites(body):
    ins = []
    for l in body.splitlines():
        m = re.match(r"\s*(\S.*?)\s*//\s*([0-9A-Fa-f]+):", l)
        if m:
            ins.append((int(m.group(2), 16), m.group(1)))
    addr_index = {a: i for i, (a, _) in enumerate(ins)}
    execz_targets, all_targets = set(), set()
    for i, (a, t) in enumerate(ins):
        m = BRANCH.match(t)
        if not m:
            continue
        off = int(m.group(2))
        if off >= 32768:
            off -= 65536
        nxt = ins[i + 1][0] if i + 1 < len(ins) else a + 4
        tgt = nxt + 4 * off
        all_targets.add(tgt)
        if m.group(1) == "s_cbranch_execz":
            execz_targets.add(tgt)
    out = []
    for tgt in sorted(execz_targets):
        i = addr_index.get(tgt)
        if i is None:
            continue
        pending = []
        for a, t in ins[i:i + 48]:
            if a != tgt and a in all_targets:
                break                                        # another block starts
            if EXEC_RESTORE.match(t):
                out += pending
                break
            # (anything else that writes EXEC ends the prologue: the arms of a structured if / else set their own lane masks - a copy
            #  there is a per-lane phi, executed by exactly the lanes that own the value)
            if BRANCH.match(t) or "saveexec" in t or re.match(r"^s_(and|andn2|or|xor|mov)\w*_b64 exec", t) or t.startswith("s_barrier") or t.startswith("s_endpgm"):
                break
            if SPILL.match(t):
                pending.append((a, t))
    return out


def main():
    objs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "drake_ddp_amd", "lib", "obj", "k_*.o")))
    objs = [o for o in objs if not re.search(r"-[0-9a-f]{8}\.o$", o)]
    bad = 0
    for o in objs:
        asm = I.device_asm(o)
        for name, body in I.functions(asm).items():
            for a, t in sites(body):
                bad += 1
                print("%s  %s  %x  %s" % (os.path.basename(o), I.demangle(name)[:80], a, t))
    print("%d kernel object(s), %d spill cop%s under a zero EXEC mask" % (len(objs), bad, "y" if bad == 1 else "ies"))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
