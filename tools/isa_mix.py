#!/usr/bin/env python
"""Instruction mix of a kernel from the ISA of a built object (no GPU needed): total instructions and the counts of the
classes that tell spill / register-shuffle traffic from arithmetic - v_readlane / v_writelane (SGPR spills live in VGPR
lanes), v_accvgpr_read / write (VGPR spills to AGPRs), scratch loads / stores, s_mov re-materializations, DPP moves, fp64
arithmetic, MFMA, LDS traffic.  Optionally restricted to the LOOP that contains most of a given opcode (--loop OPC):
the innermost backward-branch range with the largest count of it.

    python tools/isa_mix.py drake_ddp_amd/lib/obj/k_pendulum.o 'ilqr_small_kernel<mi::Pendulum, 0, 0>' [--loop v_fma_f64]
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def device_asm(obj):
    d = tempfile.mkdtemp(prefix="isa_")
    base = os.path.join(d, os.path.basename(obj))
    subprocess.run(["cp", obj, base], check=True)
    subprocess.run([OBJDUMP, "--offloading", base], cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    dev = [f for f in os.listdir(d) if "amdgcn" in f]
    if not dev:
        raise SystemExit("no device code object in " + obj)
    return subprocess.run([OBJDUMP, "-d", os.path.join(d, dev[0])], capture_output=True, text=True).stdout


def functions(asm):
    parts = re.split(r"\n[0-9a-f]+ <(\S+)>:\n", asm)
    return {parts[i]: parts[i + 1] for i in range(1, len(parts), 2)}


def demangle(n):
    return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()


CLASSES = [("fp64 VALU", r"^v_(fma|add|mul|min|max|rcp|rsq|sqrt|ldexp|frexp|fract|trunc|rndne|cmp\w*|div\w*|cvt\w*)_?\w*f64"),
           ("MFMA", r"^v_mfma"), ("v_readlane/writelane", r"^v_(readlane|writelane|readfirstlane)"),
           ("v_accvgpr_read/write", r"^v_accvgpr"), ("scratch", r"^scratch_"), ("DPP", r"_dpp$|^v_mov_b64_dpp|^v_mov_b32_dpp"),
           ("s_mov", r"^s_mov"), ("other SALU", r"^s_(?!mov|waitcnt|nop|barrier|cbranch|branch)"), ("s_waitcnt/nop", r"^s_(waitcnt|nop)"),
           ("branches", r"^s_(cbranch|branch)"), ("LDS", r"^ds_"), ("global/flat", r"^(global|flat|buffer)_"),
           ("v_mov/cndmask/other VALU", r"^v_")]


def classify(ops):
    out = collections.Counter()
    for op, c in ops.items():
        for name, pat in CLASSES:
            if re.search(pat, op):
                out[name] += c
                break
        else:
            out["other"] += c
    return out


def parse(body):
    ins = []
    for l in body.splitlines():
        m = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):(.*)", l)
        if m:
            t = re.search(r"<\S+\+0x([0-9a-fA-F]+)>", m.group(4))
            ins.append((int(m.group(3), 16), m.group(1), m.group(2) + (" <+0x%s>" % t.group(1) if t else "")))
    return ins


def loops_with(ins, opc):
    """every backward-branch range holding `opc`: (count, first, last, size), innermost (no other such range inside) first."""
    addr = {a: i for i, (a, _, _) in enumerate(ins)}
    found = []
    for i, (a, op, args) in enumerate(ins):
        if op.startswith("s_cbranch") or op == "s_branch":
            m = re.search(r"<\+0x([0-9a-fA-F]+)>", args)
            if not m:
                continue
            t_abs = ins[0][0] + int(m.group(1), 16)
            if t_abs in addr and addr[t_abs] < i:
                lo, hi = addr[t_abs], i
                cnt = sum(1 for _, o, _ in ins[lo:hi + 1] if o.startswith(opc))
                if cnt:
                    found.append((cnt, lo, hi, hi - lo + 1))
    inner = [f for f in found if not any(g is not f and g[1] >= f[1] and g[2] <= f[2] for g in found)]
    return sorted(inner, key=lambda f: -f[0])


def hottest_loop(ins, opc):
    """innermost backward branch range with the most `opc`."""
    inner = loops_with(ins, opc)
    return inner[0] if inner else None
    addr = {a: i for i, (a, _, _) in enumerate(ins)}
    best = None
    for i, (a, op, args) in enumerate(ins):
        if op.startswith("s_cbranch") or op == "s_branch":
            m = re.search(r"<\+0x([0-9a-fA-F]+)>", args)
            if not m:
                continue
            t_abs = ins[0][0] + int(m.group(1), 16)     # objdump prints targets relative to the function's symbol
            if t_abs in addr and addr[t_abs] < i:
                lo, hi = addr[t_abs], i
                cnt = sum(1 for _, o, _ in ins[lo:hi + 1] if o.startswith(opc))
                size = hi - lo + 1
                if cnt and (best is None or cnt > best[0] or (cnt == best[0] and size < best[3])):
                    best = (cnt, lo, hi, size)
    return best


def main():
    obj, pat = sys.argv[1], sys.argv[2]
    loop = sys.argv[sys.argv.index("--loop") + 1] if "--loop" in sys.argv else None
    fns = functions(device_asm(obj))
    for name, body in fns.items():
        dn = demangle(name)
        if pat not in dn and pat not in name:
            continue
        ins = parse(body)
        scope = "whole kernel"
        if loop and "--list" in sys.argv:
            for c, lo, hi, size in loops_with(ins, loop):
                sub = collections.Counter(op for _, op, _ in ins[lo:hi + 1])
                cl = classify(sub)
                print(f"{dn[:60]}: innermost loop at +0x{ins[lo][0] - ins[0][0]:x}: {size} instructions, {c} x {loop}*; " +
                      ", ".join(f"{k} {v}" for k, v in cl.most_common(7)))
            continue
        if loop:
            b = hottest_loop(ins, loop)
            if b:
                ins = ins[b[1]:b[2] + 1]
                scope = f"hottest loop for {loop} ({b[3]} instructions)"
        ops = collections.Counter(op for _, op, _ in ins)
        cl = classify(ops)
        tot = sum(ops.values())
        print(f"{dn[:100]}: {tot} instructions, {scope}")
        for k, v in cl.most_common():
            print(f"    {k:28s} {v:7d}  {100.0 * v / tot:5.1f} %")


if __name__ == "__main__":
    main()
