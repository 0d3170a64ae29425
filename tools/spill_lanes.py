#!/usr/bin/env python
"""SGPR spill slots of a kernel that are READ but never WRITTEN: scalar registers live in lanes of reserved VGPRs
(v_writelane_b32 vN, sX, lane / v_readlane_b32 sX, vN, lane).  A read of a (vN, lane) pair no instruction of the kernel writes
returns whatever the register file holds - a non-deterministic scalar.  python tools/spill_lanes.py <object> <kernel substring>"""
import re, sys, collections
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
import isa_mix as im
asm = im.device_asm(sys.argv[1])
for name, body in im.functions(asm).items():
    d = im.demangle(name)
    if sys.argv[2] not in d:
        continue
    wr, rd = collections.Counter(), collections.Counter()
    for m in re.finditer(r"v_writelane_b32 (v\d+), \S+, (\d+)", body): wr[(m.group(1), int(m.group(2)))] += 1
    for m in re.finditer(r"v_readlane_b32 \S+, (v\d+), (\d+)", body): rd[(m.group(1), int(m.group(2)))] += 1
    # VGPRs also written as ordinary vector registers are not pure spill carriers: list them
    carriers = sorted({v for v, _ in wr} | {v for v, _ in rd}, key=lambda s: int(s[1:]))
    other = {v: len(re.findall(r"^\s+v_(?!writelane|readlane)\w+ %s\b" % v, body, re.M)) for v in carriers}
    missing = sorted(k for k in rd if k not in wr)
    print(d.replace("mi::", ""), "| spill slots written", len(wr), "read", len(rd), "| read but never written:", missing[:12], "| carriers", {v: other[v] for v in carriers})
