"""Gaps between consecutive dispatches of a rocprofv3 kernel trace (csv): where a bench step's time goes
besides the solve kernel itself.   python tools/gap_trace.py <dir with *kernel_trace.csv>"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev_end, prev_name = None, None
out = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-40:]
    if prev_end is not None:
        out.append((prev_name, name, s - prev_end, e - s))
    prev_end, prev_name = e, name
for a, b, gap, dur in out[-48:]:
    print(f"{a:>42} -> {b:<42} gap {gap / 1e3:8.2f} us   dur {dur / 1e3:8.2f} us")
