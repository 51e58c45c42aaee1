#!/usr/bin/env python3
"""Print the slowest dispatches of a rocprofv3 kernel trace (last step) - which launches dominate a step."""
import csv, glob, os, sys
d = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 800
p = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-n_last:]
t0 = int(last[0]["Start_Timestamp"])
out = []
for r in last:
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    blocks = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))) * max(1, int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"]))) * max(1, int(r["Grid_Size_Z"]))
    out.append((dur, blocks, r["Kernel_Name"][:45], (int(r["Start_Timestamp"]) - t0) / 1e3))
span = (int(last[-1]["End_Timestamp"]) - t0) / 1e3
print("span of last %d dispatches: %.1f us, sum of durations %.1f us" % (n_last, span, sum(o[0] for o in out)))
for o in sorted(out, reverse=True)[:40]:
    print("%8.1f us  blocks %6d  %-45s at %9.1f" % o)
