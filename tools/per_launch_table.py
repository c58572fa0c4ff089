#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV of a few serialised bench steps -> one row per kernel launch of ONE step (name, grid, duration),
in launch order: which launches of the HBM-side helper kernels sit below the roofline, level by level.
   python tools/per_launch_table.py <dir with *kernel_trace.csv> <steps in the trace> <out.md>"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
steps = int(sys.argv[2])
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
if "--all" not in sys.argv:
    rows = [r for r in rows if "wsl::" in r["Kernel_Name"]]
per = len(rows) // steps
last = rows[-per:]
t0 = int(last[0]["Start_Timestamp"])
out = ["| # | kernel | grid | us | start us |", "|---|---|---|---|---|"]
tot = 0.0
for i, r in enumerate(last):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    name = r["Kernel_Name"].replace("void wsl::", "").replace("wsl::", "")
    name = name.split("(")[0][:60]
    out.append(f"| {i} | {name} | {r.get('Grid_Size_X', r.get('Grid_Size', '?'))}x{r.get('Grid_Size_Y', '')}x{r.get('Grid_Size_Z', '')} | {d:.1f} | {(int(r['Start_Timestamp']) - t0) / 1e3:.0f} | q{r.get('Queue_Id', '?')} |")
span = (int(last[-1]["End_Timestamp"]) - t0) / 1e3
out.append(f"\n{per} launches per step, kernel time {tot:.0f} us, span {span:.0f} us (gaps {span - tot:.0f} us)")
open(sys.argv[3], "w").write("\n".join(out) + "\n")
print(out[-1])
