#!/usr/bin/env python3
"""Per-kernel table of every SQ counter found under the given rocprofv3 --pmc output directories (one pass per directory), summed
over the dispatches of each kernel and divided by the kernel's SQ_WAVE_CYCLES of the same pass when the pass has it (else raw).
   python tools/pmc_sq_table.py <dir> [<dir> ...]"""
import collections
import csv
import glob
import re
import sys

per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))   # kernel -> pass -> counter
disp = collections.defaultdict(set)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "wsl::" not in k:
                continue
            k = re.sub(r"\(.*", "", k).replace("void wsl::", "")
            per[k][d][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[(k, d)].add(r["Dispatch_Id"])
for k in sorted(per):
    print(f"== {k}")
    for d, c in per[k].items():
        n = len(disp[(k, d)])
        wc = c.get("SQ_WAVE_CYCLES")
        cells = []
        for name, v in sorted(c.items()):
            cells.append(f"{name}={v / n:.4g}" + (f" ({v / wc:.3f} of wave cycles)" if wc and name != "SQ_WAVE_CYCLES" else ""))
        print(f"   [{n} dispatches] " + "; ".join(cells))
