#!/usr/bin/env python3
"""Cross-check: average launch duration of the conv kernel family in a rocprofv3 kernel_stats.csv vs bench.py's roofline.
   python tools/check_rocprof_vs_bench.py <kernel_stats.csv> <bench.json>"""
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
fam = [r for r in rows if "conv_mfma2l_kernel" in r["Name"] or "conv_mfma2_kernel" in r["Name"]]
t, c = sum(int(r["TotalDurationNs"]) for r in fam), sum(int(r["Calls"]) for r in fam)
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r = d["roofline"]
print(f"rocprofv3: {c} launches, avg {t / c / 1e3:.2f} us | bench.py events: {r['launches']} launches, avg {r['avg_launch_us']} us, "
      f"{r['achieved']} TFLOP/s | value {d['value']} slices/s")
