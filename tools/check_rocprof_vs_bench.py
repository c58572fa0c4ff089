#!/usr/bin/env python3
"""Cross-check: average launch duration of the conv kernel family in a rocprofv3 kernel_stats.csv vs bench.py's roofline.
   python tools/check_rocprof_vs_bench.py <kernel_stats.csv> <bench.json>"""
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r = d["roofline"]
# rocprofv3 kernel names that bench.py's event families cover (the small first-conv / classifier kernels ride in the conv
# and weight-gradient families of the library's event bracketing)
NAMES = {"conv_mfma2l_kernel": ("conv_mfma2l_kernel", "conv_mfma2_kernel", "conv_cls_kernel", "conv_nk16_kernel"),
         "conv_wino2_kernel": ("conv_wino2_kernel", "conv_wino2r_kernel"),
         "wgrad_wino_kernel": ("wgrad_wino_kernel",),
         "wgrad_direct_kernels": ("wgrad_mfma2s_kernel", "wgrad_mfma2l_kernel", "wgrad_mfma2_kernel", "wgrad_small_kernel"),
         "conv_sp_kernel": ("conv_sp_kernel",),
         "wgrad_sp_kernel": ("wgrad_sp_kernel",),
         "wgrad_mfma2s_kernel": ("wgrad_mfma2s_kernel", "wgrad_mfma2l_kernel", "wgrad_mfma2_kernel", "wgrad_small_kernel")}
for kname, k in r.get("kernels", {r["kernel"]: r}).items():
    fam = [x for x in rows if any(n in x["Name"] for n in NAMES[kname.split()[0]])]
    t, c = sum(int(x["TotalDurationNs"]) for x in fam), sum(int(x["Calls"]) for x in fam)
    print(f"{kname.split()[0]:22s} rocprofv3: {c} launches, avg {t / max(c, 1) / 1e3:.2f} us | bench.py events: {k['launches']} launches, "
          f"avg {k['avg_launch_us']} us, {k['achieved']} TFLOP/s")
print(f"dominant: {r['kernel']} | value {d['value']} slices/s")
