#!/usr/bin/env python3
"""Aggregates two rocprofv3 counter passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE; never combined with other trace domains)
of `bench.py --steps 2 --warmup 1` into HBM bytes per launch of the MFMA kernel families -> profiles/<tag>_pmc_traffic.json.
   python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json>
FETCH_SIZE / WRITE_SIZE are reported in KiB; FETCH_SIZE is doubled on gfx950 as MI355X_MICROARCH.md (HBM section)
prescribes for wide coalesced reads; WRITE_SIZE is taken as is (uncalibrated there)."""
import collections
import csv
import glob
import json
import sys


def per_kernel(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    assert f, "no counter_collection.csv under " + d
    acc = collections.defaultdict(lambda: [0, 0.0])
    seen = set()
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        k = k.replace("conv_wino2r_kernel", "conv_wino2_kernel")   # the raw-source (LDS-DMA) variant rides in the same family
        fam = next((f for f in ("conv_sp_kernel", "wgrad_sp_kernel", "conv_wino2_kernel", "wgrad_wino_kernel", "conv_wino_kernel", "conv_mfma2l_kernel", "wgrad_mfma2s_kernel", "wgrad_mfma2l_kernel",
                                "conv_mfma2_kernel", "conv_nk16_kernel", "wgrad_mfma2_kernel") if f in k), None)
        if fam is None:
            continue
        key = (r["Dispatch_Id"], fam)
        acc[fam][1] += float(r["Counter_Value"])      # one row per (dispatch, XCD/instance): sum them
        if key not in seen:
            seen.add(key)
            acc[fam][0] += 1
    return acc


def aggregate(fetch_dir, write_dir):
    import time
    fetch, write = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    out = {"collected_utc": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()), "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `bench.py --steps 2 --warmup 1`; KiB -> bytes, "
                     "FETCH_SIZE x2 per MI355X_MICROARCH.md (HBM section)", "kernels": {}}
    for fam in sorted(set(fetch) | set(write)):
        n = fetch[fam][0] or write[fam][0]
        fb = 2.0 * 1024.0 * fetch[fam][1] / max(fetch[fam][0], 1)
        wb = 1024.0 * write[fam][1] / max(write[fam][0], 1)
        out["kernels"][fam] = {"launches_sampled": n, "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
                               "hbm_bytes_per_launch": fb + wb}
    return out


if __name__ == "__main__":
    out = aggregate(sys.argv[1], sys.argv[2])
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out, indent=1))
