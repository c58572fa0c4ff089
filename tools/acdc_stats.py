#!/usr/bin/env python3
"""Reads every ACDC slice / volume file of a reference checkout with wsl4mis_amd.dataloaders.h5lite and prints the label /
scribble statistics SURVEY 8d quotes (an independent decode of the same files: labelled share 1.06 %, classes
0.66 / 0.11 / 0.17 / 0.12 %), plus the scribble-vs-dense-label agreement.   python tools/acdc_stats.py <ACDC dir>"""
import glob
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util  # noqa: E402

spec = importlib.util.spec_from_file_location("h5lite", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                   "wsl4mis_amd", "dataloaders", "h5lite.py"))
h5 = importlib.util.module_from_spec(spec)
spec.loader.exec_module(h5)

base = sys.argv[1]
sl = sorted(glob.glob(os.path.join(base, "ACDC_training_slices", "*.h5")))
vo = sorted(glob.glob(os.path.join(base, "ACDC_training_volumes", "*.h5")))
cnt_s, cnt_l = np.zeros(5), np.zeros(4)
agree = tot_s = 0
px = 0
sizes = []
for p in sl:
    with h5.File(p) as f:
        img, lab, scr = f["image"][:], f["label"][:], f["scribble"][:]
    assert img.shape == lab.shape == scr.shape and img.dtype == np.float32 and img.min() >= 0 and img.max() <= 1.0
    cnt_s += np.bincount(scr.ravel(), minlength=5)[:5]
    cnt_l += np.bincount(lab.ravel(), minlength=4)[:4]
    m = scr != 4
    agree += int((scr[m] == lab[m]).sum())
    tot_s += int(m.sum())
    px += img.size
    sizes.append(img.size)
print(f"{len(sl)} slices, {px} px; native sizes {min(sizes)} .. {max(sizes)} px, median {int(np.median(sizes))}")
print("scribble shares %:", np.round(100 * cnt_s / px, 3), " labelled %.3f %%" % (100 * cnt_s[:4].sum() / px))
print("dense label shares %:", np.round(100 * cnt_l / px, 2))
print("scribble pixels agreeing with the dense label: %.2f %%" % (100.0 * agree / tot_s))
nv = 0
for p in vo:
    with h5.File(p) as f:
        img, lab = f["image"][:], f["label"][:]
    assert img.ndim == 3 and img.shape == lab.shape
    nv += img.shape[0]
print(f"{len(vo)} volumes, {nv} slices in total")
