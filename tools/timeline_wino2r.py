#!/usr/bin/env python3
"""Per-workgroup timeline of conv_wino2r_kernel (experiments build, WSL_CONV_ABLATE=128): how long a workgroup spends in its
prologue (entry -> first chunk landed), channel loop and epilogue, how long a hardware slot stays empty between two workgroups,
and how many workgroups of a CU are in which phase at the same time.
   python tools/timeline_wino2r.py N Ci Co H W"""
import collections
import ctypes as C
import os
import sys

os.environ["WSL_CONV_ABLATE"] = "128"
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import explib  # noqa: E402

_lib = explib.use()
L = _lib.lib()
N, Ci, Co, H, W = (int(a) for a in sys.argv[1:6])
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
x = torch.randn(N, Ci, H, W, device=dev)
w = torch.randn(Co, Ci, 3, 3, device=dev) * 0.05
y = torch.empty(N, Co, H, W, device=dev)
wp = torch.empty(16 * Ci * Co, device=dev)
_lib.check(L.wsl_conv2d_pack_weights(w.data_ptr(), wp.data_ptr(), Co, Ci, 3, 2, st))
nblk = L.wsl_conv2d_stat_blocks(N, H, W, Ci, Co, 3)
part, cnt = torch.zeros(max(nblk * Co * 2, nblk * 64), device=dev), torch.empty(nblk, device=dev)
s = _lib.WslSrc()
s.x, s.bs, s.C, s.emask_scale = x.data_ptr(), Ci * H * W, Ci, 1.0
for _ in range(3):
    _lib.check(L.wsl_conv2d_fwd(C.byref(s), None, wp.data_ptr(), None, y.data_ptr(), Co * H * W, N, H, W, Co, 3, 4, part.data_ptr(), cnt.data_ptr(), st))
part.zero_()
_lib.check(L.wsl_conv2d_fwd(C.byref(s), None, wp.data_ptr(), None, y.data_ptr(), Co * H * W, N, H, W, Co, 3, 4, part.data_ptr(), cnt.data_ptr(), st))
torch.cuda.synchronize()
t = part.view(torch.int64).cpu().numpy()
co_t = 32 if Co % 32 == 0 else 16
nwg = nblk * (Co // co_t)
t = t[:8 * nwg].reshape(nwg, 8)
t = t[t[:, 0] > 0]
ticks_per_us = ((t[:, 3] - t[:, 0]) / np.maximum(1, (t[:, 5] - t[:, 4]) / 100.0))
mhz = np.median(ticks_per_us)
span = (t[:, 5].max() - t[:, 4].min()) / 100.0
print(f"{len(t)} workgroups stamped of {nwg}; shader clock ~{mhz:.0f} MHz; kernel span {span:.1f} us")
d = np.diff(t[:, :4], axis=1) / mhz
for nm, col in zip(("prologue (entry -> first chunk landed)", "channel loop", "epilogue (output transform, stores, statistics)"), d.T):
    print(f"   {nm:48s} mean {col.mean():6.2f} us   p10 {np.percentile(col, 10):6.2f}   p90 {np.percentile(col, 90):6.2f}")
print(f"   {'workgroup lifetime':48s} mean {((t[:, 3] - t[:, 0]) / mhz).mean():6.2f} us")
# slots: (xcc, se, sh?, cu, simd, wave slot) from HW_ID [3:0] wave, [5:4] simd, [11:8] cu, [12] sh, [15:13] se
hw, xcc = t[:, 6], t[:, 7] & 0xf
key = (xcc << 20) | (hw & 0xffff)
gaps, per_cu = [], collections.defaultdict(list)
for k in np.unique(key):
    rows = t[key == k]
    rows = rows[np.argsort(rows[:, 4])]
    if len(rows) > 1:
        gaps += list((rows[1:, 4] - rows[:-1, 5]) / 100.0)
    cu = k & ~0x3f          # same CU: drop the wave slot and SIMD bits
    per_cu[cu] += [(r[4] / 100.0, r[5] / 100.0) for r in rows]
gaps = np.array(gaps) if gaps else np.zeros(1)
print(f"   hardware slots seen: {len(np.unique(key))}; empty time between two workgroups of a slot (real-time counter): mean {gaps.mean():.2f} us, "
      f"p50 {np.median(gaps):.2f}, p90 {np.percentile(gaps, 90):.2f}")
print(f"   rounds per slot: {len(t) / max(1, len(np.unique(key))):.1f}")
