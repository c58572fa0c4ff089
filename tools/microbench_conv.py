#!/usr/bin/env python3
"""Times one conv layer through the C ABI (packed fast path) with HIP events on the launch stream.
   python tools/microbench_conv.py N Ci Co H W [ks] [dgrad]      (WSL_CONV_ABLATE=1|2|4 for phase ablations;
   MB_WINO=1: the Winograd kernel (wmode 4) where the layer has a Winograd shape; MB_RAW=1: a plain source)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wsl4mis_amd import _lib  # noqa: E402

if os.environ.get("WSL_TOOLS_EXP", "1") != "0":   # the experiments build carries the knobs / probes these tools drive
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import explib  # noqa: E402
    explib.use()

N, Ci, Co, H, W = (int(a) for a in sys.argv[1:6])
ks = int(sys.argv[6]) if len(sys.argv) > 6 else 3
L = _lib.lib()
dev = torch.device("cuda:0")
x = torch.randn(N, Ci, H, W, device=dev)
w = torch.randn(Co, Ci, ks, ks, device=dev) * 0.05
wp = torch.empty(ks * ks * Ci * Co, device=dev)
y = torch.empty(N, Co, H, W, device=dev)
scale, shift = torch.rand(Ci, device=dev) + 0.5, torch.randn(Ci, device=dev) * 0.1
s = _lib.WslSrc()
s.x, s.bs, s.C, s.emask_scale = x.data_ptr(), Ci * H * W, Ci, 1.0
if not os.environ.get("MB_RAW"):          # MB_RAW=1: a plain source (what every data-gradient launch sees)
    s.scale, s.shift = scale.data_ptr(), shift.data_ptr()
nblk = L.wsl_conv2d_stat_blocks(N, H, W, Ci, Co, ks)
part, cnt = torch.zeros(max(nblk * Co * 2, nblk * 64), device=dev), torch.empty(nblk, device=dev)
st = torch.cuda.current_stream().cuda_stream
wino = bool(os.environ.get("MB_WINO")) and bool(L.wsl_conv2d_wino_ok(N, H, W, Ci, 0, Co, ks))
if wino:
    wp = torch.empty(16 * Ci * Co, device=dev)
_lib.check(L.wsl_conv2d_pack_weights(w.data_ptr(), wp.data_ptr(), Co, Ci, ks, 2 if wino else 0, st))


def run():
    _lib.check(L.wsl_conv2d_fwd(C.byref(s), None, wp.data_ptr(), None, y.data_ptr(), Co * H * W, N, H, W, Co, ks, 4 if wino else 2,
                                part.data_ptr(), cnt.data_ptr(), st))


for _ in range(3):
    run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
R = 20
for _ in range(R):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / R
fl = 2.0 * N * H * W * Co * Ci * ks * ks
print(f"{'wino ' if wino else 'direct '}ablate={os.environ.get('WSL_CONV_ABLATE','0')} N={N} {Ci}->{Co} {H}x{W} k{ks}: {us:8.1f} us  {fl/us/1e6:7.1f} TFLOP/s  "
      f"in+out {4.0*N*H*W*(Ci+Co)/us/1e3:7.1f} GB/s")

if int(os.environ.get('WSL_CONV_ABLATE', '0')) & 64:
    la = cnt.view(torch.int32).cpu().numpy().astype('uint32')
    hw = part.view(torch.int32).cpu().numpy().astype('uint32')[::Co * 2][:len(la)]
    import collections
    print('LDS_ALLOC values:', collections.Counter(hex(v) for v in la).most_common(8))
    print('HW_ID wave slots:', collections.Counter(int(v & 0xf) for v in hw).most_common(8))
    print('HW_ID simd:', collections.Counter(int((v >> 4) & 3) for v in hw).most_common(8))
    print('first 16 hw ids:', [hex(v) for v in hw[:16]])

if int(os.environ.get('WSL_CONV_ABLATE', '0')) & 128:
    import numpy as np
    t = part.view(torch.int64).cpu().numpy()[:nblk * 32].reshape(nblk, 32)
    ok = t[:, 0] > 0
    ids = np.where(ok)[0]
    t = t[ok]
    hw = t[:, 28]
    f = ((t[:, 26] - t[:, 0]) / ((t[:, 30] - t[:, 29]) / 100.0)).mean()   # shader-clock ticks per us
    rt0 = t[:, 29].min()
    print(f'{len(t)} workgroups stamped; shader clock {f:.0f} MHz; kernel span {(t[:, 30].max() - rt0) / 100.0:.1f} us')
    nch = min((Ci + 7) // 8, 4)
    names = ['prologue(issue0+tables)'] + sum([[f'c{c}:wait-data', f'c{c}:commit', f'c{c}:barrier', f'c{c}:issue', f'c{c}:mfma', f'c{c}:barrier2'] for c in range(nch)], []) + ['epilogue']
    if wino:
        names = ['prologue(issue0+tables)'] + sum([[f'c{c}:barrier+wait-data', f'c{c}:commit+issue', f'c{c}:barrier', f'c{c}:input transform', f'c{c}:barrier', f'c{c}:mfma'] for c in range(nch)], []) + ['barrier+out transform+stores']
    idx = [0, 1] + sum([[2 + 6 * c + j for j in range(6)] for c in range(nch)], []) + [26]
    d = np.diff(t[:, idx], axis=1)
    slot = hw & 0xf
    print('mean phase durations (shader cycles): all | by wave slot')
    for j, nm in enumerate(names):
        print(f'   {nm:28s} {d[:, j].mean():8.0f} | ' + ' '.join(f'{d[slot == sl, j].mean():8.0f}' for sl in sorted(set(slot))))
    print(f'   total per workgroup          {(t[:, 26] - t[:, 0]).mean():8.0f}')
