#!/usr/bin/env python3
"""Times the weight gradient of one conv layer through the C ABI with HIP events on the launch stream.
   python tools/microbench_wgrad.py N Ci Co H W [ks]      (WSL_WGRAD_ABLATE=1|2|8 for phase ablations)"""
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
dy = torch.randn(N, Co, H, W, device=dev)
dw, db = torch.empty(Co, Ci, ks, ks, device=dev), torch.empty(Co, device=dev)
scale, shift = torch.rand(Ci, device=dev) + 0.5, torch.randn(Ci, device=dev) * 0.1
s = _lib.WslSrc()
s.x, s.bs, s.C, s.scale, s.shift, s.emask_scale = x.data_ptr(), Ci * H * W, Ci, scale.data_ptr(), shift.data_ptr(), 1.0
L.wsl_conv2d_wgrad_ws_bytes.restype = C.c_size_t
wsb = L.wsl_conv2d_wgrad_ws_bytes(N, H, W, Ci, Co, ks)
ws = torch.empty(wsb // 4 + 16, device=dev)
st = torch.cuda.current_stream().cuda_stream


def run():
    _lib.check(L.wsl_conv2d_wgrad(C.byref(s), None, dy.data_ptr(), Co * H * W, dw.data_ptr(), db.data_ptr(), N, H, W, Co, ks,
                                  ws.data_ptr(), C.c_size_t(wsb), st))


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
print(f"wgrad ablate={os.environ.get('WSL_WGRAD_ABLATE','0')} N={N} {Ci}->{Co} {H}x{W} k{ks}: {us:8.1f} us (incl. reduce)  "
      f"{fl/us/1e6:7.1f} TFLOP/s  in+dy {4.0*N*H*W*(Ci+Co)/us/1e3:7.1f} GB/s")
