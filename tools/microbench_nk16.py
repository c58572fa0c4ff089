#!/usr/bin/env python3
"""First convolution (1 -> 16 forward, with BatchNorm partials) and classifier data gradient (4 -> 16) at the benchmark size,
microseconds per launch (experiments build: WSL_CONV_NK16=0 = the generic kernel)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("WSL_LIB"):          # any other build of the library (A / B timing)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from wsl4mis_amd import _lib  # noqa: E402
    _lib.LIB_PATH = os.environ["WSL_LIB"]
else:
    import explib  # noqa: E402
    _lib = explib.use()
L = _lib.lib()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
N, H, W = 64, 256, 256
for Ci, dgrad in ((1, False), (4, True)):
    x = torch.randn(N, Ci, H, W, device=dev)
    w = torch.randn(16, Ci, 3, 3, device=dev) if not dgrad else torch.randn(Ci, 16, 3, 3, device=dev)
    wp = torch.empty(9 * Ci * 16, device=dev)
    if dgrad:
        _lib.check(L.wsl_conv2d_pack_weights(w.data_ptr(), wp.data_ptr(), Ci, 16, 3, 1, st))
    else:
        _lib.check(L.wsl_conv2d_pack_weights(w.data_ptr(), wp.data_ptr(), 16, Ci, 3, 0, st))
    y = torch.empty(N, 16, H, W, device=dev)
    s = _lib.WslSrc()
    s.x, s.bs, s.C, s.emask_scale = x.data_ptr(), Ci * H * W, Ci, 1.0
    nblk = L.wsl_conv2d_stat_blocks(N, H, W, Ci, 16, 3)
    part, cnt = torch.zeros(nblk * 32, device=dev), torch.zeros(nblk, device=dev)

    def run():
        _lib.check(L.wsl_conv2d_fwd(C.byref(s), None, wp.data_ptr(), None, y.data_ptr(), 16 * H * W, N, H, W, 16, 3, 3 if dgrad else 2,
                                    None if (dgrad or os.environ.get('MB_NOSTAT')) else part.data_ptr(),
                                    None if (dgrad or os.environ.get('MB_NOSTAT')) else cnt.data_ptr(), st))
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"NK16={os.environ.get('WSL_CONV_NK16', '1')} {Ci}->16 {'dgrad' if dgrad else 'fwd'}: {us:7.1f} us  "
          f"{4.0 * N * H * W * (16 + Ci) / us / 1e3:7.1f} GB/s")
