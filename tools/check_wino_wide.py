#!/usr/bin/env python3
"""Numerics of the wide-tile Winograd instantiations (experiments build, WSL_WINO_WIDE) against torch's conv2d on the GPU:
   WSL_WINO_WIDE=3 python tools/check_wino_wide.py"""
import ctypes as C
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import explib  # noqa: E402

_lib = explib.use()
L = _lib.lib()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
for Ci, Co, S, bn in ((16, 16, 256, True), (32, 16, 256, False), (32, 32, 128, True), (64, 32, 128, False), (64, 64, 64, True)):
    N = 4
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, Ci, S, S, generator=g).to(dev)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) * 0.1).to(dev)
    sc, sh = (torch.rand(Ci, generator=g) + 0.5).to(dev), (torch.randn(Ci, generator=g) * 0.3).to(dev)
    s = _lib.WslSrc()
    s.x, s.bs, s.C, s.emask_scale = x.data_ptr(), Ci * S * S, Ci, 1.0
    if bn:
        s.scale, s.shift = sc.data_ptr(), sh.data_ptr()
    u = torch.empty(16 * Ci * Co + 16, device=dev)
    _lib.check(L.wsl_conv2d_pack_weights(w.data_ptr(), u.data_ptr(), Co, Ci, 3, 2, st))
    y = torch.zeros(N, Co, S, S, device=dev)
    nb = L.wsl_conv2d_stat_blocks(N, S, S, Ci, Co, 3)
    sp, scnt = torch.zeros(Co * nb * 2, device=dev), torch.zeros(nb, device=dev)
    _lib.check(L.wsl_conv2d_fwd(C.byref(s), None, u.data_ptr(), None, y.data_ptr(), Co * S * S, N, S, S, Co, 3, 4, sp.data_ptr(), scnt.data_ptr(), st))
    v = F.leaky_relu(x * sc[None, :, None, None] + sh[None, :, None, None], 0.01) if bn else x
    ref = F.conv2d(v.double(), w.double(), padding=1).float()
    err = float((y - ref).abs().max() / ref.abs().max())
    part = sp.view(Co, nb, 2)
    mean = part[:, :, 0].sum(1) / (N * S * S)
    merr = float((mean - ref.mean((0, 2, 3))).abs().max() / ref.abs().max())
    print(f"{Ci}->{Co} @{S} {'bn' if bn else 'raw'} WSL_WINO_WIDE={os.environ.get('WSL_WINO_WIDE', '0')}: max rel err {err:.2e}, channel mean from the partials {merr:.2e}, cnt sum {float(scnt.sum()):.0f} of {N * S * S}")
