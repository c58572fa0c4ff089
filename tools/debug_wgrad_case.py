#!/usr/bin/env python3
"""One weight-gradient launch vs torch on the GPU box (debugging aid): python tools/debug_wgrad_case.py N H W Ca Cb Co"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import explib  # noqa: E402

_lib = explib.use()
L = _lib.lib()
N, H, W, Ca, Cb, Co = (int(a) for a in sys.argv[1:7])
torch.manual_seed(0)
dev = torch.device("cuda:0")
Ci = Ca + Cb
xa, xb = torch.randn(N, Ca, H, W), torch.randn(N, max(Cb, 1), H, W)
dy = torch.randn(N, Co, H, W)
xin = torch.cat([xa, xb[:, :Cb]], 1)
w = torch.zeros(Co, Ci, 3, 3, requires_grad=True)
torch.nn.functional.conv2d(xin, w, padding=1).backward(dy)
xa_d, xb_d, dy_d = xa.to(dev), xb.to(dev), dy.to(dev)
sa, sb = _lib.WslSrc(), _lib.WslSrc()
sa.x, sa.bs, sa.C, sa.emask_scale = xa_d.data_ptr(), Ca * H * W, Ca, 1.0
sb.x, sb.bs, sb.C, sb.emask_scale = xb_d.data_ptr(), max(Cb, 1) * H * W, Cb, 1.0
L.wsl_conv2d_wgrad_ws_bytes.restype = C.c_size_t
wsb = L.wsl_conv2d_wgrad_ws_bytes(N, H, W, Ci, Co, 3)
ws = torch.zeros(wsb // 4 + 16, device=dev)
dw, db = torch.zeros(Co, Ci, 3, 3, device=dev), torch.zeros(Co, device=dev)
st = torch.cuda.current_stream().cuda_stream
_lib.check(L.wsl_conv2d_wgrad(C.byref(sa), C.byref(sb) if Cb else None, dy_d.data_ptr(), Co * H * W, dw.data_ptr(), db.data_ptr(), N, H, W,
                              Co, 3, ws.data_ptr(), C.c_size_t(wsb), st))
torch.cuda.synchronize()
err = (dw.cpu() - w.grad).abs()
print(f"lib={os.environ.get('WSL_EXP_LIB', 'new')} xcd={os.environ.get('WSL_WGRAD_XCD', '-')} max|err|={float(err.max()):.3e}  "
      f"db err={float((db.cpu() - dy.sum((0, 2, 3))).abs().max()):.3e}")
print("max err per tap:", [f"{float(err[:, :, i // 3, i % 3].max()):.2e}" for i in range(9)])
bad = (err > 1e-3).nonzero()
print("bad entries:", len(bad), bad[:12].tolist())
