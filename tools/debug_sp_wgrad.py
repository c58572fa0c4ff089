#!/usr/bin/env python3
"""Where a split-precision weight gradient deviates from an fp64 reference (GPU), and whether repeated launches agree bit for bit.
   python tools/debug_sp_wgrad.py N H W Ci Co [bn] [reps]"""
import ctypes as C
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("WSL_LIB"):
    from wsl4mis_amd import _lib  # noqa: E402
    _lib.LIB_PATH = os.environ["WSL_LIB"]
else:
    from wsl4mis_amd import _lib  # noqa: E402

L = _lib.lib()
N, H, W, Ci, Co = (int(v) for v in sys.argv[1:6])
bn = len(sys.argv) > 6 and sys.argv[6] in ("bn", "pre")
pre = len(sys.argv) > 6 and sys.argv[6] == "pre"      # the BatchNorm-ed activations handed over as a plain source: same DATA, no loader transform
reps = int(sys.argv[-1]) if sys.argv[-1].isdigit() and len(sys.argv) > 6 else 4
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(7)
x = torch.randn(N, Ci, H, W, generator=g)
dy = torch.randn(N, Co, H, W, generator=g) * 3e-5
scale, shift = torch.rand(Ci, generator=g) + 0.5, torch.randn(Ci, generator=g) * 0.3
vact = F.leaky_relu(x * scale[None, :, None, None] + shift[None, :, None, None], 0.01) if bn else x
torch.set_num_threads(min(32, torch.get_num_threads()))
ref = torch.nn.grad.conv2d_weight(vact.double(), (Co, Ci, 3, 3), dy.double(), padding=1)
refb = dy.double().sum((0, 2, 3))
xd, dyd, sd, hd = (vact if pre else x).to(dev), dy.to(dev), scale.to(dev), shift.to(dev)
s = _lib.WslSrc()
s.x, s.bs, s.C, s.emask_scale = xd.data_ptr(), Ci * H * W, Ci, 1.0
if bn and not pre:
    s.scale, s.shift = sd.data_ptr(), hd.data_ptr()
dymax = torch.zeros(64, dtype=torch.int32, device=dev)
dymax[37] = int(dyd.abs().max().view(torch.int32))
wsb = L.wsl_sp_conv2d_wgrad_ws_bytes(N, H, W, Ci, Co)
ws = torch.empty(wsb // 4 + 16, device=dev)
outs = []
for r in range(reps):
    dw, db = torch.zeros(Co, Ci, 3, 3, device=dev), torch.zeros(Co, device=dev)
    pend = _lib.WslWgradPending()
    _lib.check(L.wsl_sp_conv2d_wgrad_partial(C.byref(s), None, dyd.data_ptr(), Co * H * W, dymax.data_ptr(), dw.data_ptr(), db.data_ptr(),
                                             N, H, W, Co, ws.data_ptr(), C.c_size_t(wsb), C.byref(pend), st))
    _lib.check(L.wsl_wgrad_reduce_batch(C.byref(pend), 1, st))
    torch.cuda.synchronize()
    outs.append((dw.cpu(), db.cpu()))
tol = 1e-4 * float(ref.abs().max())
nbad = 0
for r, (dw, db) in enumerate(outs):
    err = (dw.double() - ref).abs()
    bad = err > tol
    nbad += int(bad.sum())
    same = bool((dw == outs[0][0]).all())
    print(f"run {r}: max err {float(err.max()):.3e} (tol {tol:.3e}, typical {float(err.median()):.1e}), bad entries {int(bad.sum())} of {bad.numel()}, "
          f"db max err {float((db.double() - refb).abs().max()):.2e} of {float(refb.abs().max()):.2e}; bit-equal to run 0: {same}")
    if bad.any() and r == 0:
        idx = bad.nonzero()
        print("  co:", torch.unique(idx[:, 0]).tolist()[:64])
        print("  ci:", torch.unique(idx[:, 1]).tolist()[:64])
        print("  ky:", torch.unique(idx[:, 2]).tolist(), "kx:", torch.unique(idx[:, 3]).tolist())
# the f32 kernel at the same criterion
wsb2 = L.wsl_conv2d_wgrad_ws_bytes(N, H, W, Ci, Co, 3)
ws2 = torch.empty(wsb2 // 4 + 16, device=dev)
dw, db = torch.zeros(Co, Ci, 3, 3, device=dev), torch.zeros(Co, device=dev)
_lib.check(L.wsl_conv2d_wgrad(C.byref(s), None, dyd.data_ptr(), Co * H * W, dw.data_ptr(), db.data_ptr(), N, H, W, Co, 3, ws2.data_ptr(),
                              C.c_size_t(wsb2), st))
torch.cuda.synchronize()
err = (dw.cpu().double() - ref).abs()
print(f"f32 kernel: max err {float(err.max()):.3e}, typical {float(err.median()):.1e}")
print("bad elements", nbad, "of all runs")
