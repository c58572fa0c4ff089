#!/usr/bin/env python3
"""One line per 3x3 layer shape of the unet_cct step (batch 64): Winograd conv with a BatchNorm source (= forward launches),
with a plain source (= data-gradient launches) and the Winograd weight gradient, microseconds per launch (HIP events, 20
launches each).  Runs on the experiments build so the WSL_* knobs apply; WSL_EXP_LIB=old picks tools/exp/libwslhip_exp_old.so
(a build of another source revision) for A/B runs:
   WSL_EXP_LIB=old python tools/sweep_layers.py ; WSL_WGRAD_XCD=0 WSL_WINO_XCD_Y=0 python tools/sweep_layers.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import explib  # noqa: E402

_lib = explib.use()
L = _lib.lib()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
N = int(os.environ.get("SWEEP_N", "64"))
SHAPES = [(16, 16, 256), (32, 16, 256), (32, 32, 128), (64, 32, 128), (64, 64, 64), (128, 64, 64), (128, 128, 32),
          (256, 128, 32), (256, 256, 16)]
if os.environ.get("SWEEP_SHAPES"):   # e.g. SWEEP_SHAPES="64,64,64;128,64,64" (Ci,Co,S per layer): one layer under a counter pass
    SHAPES = [tuple(int(v) for v in t.split(",")) for t in os.environ["SWEEP_SHAPES"].split(";")]
REPS = int(os.environ.get("SWEEP_REPS", "20"))
L.wsl_conv2d_wgrad_ws_bytes.restype = C.c_size_t


def timed(fn, reps=REPS):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


tot = [0.0, 0.0, 0.0, 0.0]
print(f"lib={os.environ.get('WSL_EXP_LIB', 'new')} WSL_WGRAD_XCD={os.environ.get('WSL_WGRAD_XCD', '-')} "
      f"WSL_WINO_XCD_Y={os.environ.get('WSL_WINO_XCD_Y', '-')}")
print("| layer | conv, BN source (fwd) | conv, plain source (fwd partials) | wgrad (incl. reduce) | data gradient Co -> Ci with the BatchNorm-backward epilogue |")
print("|---|---|---|---|---|")
for Ci, Co, S in SHAPES:
    H = W = S
    x = torch.randn(N, Ci, H, W, device=dev)
    w = torch.randn(Co, Ci, 3, 3, device=dev) * 0.05
    y = torch.empty(N, Co, H, W, device=dev)
    dy = torch.randn(N, Co, H, W, device=dev)
    scale, shift = torch.rand(Ci, device=dev) + 0.5, torch.randn(Ci, device=dev) * 0.1
    wp = torch.empty(16 * Ci * Co, device=dev)
    _lib.check(L.wsl_conv2d_pack_weights(w.data_ptr(), wp.data_ptr(), Co, Ci, 3, 2, st))
    nblk = L.wsl_conv2d_stat_blocks(N, H, W, Ci, Co, 3)
    part, cnt = torch.zeros(max(nblk * Co * 2, nblk * 64), device=dev), torch.empty(nblk, device=dev)
    res = []
    for raw in (0, 1):
        s = _lib.WslSrc()
        s.x, s.bs, s.C, s.emask_scale = x.data_ptr(), Ci * H * W, Ci, 1.0
        if not raw:
            s.scale, s.shift = scale.data_ptr(), shift.data_ptr()
        res.append(timed(lambda: _lib.check(L.wsl_conv2d_fwd(C.byref(s), None, wp.data_ptr(), None, y.data_ptr(), Co * H * W, N, H,
                                                             W, Co, 3, 4, part.data_ptr(), cnt.data_ptr(), st))))
    s = _lib.WslSrc()
    s.x, s.bs, s.C, s.scale, s.shift, s.emask_scale = x.data_ptr(), Ci * H * W, Ci, scale.data_ptr(), shift.data_ptr(), 1.0
    dw, db = torch.empty(Co, Ci, 3, 3, device=dev), torch.empty(Co, device=dev)
    wsb = L.wsl_conv2d_wgrad_ws_bytes(N, H, W, Ci, Co, 3)
    ws = torch.empty(wsb // 4 + 16, device=dev)
    res.append(timed(lambda: _lib.check(L.wsl_conv2d_wgrad(C.byref(s), None, dy.data_ptr(), Co * H * W, dw.data_ptr(), db.data_ptr(),
                                                           N, H, W, Co, 3, ws.data_ptr(), C.c_size_t(wsb), st))))
    # a REAL data-gradient launch of the layer (Co -> Ci, Winograd data-gradient image) with the BatchNorm-backward statistics of the
    # consumer layer in its epilogue (y, keep mask, coefficients): what the network's backward launches -- column 2 is the same kernel
    # with the forward's BatchNorm partials instead
    wd = torch.empty(16 * Ci * Co, device=dev)
    _lib.check(L.wsl_conv2d_pack_weights(w.data_ptr(), wd.data_ptr(), Co, Ci, 3, 3, st))
    sd = _lib.WslSrc()
    sd.x, sd.bs, sd.C, sd.emask_scale = dy.data_ptr(), Co * H * W, Co, 1.0
    g = torch.empty(N, Ci, H, W, device=dev)
    bn_st = torch.cat([torch.zeros(Ci, device=dev), torch.ones(Ci, device=dev), scale, shift]).contiguous()
    em = (torch.rand(N, Ci, H, W, device=dev) > 0.05).to(torch.uint8)
    nb2 = L.wsl_conv2d_stat_blocks(N, H, W, Co, Ci, 3)
    bnws = torch.zeros(max(nb2 * Ci * 2, 64) + 64, device=dev)
    fused = C.c_int(0)
    if L.wsl_conv2d_wino_ok(N, H, W, Co, 0, Ci, 3):
        res.append(timed(lambda: _lib.check(L.wsl_conv2d_dgrad_bn(C.byref(sd), wd.data_ptr(), g.data_ptr(), Ci * H * W, N, H, W, Ci, 3, 5, x.data_ptr(),
                                                                  bn_st.data_ptr(), em.data_ptr(), 1.0 / 0.95, bnws.data_ptr(), C.byref(fused), st))))
    else:
        res.append(float("nan"))
    for i in range(4):
        tot[i] += res[i]
    print(f"| {Ci}->{Co} @{S} | {res[0]:.1f} | {res[1]:.1f} | {res[2]:.1f} | {res[3]:.1f}{'' if fused.value else ' (statistics not fused)'} |")
    del x, w, y, dy, wp, part, cnt, ws, wd, g, em, bnws
print(f"| sum | {tot[0]:.1f} | {tot[1]:.1f} | {tot[2]:.1f} | {tot[3]:.1f} |")
