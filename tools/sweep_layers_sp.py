#!/usr/bin/env python3
"""Split-precision kernels next to the f32 kernels, one line per 3x3 layer shape of the unet_cct step (batch 64; decoder
shapes with --dec): conv with a BatchNorm source (= forward launches), with a plain source (= data-gradient launches), weight
gradient (+ its share of the batched second stage), microseconds per launch (HIP events, 20 launches each), product library.
   python tools/sweep_layers_sp.py [--dec] [--only-sp]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--exp" in sys.argv:          # the experiments build: WSL_SP_ABLATE and friends apply
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import explib  # noqa: E402
    _lib = explib.use()
else:
    from wsl4mis_amd import _lib  # noqa: E402
    if os.environ.get("WSL_LIB"):          # any other build of the library (A / B timing)
        _lib.LIB_PATH = os.environ["WSL_LIB"]

L = _lib.lib()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
N = int(os.environ.get("SWEEP_N", "64"))
ENC = [(16, 16, 256), (16, 32, 128), (32, 32, 128), (32, 64, 64), (64, 64, 64), (64, 128, 32), (128, 128, 32), (128, 256, 16),
       (256, 256, 16)]
DEC = [(256, 128, 32), (128, 64, 64), (64, 32, 128), (32, 16, 256)]
SHAPES = ENC + (DEC if "--dec" in sys.argv else [])
if "--few" in sys.argv:
    SHAPES = [(16, 16, 256), (32, 16, 256), (64, 64, 64), (128, 128, 32)]
if "--mid" in sys.argv:
    SHAPES = [(32, 32, 128), (64, 64, 64), (128, 128, 32), (128, 64, 64), (256, 256, 16)]
REPS = int(os.environ.get("SWEEP_REPS", "20"))
only_sp = "--only-sp" in sys.argv


def timed(fn, reps=None):
    """microseconds per launch: the best of SWEEP_BEST (default 1) rounds of `reps` back-to-back launches"""
    reps = reps or REPS
    for _ in range(3):
        fn()
    best = float("inf")
    for _ in range(int(os.environ.get("SWEEP_BEST", "1"))):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


def src(x, Cn, HW, scale=None, shift=None):
    s = _lib.WslSrc()
    s.x, s.bs, s.C, s.emask_scale = x.data_ptr(), Cn * HW, Cn, 1.0
    if scale is not None:
        s.scale, s.shift = scale.data_ptr(), shift.data_ptr()
    return s


tot = [0.0] * 6
print("| layer | f32 fwd | sp fwd | f32 dgrad | sp dgrad | f32 wgrad | sp wgrad | sp fwd TF | sp fwd TB/s |")
print("|---|---|---|---|---|---|---|---|---|")
for Ci, Co, S in SHAPES:
    H = W = S
    x = torch.randn(N, Ci, H, W, device=dev)
    w = torch.randn(Co, Ci, 3, 3, device=dev) * 0.05
    y = torch.empty(N, Co, H, W, device=dev)
    dy = torch.randn(N, Co, H, W, device=dev) * 1e-4
    dx = torch.empty(N, Ci, H, W, device=dev)
    scale, shift = torch.rand(Ci, device=dev) + 0.5, torch.randn(Ci, device=dev) * 0.1
    res = [0.0] * 6
    sb = src(x, Ci, H * W, scale, shift)
    sdy = src(dy, Co, H * W)
    dw, db = torch.empty(Co, Ci, 3, 3, device=dev), torch.empty(Co, device=dev)
    if not only_sp:
        wp, wpd = torch.empty(16 * Ci * Co, device=dev), torch.empty(16 * Ci * Co, device=dev)
        _lib.check(L.wsl_conv2d_pack_weights(w.data_ptr(), wp.data_ptr(), Co, Ci, 3, 2, st))
        _lib.check(L.wsl_conv2d_pack_weights(w.data_ptr(), wpd.data_ptr(), Ci, Co, 3, 3, st))
        nblk = L.wsl_conv2d_stat_blocks(N, H, W, Ci, Co, 3)
        part, cnt = torch.zeros(max(nblk * Co * 2, nblk * 64), device=dev), torch.empty(nblk, device=dev)
        res[0] = timed(lambda: _lib.check(L.wsl_conv2d_fwd(C.byref(sb), None, wp.data_ptr(), None, y.data_ptr(), Co * H * W, N, H, W, Co,
                                                           3, 4, part.data_ptr(), cnt.data_ptr(), st)))
        res[2] = timed(lambda: _lib.check(L.wsl_conv2d_fwd(C.byref(sdy), None, wpd.data_ptr(), None, dx.data_ptr(), Ci * H * W, N, H, W,
                                                           Ci, 3, 5, None, None, st)))
        wsb = L.wsl_conv2d_wgrad_ws_bytes(N, H, W, Ci, Co, 3)
        ws = torch.empty(wsb // 4 + 16, device=dev)
        res[4] = timed(lambda: _lib.check(L.wsl_conv2d_wgrad(C.byref(sb), None, dy.data_ptr(), Co * H * W, dw.data_ptr(), db.data_ptr(),
                                                             N, H, W, Co, 3, ws.data_ptr(), C.c_size_t(wsb), st)))
    # ---- split path
    img = torch.empty(10 * Ci * Co + 16, device=dev)
    imgd = torch.empty(10 * Ci * Co + 16, device=dev)
    wmax, wmaxd = torch.zeros(4, dtype=torch.int64, device=dev), torch.zeros(4, dtype=torch.int64, device=dev)
    dymax = torch.zeros(4, dtype=torch.int32, device=dev)
    dymax[0] = int(dy.abs().max().view(torch.int32))
    _lib.check(L.wsl_sp_pack_weights(w.data_ptr(), img.data_ptr(), wmax.data_ptr(), Co, Ci, 0, st))
    _lib.check(L.wsl_sp_pack_weights(w.data_ptr(), imgd.data_ptr(), wmaxd.data_ptr(), Ci, Co, 1, st))
    nblk = L.wsl_sp_conv2d_stat_blocks(N, H, W, Ci, Co)
    part2, cnt2 = torch.zeros(nblk * Co * 2, device=dev), torch.empty(nblk, device=dev)
    res[1] = timed(lambda: _lib.check(L.wsl_sp_conv2d_fwd(C.byref(sb), None, img.data_ptr(), wmax.data_ptr(), None, None, y.data_ptr(),
                                                          Co * H * W, N, H, W, Co, part2.data_ptr(), cnt2.data_ptr(), st)))
    res[3] = timed(lambda: _lib.check(L.wsl_sp_conv2d_fwd(C.byref(sdy), None, imgd.data_ptr(), wmaxd.data_ptr(), dymax.data_ptr(), None,
                                                          dx.data_ptr(), Ci * H * W, N, H, W, Ci, None, None, st)))
    wsb2 = L.wsl_sp_conv2d_wgrad_ws_bytes(N, H, W, Ci, Co)
    ws2 = torch.empty(wsb2 // 4 + 16, device=dev)
    pend = _lib.WslWgradPending()

    def sp_wgrad():
        _lib.check(L.wsl_sp_conv2d_wgrad_partial(C.byref(sb), None, dy.data_ptr(), Co * H * W, dymax.data_ptr(), dw.data_ptr(),
                                                 db.data_ptr(), N, H, W, Co, ws2.data_ptr(), C.c_size_t(wsb2), C.byref(pend), st))
        _lib.check(L.wsl_wgrad_reduce_batch(C.byref(pend), 1, st))
    res[5] = timed(sp_wgrad)
    for i in range(6):
        tot[i] += res[i]
    fl = 2.0 * N * H * W * Ci * Co * 9
    by = 4.0 * N * H * W * (Ci + Co)
    print(f"| {Ci}->{Co} @{S} | {res[0]:.1f} | {res[1]:.1f} | {res[2]:.1f} | {res[3]:.1f} | {res[4]:.1f} | {res[5]:.1f} | "
          f"{fl / res[1] / 1e6:.0f} | {by / res[1] / 1e6:.2f} |", flush=True)
    del x, w, y, dy, dx
print("| sum | " + " | ".join(f"{t:.1f}" for t in tot) + " | | |")
