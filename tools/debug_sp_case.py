#!/usr/bin/env python3
"""Where a split-precision forward conv deviates from torch (GPU): error pattern per sample / channel block / tile.
   python tools/debug_sp_case.py N H W Ci Co [bn]"""
import ctypes as C
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("WSL_LIB"):          # any other build of the library (A / B hunting)
    from wsl4mis_amd import _lib  # noqa: E402
    _lib.LIB_PATH = os.environ["WSL_LIB"]
elif os.environ.get("WSL_USE_EXP"):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import explib  # noqa: E402
    _lib = explib.use()
else:
    from wsl4mis_amd import _lib  # noqa: E402

L = _lib.lib()
N, H, W, Ci, Co = (int(v) for v in sys.argv[1:6])
bn = len(sys.argv) > 6
pre = bn and sys.argv[6] == "pre"      # the BatchNorm-ed activations handed over as a plain source: same DATA, no loader transform
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(5)
x = torch.randn(N, Ci, H, W, generator=g)
w = torch.randn(Co, Ci, 3, 3, generator=g) * 0.07
scale, shift = torch.rand(Ci, generator=g) + 0.5, torch.randn(Ci, generator=g) * 0.3
vact = F.leaky_relu(x * scale[None, :, None, None] + shift[None, :, None, None], 0.01) if bn else x
ref = F.conv2d(vact, w, padding=1)
xd, wd, sd, hd = (vact if pre else x).to(dev), w.to(dev), scale.to(dev), shift.to(dev)
s = _lib.WslSrc()
s.x, s.bs, s.C, s.emask_scale = xd.data_ptr(), Ci * H * W, Ci, 1.0
if bn and not pre:
    s.scale, s.shift = sd.data_ptr(), hd.data_ptr()
img = torch.empty(10 * Ci * Co + 16, device=dev)
wmax = torch.zeros(4, dtype=torch.int64, device=dev)
_lib.check(L.wsl_sp_pack_weights(wd.data_ptr(), img.data_ptr(), wmax.data_ptr(), Co, Ci, 0, st))
y = torch.zeros(N, Co, H, W, device=dev)
_lib.check(L.wsl_sp_conv2d_fwd(C.byref(s), None, img.data_ptr(), wmax.data_ptr(), None, None, y.data_ptr(), Co * H * W, N, H, W, Co, None, None, st))
torch.cuda.synchronize()
err = (y.cpu() - ref).abs()
tol = 1e-4 * ref.abs().max()
bad = err > tol
print("max err", float(err.max()), "tol", float(tol), "bad elements", int(bad.sum()), "of", bad.numel())
if bad.any():
    idx = bad.nonzero()
    print("samples:", torch.unique(idx[:, 0]).tolist()[:40])
    print("channels:", torch.unique(idx[:, 1]).tolist()[:70])
    print("rows:", torch.unique(idx[:, 2]).tolist()[:70])
    print("cols:", torch.unique(idx[:, 3]).tolist()[:70])
    th, tw = 8, 32 if W % 32 == 0 else 16
    tiles = torch.unique(torch.stack([idx[:, 0], idx[:, 2] // th, idx[:, 3] // tw], 1), dim=0)
    print("bad (n, ty, tx) tiles:", len(tiles), tiles[:24].tolist())
if bad.any():
    # inside the first bad tile: which output pixels / channels
    n0, ty0, tx0 = tiles[0].tolist()
    e = err[n0, :, ty0 * th:(ty0 + 1) * th, tx0 * tw:(tx0 + 1) * tw]
    b = e > tol
    print("first bad tile", (n0, ty0, tx0), "bad channels", b.any(2).any(1).nonzero().flatten().tolist())
    print("bad rows in tile", b.any(0).any(1).nonzero().flatten().tolist(), "bad cols in tile", b.any(0).any(0).nonzero().flatten().tolist())
if hasattr(L, "wsl_debug_sp_counters"):
    out = (C.c_uint * 64)()
    L.wsl_debug_sp_counters(out, 0)
    print("readback mismatches hi[j=0..3]", list(out[0:4]), "lo[j=0..3]", list(out[4:8]), "checks", out[8], "after barrier", out[9], "after MFMA loop", out[10])
    print("lo j=0: wrote/read", [hex(v) for v in out[16:20]], "hi j=0:", [hex(v) for v in out[20:24]])
if bad.any():
    # per bad tile: which INPUT positions (staged row, quad) are implied: a wrong input pixel (r, x) spoils outputs (r-1..r+1, x-1..x+1)
    from collections import Counter
    cnt = Counter()
    for (n0, ty0, tx0) in tiles.tolist():
        b = (err[n0, :, ty0 * th:(ty0 + 1) * th, tx0 * tw:(tx0 + 1) * tw] > tol).any(0)      # [th, tw]
        rows = b.any(1).nonzero().flatten().tolist()
        cols = b.any(0).nonzero().flatten().tolist()
        # centre rows / cols of 3-wide groups
        rset = sorted(set(r for r in rows if (r - 1 in rows or r == 0) and (r + 1 in rows or r == th - 1)))
        cset = sorted(set(c for c in cols if (c - 1 in cols or c == 0) and (c + 1 in cols or c == tw - 1)))
        for r in rset:
            for c in cset:
                if b[max(r - 1, 0):r + 2, max(c - 1, 0):c + 2].all():
                    cnt[(r + 1, (c + 4) // 4, (c + 4) % 4)] += 1          # staged row, quad, pixel-of-quad
    print("implied wrong input (staged row, quad, px): count over bad tiles")
    for k, v in sorted(cnt.items()):
        print("   ", k, v, " task k(oct0) =", k[0] * 10 + k[1], " lane", (k[0] * 10 + k[1]) % 64, "wave", (k[0] * 10 + k[1]) // 64)
if bad.any() and cnt:
    # solve for the input perturbation at one implied position: err[co, oy, ox] = sum_ci w[co, ci, ky, kx] * delta[ci]
    n0, ty0, tx0 = tiles[0].tolist()
    b = (err[n0, :, ty0 * th:(ty0 + 1) * th, tx0 * tw:(tx0 + 1) * tw] > tol).any(0)
    done = 0
    for (sr, q, px), _ in sorted(cnt.items()):
        r, c = sr - 1, 4 * q + px - 4                      # tile coordinates of the input pixel
        if not (1 <= r < th - 1 and 1 <= c < tw - 1) or not b[r - 1:r + 2, c - 1:c + 2].all():
            continue
        Y, X = ty0 * th + r, tx0 * tw + c
        d = (y.cpu() - ref)[n0, :, Y - 1:Y + 2, X - 1:X + 2]          # [Co, 3, 3]: output (Y + oy - 1, X + ox - 1)
        A = torch.zeros(Co * 9, Ci)
        for oy in range(3):
            for ox in range(3):
                # output (Y+oy-1, X+ox-1) reads input (Y, X) through tap ky = 1 - (oy - 1), kx = 1 - (ox - 1)
                A[(torch.arange(Co) * 9 + oy * 3 + ox)] = w[:, :, 2 - oy, 2 - ox]
        sol = torch.linalg.lstsq(A, d.reshape(-1, 1)).solution.flatten()
        res = float((A @ sol - d.reshape(-1)).abs().max())
        vin = vact[n0, :, Y, X]
        print(f"tile {(n0, ty0, tx0)} input px (row {r}, col {c}) = staged (row {sr}, quad {q}, px {px}): residual {res:.2e}")
        print("  delta[ci]    ", [round(float(t), 4) for t in sol[:16]])
        print("  true v[ci]   ", [round(float(t), 4) for t in vin[:16]])
        print("  delta + v    ", [round(float(t), 4) for t in (sol + vin)[:16]])
        for name, cand in (("v at col+4 (next quad px0)", vact[n0, :, Y, X + 4] if X + 4 < W else None), ("v at row+1", vact[n0, :, Y + 1, X]),
                           ("v at row-1", vact[n0, :, Y - 1, X]), ("raw x (no BN)", x[n0, :, Y, X]), ("v of sample n+1", vact[(n0 + 1) % N, :, Y, X]), ("zero", torch.zeros(Ci))):
            if cand is not None:
                print(f"  |delta + v - {name}| max", float((sol + vin - cand).abs().max()))
        done += 1
        if done >= 2:
            break
if bad.any():
    e = (y.cpu() - ref)
    n0, ty0, tx0 = tiles[0].tolist()
    R0 = ty0 * th
    print("signed error, tile", (n0, ty0, tx0), "channel 0, rows x first 16 cols:")
    for rr in range(th):
        print("  ", [round(float(t), 3) for t in e[n0, 0, R0 + rr, tx0 * tw: tx0 * tw + 16]])
    print("channel 5:")
    for rr in range(th):
        print("  ", [round(float(t), 3) for t in e[n0, 5, R0 + rr, tx0 * tw: tx0 * tw + 16]])
if bad.any():
    e = (y.cpu() - ref)
    found = 0
    for (n0, ty0, tx0) in tiles.tolist():
        b = (err[n0, :, ty0 * th:(ty0 + 1) * th, tx0 * tw:(tx0 + 1) * tw] > tol).any(0)
        rows = b.any(1).nonzero().flatten().tolist()
        if len(rows) != 3 or rows[2] - rows[0] != 2:
            continue
        r = rows[1]
        cols = [c for c in range(4, tw - 4, 4) if b[r, c]]
        if not cols:
            continue
        c = cols[0]
        Y, X = ty0 * th + r, tx0 * tw + c
        d = e[n0, :, Y - 1:Y + 2, X - 1:X + 2]
        A = torch.zeros(Co * 9, Ci)
        for oy in range(3):
            for ox in range(3):
                A[(torch.arange(Co) * 9 + oy * 3 + ox)] = w[:, :, 2 - oy, 2 - ox]
        sol = torch.linalg.lstsq(A, d.reshape(-1, 1)).solution.flatten()
        res = float((A @ sol - d.reshape(-1)).abs().max())
        vin, xin = vact[n0, :, Y, X], x[n0, :, Y, X]
        print(f"ISOLATED tile {(n0, ty0, tx0)} input (row {r}, col {c}): residual {res:.2e}, max |err| {float(d.abs().max()):.3f}")
        torch.set_printoptions(precision=4, linewidth=200)
        print("  delta ", sol)
        print("  v     ", vin)
        print("  x     ", xin)
        print("  scale ", scale)
        print("  shift ", shift)
        found += 1
        if found >= 2:
            break
