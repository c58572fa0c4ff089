"""Per-kernel parity of the conv stack through the C ABI: conv2d forward / data-gradient mode / weight gradient,
BatchNorm statistics, with the loader transforms.  Checker: stock torch fp32 ops on the CPU (autograd for the
gradients).  `be` runs every case on the host emulator (CPU) and, with -m gpu, on the MI355X."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import close, rel_err

TOL = 1e-4


def virt_input(x, scale, shift, emask, es, cmask):
    v = torch.from_numpy(x)
    if scale is not None:
        v = F.leaky_relu(v * torch.from_numpy(scale)[None, :, None, None] + torch.from_numpy(shift)[None, :, None, None], 0.01)
    if emask is not None:
        v = v * torch.from_numpy(emask).float() * es
    if cmask is not None:
        v = v * torch.from_numpy(cmask)[:, :, None, None]
    return v


CASES = [  # N, H, W, Ca, Cb, Co, ks, transforms on a
    (2, 9, 11, 5, 0, 7, 3, False),
    (2, 16, 16, 3, 6, 20, 3, True),
    (1, 8, 8, 12, 0, 6, 1, False),
    (1, 12, 40, 4, 4, 16, 3, True),
    (1, 10, 70, 2, 0, 33, 3, False),
    (1, 20, 32, 9, 0, 40, 3, False),
    (2, 16, 16, 17, 0, 48, 1, True),
    (1, 16, 64, 12, 6, 24, 3, True),
    (1, 24, 128, 4, 0, 8, 3, True),
    (2, 8, 32, 20, 12, 70, 3, False),
    (1, 16, 64, 9, 0, 12, 1, True),
    (3, 16, 32, 8, 8, 16, 3, True),
    (4, 8, 16, 4, 0, 40, 3, True),
    # shapes the lean kernel takes (full tiles, Ci % 8 == 0, Co a multiple of the channel block)
    (2, 16, 16, 16, 16, 32, 3, True),
    (1, 8, 64, 8, 0, 16, 1, True),
    (1, 16, 64, 24, 8, 32, 3, True),
    (2, 8, 32, 8, 0, 64, 3, False),
    (1, 16, 16, 16, 0, 64, 1, True),
    (1, 16, 128, 8, 8, 16, 3, True),
    # ... and the lean weight-gradient kernel (channel blocks inside one source)
    (2, 16, 16, 16, 16, 16, 3, True),
    (1, 8, 32, 32, 32, 32, 3, True),
    (1, 8, 64, 16, 0, 16, 3, True),
    (1, 8, 32, 16, 0, 16, 3, True),
    (2, 16, 16, 32, 32, 32, 3, True),
    (1, 8, 32, 32, 0, 32, 1, True),
    (1, 8, 32, 32, 0, 64, 3, True),
    (2, 16, 16, 32, 32, 128, 3, True),
    # the narrow-operand weight-gradient kernel: first convolution (1 -> 16) and 4-class classifier (16 -> 4)
    (2, 16, 64, 1, 0, 16, 3, False),
    (2, 8, 128, 16, 0, 4, 3, False),
    (1, 16, 64, 16, 0, 4, 3, 'bn'),
    (3, 24, 64, 16, 0, 4, 3, 'bn'),
]


@pytest.fixture
def variant(be):
    """the packed-path kernels with Winograd wherever it fits, including the 16-channel blocks: the PRODUCT library's fixed routing
    (it has no switch to set: the hook below exists in the emulator / experiments builds only, where an environment variable could
    have moved the default)"""
    if hasattr(be.lib, "wsl_debug_conv_wino"):
        be.call("wsl_debug_conv_wino", 2)
        yield 2
        be.call("wsl_debug_conv_wino", -1)
    else:
        assert be.lib.wsl_conv2d_wino_ok(1, 8, 64, 16, 0, 16, 3) == 1 and be.lib.wsl_conv2d_wino_ok(1, 16, 16, 32, 32, 32, 3) == 1
        yield 2


@pytest.mark.parametrize("case", CASES)
def test_conv_fwd_dgrad_wgrad_stats(be, variant, case):
    N, H, W, Ca, Cb, Co, ks, tr = case
    rng = np.random.default_rng(hash(case) % 2**31)
    xa = rng.standard_normal((N, Ca, H, W)).astype(np.float32)
    xb = rng.standard_normal((N, Cb, H, W)).astype(np.float32) if Cb else None
    Ci = Ca + Cb
    w = (rng.standard_normal((Co, Ci, ks, ks)) * 0.2).astype(np.float32)
    bias = rng.standard_normal(Co).astype(np.float32)
    scale = shift = emask = cmask = None
    es = 1.0
    if tr:
        scale = (rng.standard_normal(Ca) * 0.5 + 1).astype(np.float32)
        shift = (rng.standard_normal(Ca) * 0.3).astype(np.float32)
    if tr is True:
        emask = (rng.random((N, Ca, H, W)) > 0.3).astype(np.uint8)
        es = float(np.float32(1 / 0.7))
        cmask = ((rng.random((N, Ca)) > 0.5) * 2.0).astype(np.float32)
    # ---- checker
    va = virt_input(xa, scale, shift, emask, es, cmask)
    vin = (torch.cat([va, torch.from_numpy(xb)], 1) if Cb else va).requires_grad_()
    wt = torch.from_numpy(w).requires_grad_()
    bt = torch.from_numpy(bias).requires_grad_()
    y_ref = F.conv2d(vin, wt, bt, padding=ks // 2)
    r = torch.from_numpy(rng.standard_normal((N, Co, H, W)).astype(np.float32))
    (y_ref * r).sum().backward()
    # ---- device
    d = {k: (be.arr(v) if v is not None else None) for k, v in
         dict(xa=xa, xb=xb, w=w, bias=bias, scale=scale, shift=shift, emask=emask, cmask=cmask, r=r.numpy()).items()}
    sa = be.src(d["xa"], Ca, scale=d["scale"], shift=d["shift"], emask=d["emask"], es=es, cmask=d["cmask"])
    sb = be.src(d["xb"], Cb) if Cb else be.src()
    y = be.zeros((N, Co, H, W))
    nblk = be.lib.wsl_conv2d_stat_blocks(N, H, W, Ci, Co, ks)
    part, cnt = be.zeros((nblk, Co, 2)), be.zeros((nblk,))
    be.call("wsl_conv2d_fwd", sa, sb, be.ptr(d["w"]), be.ptr(d["bias"]), be.ptr(y), Co * H * W, N, H, W, Co, ks, 0,
            be.ptr(part), be.ptr(cnt), be.stream)
    assert close(be.np(y), y_ref.detach().numpy(), TOL)
    # ---- packed fast path (aligned float4 staging + register prefetch), same outputs incl. the statistics
    fast = bool(be.lib.wsl_conv2d_fast_ok(sa, sb, be.ptr(y), Co * H * W, W))
    assert fast == (W % 4 == 0 and (Cb == 0 or Ca % 4 == 0))
    if fast:
        wp, y2 = be.zeros((ks * ks, Ci, Co)), be.zeros((N, Co, H, W))
        part2, cnt2 = be.zeros((nblk, Co, 2)), be.zeros((nblk,))
        be.call("wsl_conv2d_pack_weights", be.ptr(d["w"]), be.ptr(wp), Co, Ci, ks, 0, be.stream)
        be.call("wsl_conv2d_fwd", sa, sb, be.ptr(wp), be.ptr(d["bias"]), be.ptr(y2), Co * H * W, N, H, W, Co, ks, 2,
                be.ptr(part2), be.ptr(cnt2), be.stream)
        assert close(be.np(y2), y_ref.detach().numpy(), TOL)
        assert float(be.np(cnt2).sum()) == N * H * W     # per-wave partials: the counts tile the tensor exactly
        # without statistics (how the network calls a layer that no BatchNorm follows): 16 -> 4 classifiers with full
        # 8x64 tiles take the 4x4x1-MFMA kernel of wsl_conv4.hip
        y3 = be.zeros((N, Co, H, W))
        be.call("wsl_conv2d_fwd", sa, sb, be.ptr(wp), be.ptr(d["bias"]), be.ptr(y3), Co * H * W, N, H, W, Co, ks, 2,
                None, None, be.stream)
        assert close(be.np(y3), y_ref.detach().numpy(), TOL)
    # ---- Winograd F(2x2, 3x3) path for the layers it fits: same outputs and statistics
    def wino_shape(Ca_, Cb_, Co_):
        # 32-channel output blocks on 8 x 32 / 16 x 16 tiles; 16-channel blocks only on 8 x 64 / 8 x 32 tiles
        return (ks == 3 and (Ca_ + Cb_) % 8 == 0 and Ca_ + Cb_ <= 256 and (Cb_ == 0 or Ca_ % 8 == 0) and Co_ % 16 == 0
                and ((H % 8 == 0 and W % 32 == 0) or (H % 16 == 0 and W % 16 == 0 and Co_ % 32 == 0)))
    wino = bool(be.lib.wsl_conv2d_wino_ok(N, H, W, Ca, Cb, Co, ks))
    assert wino == (variant == 2 and wino_shape(Ca, Cb, Co))
    if wino:
        u, y4 = be.zeros((16, Ci, Co)), be.zeros((N, Co, H, W))
        part4, cnt4 = be.zeros((Co, nblk, 2)), be.zeros((nblk,))
        be.call("wsl_conv2d_pack_weights", be.ptr(d["w"]), be.ptr(u), Co, Ci, ks, 2, be.stream)
        be.call("wsl_conv2d_fwd", sa, sb, be.ptr(u), be.ptr(d["bias"]), be.ptr(y4), Co * H * W, N, H, W, Co, ks, 4,
                be.ptr(part4), be.ptr(cnt4), be.stream)
        assert close(be.np(y4), y_ref.detach().numpy(), TOL)
        assert rel_err(be.np(y4), be.np(y2)) < 1e-5                     # vs the direct MFMA kernel: fp32 round-off only
        assert float(be.np(cnt4).sum()) == N * H * W
        mean4, invstd4, sc4, sh4 = (be.zeros((Co,)) for _ in range(4))
        g4, b4 = be.arr(np.ones(Co, np.float32)), be.arr(np.zeros(Co, np.float32))
        be.call("wsl_bn_stats_finalize", be.ptr(part4), be.ptr(cnt4), nblk, Co, be.ptr(g4), be.ptr(b4), 1e-5, 0.1,
                None, None, None, be.ptr(mean4), be.ptr(invstd4), be.ptr(sc4), be.ptr(sh4), be.stream)
        yr4 = y_ref.detach()
        assert rel_err(be.np(mean4), yr4.mean((0, 2, 3)).numpy()) < 1e-5
        assert rel_err(be.np(invstd4), (1 / torch.sqrt(yr4.var((0, 2, 3), unbiased=False) + 1e-5)).numpy()) < 1e-5
        y5 = be.zeros((N, Co, H, W))                                     # without statistics / bias
        be.call("wsl_conv2d_fwd", sa, sb, be.ptr(u), None, be.ptr(y5), Co * H * W, N, H, W, Co, ks, 4, None, None, be.stream)
        assert close(be.np(y5), (y_ref.detach() - torch.from_numpy(bias)[None, :, None, None]).numpy(), TOL)
    elif fast and ks == 3:
        with pytest.raises(Exception, match="wino_ok"):
            be.call("wsl_conv2d_fwd", sa, sb, be.ptr(wp), None, be.ptr(y2), Co * H * W, N, H, W, Co, ks, 4, None, None, be.stream)
    # BatchNorm statistics from the epilogue partials
    gamma, beta = be.arr(np.linspace(0.5, 1.5, Co, dtype=np.float32)), be.arr(np.linspace(-0.2, 0.2, Co, dtype=np.float32))
    rm, rv = be.arr(np.full(Co, 0.1, np.float32)), be.arr(np.full(Co, 0.9, np.float32))
    nbt = be.arr(np.array([3], dtype=np.int64))
    mean, invstd, sc, sh = (be.zeros((Co,)) for _ in range(4))
    be.call("wsl_bn_stats_finalize", be.ptr(part), be.ptr(cnt), nblk, Co, be.ptr(gamma), be.ptr(beta), 1e-5, 0.1,
            be.ptr(rm), be.ptr(rv), be.ptr(nbt), be.ptr(mean), be.ptr(invstd), be.ptr(sc), be.ptr(sh), be.stream)
    yr = y_ref.detach()
    m_ref, v_ref = yr.mean((0, 2, 3)), yr.var((0, 2, 3), unbiased=False)
    n = N * H * W
    assert rel_err(be.np(mean), m_ref.numpy()) < 1e-5
    assert rel_err(be.np(invstd), (1 / torch.sqrt(v_ref + 1e-5)).numpy()) < 1e-5
    assert rel_err(be.np(rm), (0.9 * 0.1 + 0.1 * m_ref).numpy()) < 1e-5
    assert rel_err(be.np(rv), (0.9 * 0.9 + 0.1 * v_ref * n / (n - 1)).numpy()) < 1e-5
    assert int(be.np(nbt)[0]) == 4
    if fast:   # the fast path's (per-wave) partials finalise to the same statistics
        mean2, invstd2, sc2, sh2 = (be.zeros((Co,)) for _ in range(4))
        be.call("wsl_bn_stats_finalize", be.ptr(part2), be.ptr(cnt2), nblk, Co, be.ptr(gamma), be.ptr(beta), 1e-5, 0.1,
                None, None, None, be.ptr(mean2), be.ptr(invstd2), be.ptr(sc2), be.ptr(sh2), be.stream)
        assert rel_err(be.np(mean2), m_ref.numpy()) < 1e-5
        assert rel_err(be.np(invstd2), (1 / torch.sqrt(v_ref + 1e-5)).numpy()) < 1e-5
    # ---- data-gradient mode: d(vin) = conv(r, W^T flipped)
    dx = be.zeros((N, Ci, H, W))
    sr = be.src(d["r"], Co)
    be.call("wsl_conv2d_fwd", sr, be.src(), be.ptr(d["w"]), None, be.ptr(dx), Ci * H * W, N, H, W, Ci, ks, 1, None, None,
            be.stream)
    assert close(be.np(dx), vin.grad.numpy(), TOL)
    if bool(be.lib.wsl_conv2d_fast_ok(sr, be.src(), be.ptr(dx), Ci * H * W, W)):
        wpd, dx2 = be.zeros((ks * ks, Co, Ci)), be.zeros((N, Ci, H, W))
        be.call("wsl_conv2d_pack_weights", be.ptr(d["w"]), be.ptr(wpd), Ci, Co, ks, 1, be.stream)
        be.call("wsl_conv2d_fwd", sr, be.src(), be.ptr(wpd), None, be.ptr(dx2), Ci * H * W, N, H, W, Ci, ks, 3, None, None,
                be.stream)
        assert close(be.np(dx2), vin.grad.numpy(), TOL)
        if bool(be.lib.wsl_conv2d_wino_ok(N, H, W, Co, 0, Ci, ks)):
            assert variant == 2 and wino_shape(Co, 0, Ci)
            ud, dx4 = be.zeros((16, Co, Ci)), be.zeros((N, Ci, H, W))
            be.call("wsl_conv2d_pack_weights", be.ptr(d["w"]), be.ptr(ud), Ci, Co, ks, 3, be.stream)
            be.call("wsl_conv2d_fwd", sr, be.src(), be.ptr(ud), None, be.ptr(dx4), Ci * H * W, N, H, W, Ci, ks, 5, None, None,
                    be.stream)
            assert close(be.np(dx4), vin.grad.numpy(), TOL)
        else:
            assert not (variant == 2 and wino_shape(Co, 0, Ci))
    # ---- weight / bias gradient
    nws = be.lib.wsl_conv2d_wgrad_ws_bytes(N, H, W, Ci, Co, ks)
    ws, dw, db = be.ws(nws), be.zeros((Co, Ci, ks, ks)), be.zeros((Co,))
    be.call("wsl_conv2d_wgrad", sa, sb, be.ptr(d["r"]), Co * H * W, be.ptr(dw), be.ptr(db), N, H, W, Co, ks, be.ptr(ws),
            nws, be.stream)
    assert close(be.np(dw), wt.grad.numpy(), TOL)
    assert close(be.np(db), bt.grad.numpy(), TOL)


# (N, H, W, Ca, Cb, Co, persistent workgroups forced): Winograd weight-gradient launches whose workgroups walk several tiles each, in the
# interleaved XCD-grouped order (splits % 8 == 0) and in contiguous runs (otherwise); tiles 4 x 64, 4 x 32, 8 x 16; one and two sources
WALKS = [(3, 24, 64, 16, 0, 16, 8), (5, 16, 32, 16, 0, 16, 8), (2, 16, 128, 16, 0, 16, 8), (6, 16, 16, 16, 0, 16, 8),
         (3, 16, 64, 32, 32, 32, 16), (3, 32, 16, 32, 0, 32, 8), (2, 8, 128, 16, 16, 16, 16), (3, 24, 64, 16, 0, 16, 5),
         (2, 16, 64, 64, 0, 64, 32),
         # the narrow-operand kernel (first convolution 1 -> 16, classifier 16 -> 4; 8 x 64 tiles)
         (3, 24, 128, 1, 0, 16, 8), (5, 16, 64, 16, 0, 4, 8), (3, 24, 128, 16, 0, 4, 5)]


@pytest.mark.parametrize("case", WALKS)
def test_wgrad_tile_walk_orders(be_route, case):
    be = be_route   # forces the number of persistent workgroups: emulator / experiments build
    N, H, W, Ca, Cb, Co, wgs = case
    Ci = Ca + Cb
    rng = np.random.default_rng(sum(case))
    xa = rng.standard_normal((N, Ca, H, W)).astype(np.float32)
    xb = rng.standard_normal((N, Cb, H, W)).astype(np.float32) if Cb else None
    scale, shift = (rng.standard_normal(Ca) * 0.5 + 1).astype(np.float32), (rng.standard_normal(Ca) * 0.3).astype(np.float32)
    va = virt_input(xa, scale, shift, None, 1.0, None) if Ca > 1 else torch.from_numpy(xa)
    vin = torch.cat([va, torch.from_numpy(xb)], 1) if Cb else va
    wt = torch.zeros(Co, Ci, 3, 3, requires_grad=True)
    bt = torch.zeros(Co, requires_grad=True)
    r = rng.standard_normal((N, Co, H, W)).astype(np.float32)
    (F.conv2d(vin, wt, bt, padding=1) * torch.from_numpy(r)).sum().backward()
    d = {k: (be.arr(v) if v is not None else None) for k, v in dict(xa=xa, xb=xb, scale=scale, shift=shift, r=r).items()}
    sa = be.src(d["xa"], Ca, scale=d["scale"], shift=d["shift"]) if Ca > 1 else be.src(d["xa"], Ca)
    sb = be.src(d["xb"], Cb) if Cb else be.src()
    be.call("wsl_debug_wgrad_workgroups", wgs)
    try:
        nws = be.lib.wsl_conv2d_wgrad_ws_bytes(N, H, W, Ci, Co, 3)
        ws, dw, db = be.ws(nws), be.zeros((Co, Ci, 3, 3)), be.zeros((Co,))
        be.call("wsl_conv2d_wgrad", sa, sb, be.ptr(d["r"]), Co * H * W, be.ptr(dw), be.ptr(db), N, H, W, Co, 3, be.ptr(ws), nws,
                be.stream)
        assert close(be.np(dw), wt.grad.numpy(), TOL)
        assert close(be.np(db), bt.grad.numpy(), TOL)
        if Co == 4 and Ca == 16 and Cb == 0:   # the classifier's forward kernel is persistent too (conv_cls_kernel): same walk orders
            w = (rng.standard_normal((Co, Ci, 3, 3)) * 0.2).astype(np.float32)
            bias = rng.standard_normal(Co).astype(np.float32)
            y_ref = F.conv2d(vin, torch.from_numpy(w), torch.from_numpy(bias), padding=1).numpy()
            dwt, dbias, wp, y = be.arr(w), be.arr(bias), be.zeros((9, Ci, Co)), be.zeros((N, Co, H, W))
            be.call("wsl_conv2d_pack_weights", be.ptr(dwt), be.ptr(wp), Co, Ci, 3, 0, be.stream)
            be.call("wsl_conv2d_fwd", sa, sb, be.ptr(wp), be.ptr(dbias), be.ptr(y), Co * H * W, N, H, W, Co, 3, 2, None, None, be.stream)
            assert close(be.np(y), y_ref, TOL)
    finally:
        be.call("wsl_debug_wgrad_workgroups", 0)


def test_conv_channel_slice_views(be):
    """batch strides: read a channel slice of a wider tensor, write into a channel slice (no copies for cat/split)."""
    rng = np.random.default_rng(3)
    N, H, W = 2, 8, 16
    big = rng.standard_normal((N, 10, H, W)).astype(np.float32)
    w = (rng.standard_normal((6, 4, 3, 3)) * 0.2).astype(np.float32)
    y_ref = F.conv2d(torch.from_numpy(big[:, 3:7]), torch.from_numpy(w), padding=1).numpy()
    dbig, dw_ = be.arr(big), be.arr(w)
    out = be.zeros((N, 9, H, W))
    s = be.src(dbig, 4, bs=10 * H * W)
    s.x = be.ptr(dbig) + 3 * H * W * 4
    be.call("wsl_conv2d_fwd", s, be.src(), be.ptr(dw_), None, be.ptr(out) + 2 * H * W * 4, 9 * H * W, N, H, W, 6, 3, 0,
            None, None, be.stream)
    o = be.np(out)
    assert close(o[:, 2:8], y_ref, TOL)
    assert np.all(o[:, :2] == 0) and np.all(o[:, 8:] == 0)


def test_conv_argument_errors(be):
    x = be.zeros((1, 1, 4, 4))
    s = be.src(x, 1)
    with pytest.raises(Exception, match="kernel size 5"):
        be.call("wsl_conv2d_fwd", s, be.src(), be.ptr(x), None, be.ptr(x), 16, 1, 4, 4, 1, 5, 0, None, None, be.stream)
    with pytest.raises(Exception, match="null"):
        be.call("wsl_conv2d_fwd", s, be.src(), None, None, be.ptr(x), 16, 1, 4, 4, 1, 3, 0, None, None, be.stream)


# (run the suite with WSL_CONV_DMA=1 to route the data-gradient launches below through the opt-in LDS-DMA kernel)
PLANS = [(8, 64, 16), (8, 64, 32), (8, 32, 16), (8, 32, 32), (8, 32, 64), (16, 16, 16), (16, 16, 32), (16, 16, 64)]


@pytest.mark.parametrize("plan", PLANS)
@pytest.mark.parametrize("ks", [3, 1])
def test_every_lean_conv_instantiation(be_route, plan, ks):
    """each tile shape of conv_mfma2l_kernel, forced at a small size (the built-in table only picks the wide channel
    blocks for launches that fill the chip): forward with a transformed + a raw source, data gradient, statistics"""
    be = be_route   # forces tile plans: emulator / experiments build
    th, tw, ct = plan
    N, H, W, Ca, Cb, Co = 2, 2 * th if th == 8 else 16, tw, 16, 8, 64
    rng = np.random.default_rng(th * 1000 + tw * 10 + ct + ks)
    xa, xb = rng.standard_normal((N, Ca, H, W)).astype(np.float32), rng.standard_normal((N, Cb, H, W)).astype(np.float32)
    w = (rng.standard_normal((Co, Ca + Cb, ks, ks)) * 0.2).astype(np.float32)
    bias = rng.standard_normal(Co).astype(np.float32)
    scale, shift = (rng.standard_normal(Ca) * 0.5 + 1).astype(np.float32), (rng.standard_normal(Ca) * 0.3).astype(np.float32)
    emask = (rng.random((N, Ca, H, W)) > 0.3).astype(np.uint8)
    es = float(np.float32(1 / 0.7))
    vin = torch.cat([virt_input(xa, scale, shift, emask, es, None), torch.from_numpy(xb)], 1).requires_grad_()
    y_ref = F.conv2d(vin, torch.from_numpy(w), torch.from_numpy(bias), padding=ks // 2)
    r = rng.standard_normal((N, Co, H, W)).astype(np.float32)
    (y_ref * torch.from_numpy(r)).sum().backward()
    d = {k: be.arr(v) for k, v in dict(xa=xa, xb=xb, w=w, bias=bias, scale=scale, shift=shift, emask=emask, r=r).items()}
    sa = be.src(d["xa"], Ca, scale=d["scale"], shift=d["shift"], emask=d["emask"], es=es)
    sb = be.src(d["xb"], Cb)
    be.call("wsl_debug_conv_plan", th, tw, ct)
    try:
        nblk = be.lib.wsl_conv2d_stat_blocks(N, H, W, Ca + Cb, Co, ks)
        assert nblk == N * (H // th) * (W // tw)
        wp, y = be.zeros((ks * ks, Ca + Cb, Co)), be.zeros((N, Co, H, W))
        part, cnt = be.zeros((Co, nblk, 2)), be.zeros((nblk,))
        be.call("wsl_conv2d_pack_weights", be.ptr(d["w"]), be.ptr(wp), Co, Ca + Cb, ks, 0, be.stream)
        be.call("wsl_conv2d_fwd", sa, sb, be.ptr(wp), be.ptr(d["bias"]), be.ptr(y), Co * H * W, N, H, W, Co, ks, 2,
                be.ptr(part), be.ptr(cnt), be.stream)
        assert close(be.np(y), y_ref.detach().numpy(), TOL)
        p = be.np(part)
        assert rel_err(p[:, :, 0].sum(1), y_ref.detach().sum((0, 2, 3)).numpy()) < 1e-5
        # data gradient: dL/d(vin) = conv(r, w rotated), output channels = Ca + Cb = 24 -> blocks of 8 would not divide:
        # run it with a plan whose channel block divides 24 only when it is 8... so check the gradient w.r.t. xb's 8
        # channels through a second layer whose "output" is the 64-channel tensor instead: dgrad of a 64 -> 64 layer
        w2 = (rng.standard_normal((64, 64, ks, ks)) * 0.1).astype(np.float32)
        g = torch.from_numpy(r).requires_grad_()
        ref2 = F.conv_transpose2d(g, torch.from_numpy(w2), padding=ks // 2)          # = dL/dx of conv2d(x, w2) for dL/dy = r
        wd, dx = be.zeros((ks * ks, 64, 64)), be.zeros((N, 64, H, W))
        dw2 = be.arr(w2)
        be.call("wsl_conv2d_pack_weights", be.ptr(dw2), be.ptr(wd), 64, 64, ks, 1, be.stream)
        be.call("wsl_conv2d_fwd", be.src(d["r"], 64), be.src(), be.ptr(wd), None, be.ptr(dx), 64 * H * W, N, H, W, 64, ks, 3,
                None, None, be.stream)
        assert close(be.np(dx), ref2.detach().numpy(), TOL)
    finally:
        be.call("wsl_debug_conv_plan", 0, 0, 0)
