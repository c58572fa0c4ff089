"""BatchNorm+LeakyReLU+Dropout backward, 2x2 max-pool (fwd + arg-max routing), bilinear x2 (fwd + transpose), the
gradient fan-in of an encoder feature -- through the C ABI, against torch autograd and the reference's golden vectors."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import close, golden, rel_err

TOL = 1e-4


@pytest.mark.parametrize("shape,drop", [((2, 5, 9, 11), True), ((3, 16, 16, 16), False), ((1, 3, 70, 66), True)])
def test_bnact_bwd(be, shape, drop):
    N, C, H, W = shape
    rng = np.random.default_rng(N * 100 + C)
    y = (rng.standard_normal(shape) * 1.3 + 0.4).astype(np.float32)
    g = rng.standard_normal(shape).astype(np.float32)
    gamma = (rng.random(C) + 0.5).astype(np.float32)
    beta = (rng.standard_normal(C) * 0.2).astype(np.float32)
    emask = (rng.random(shape) > 0.25).astype(np.uint8) if drop else None
    es = float(np.float32(1 / 0.75)) if drop else 1.0
    yt = torch.from_numpy(y).requires_grad_()
    gt, bt = torch.from_numpy(gamma).requires_grad_(), torch.from_numpy(beta).requires_grad_()
    out = F.leaky_relu(F.batch_norm(yt, None, None, gt, bt, True, 0.1, 1e-5), 0.01)
    if drop:
        out = out * torch.from_numpy(emask).float() * es
    (out * torch.from_numpy(g)).sum().backward()
    mean = y.mean((0, 2, 3)).astype(np.float32)
    invstd = (1 / np.sqrt(y.astype(np.float64).var((0, 2, 3)) + 1e-5)).astype(np.float32)
    d = [be.arr(a) for a in (g, y, mean, invstd, gamma, beta)]
    dm = be.arr(emask) if drop else None
    dy, dgam, dbet = be.zeros(shape), be.zeros((C,)), be.zeros((C,))
    nws = be.lib.wsl_bnact_bwd_ws_bytes(N, C, H, W)
    ws = be.ws(nws)
    be.call("wsl_bnact_bwd", be.ptr(d[0]), C * H * W, *[be.ptr(a) for a in d[1:]], be.ptr(dm) if drop else None, es,
            be.ptr(dy), be.ptr(dgam), be.ptr(dbet), N, C, H, W, be.ptr(ws), nws, be.stream)
    assert close(be.np(dy), yt.grad.numpy(), TOL)
    assert close(be.np(dgam), gt.grad.numpy(), TOL)
    assert close(be.np(dbet), bt.grad.numpy(), TOL)
    # the same pass leaving max |dy| for the split-precision consumers: the largest of the 64 slots is the bit pattern of the maximum
    # of the dy it wrote (every slot written: the garbage put there first must be gone), dy itself unchanged
    dy2, slots = be.zeros(shape), be.arr(np.full(64, 0x7f7fffff, np.uint32).view(np.float32))
    be.call("wsl_bnact_bwd_amax", be.ptr(d[0]), C * H * W, *[be.ptr(a) for a in d[1:]], be.ptr(dm) if drop else None, es,
            be.ptr(dy2), be.ptr(dgam), be.ptr(dbet), N, C, H, W, be.ptr(ws), nws, be.ptr(slots), be.stream)
    assert np.array_equal(be.np(dy2), be.np(dy))
    assert be.np(slots).view(np.uint32).max() == np.abs(be.np(dy)).max().view(np.uint32)


def test_pool_and_routing_golden(be):
    """nn.MaxPool2d(2) forward and first-maximum gradient routing on the tie-heavy golden case."""
    g = golden("g1_pool_up")
    x, r = g["mp_x"], g["mp_r"]
    N, C, H, W = x.shape
    dx_, dr = be.arr(x), be.arr(r)
    out = be.zeros((N, C, H // 2, W // 2))
    s = be.src(dx_, C)
    be.call("wsl_pool2_fwd", s, be.ptr(out), N, H, W, be.stream)
    assert np.array_equal(be.np(out), g["mp_y"])
    gx = be.zeros(x.shape)
    be.call("wsl_feat_grad_combine", s, None, 0, None, 0, None, be.ptr(dr), be.ptr(gx), N, H, W, be.stream)
    assert np.array_equal(be.np(gx), g["mp_dx"])


def test_feat_grad_combine_all_sources(be):
    rng = np.random.default_rng(5)
    N, C, H, W = 2, 3, 7, 10           # odd H: last row belongs to no pooling window
    yv = rng.standard_normal((N, C, H, W)).astype(np.float32)
    scale, shift = (rng.standard_normal(C)).astype(np.float32), rng.standard_normal(C).astype(np.float32)
    ga = rng.standard_normal((N, C + 2, H, W)).astype(np.float32)     # channel slice [1:1+C] of a wider tensor
    gb = rng.standard_normal((N, C, H, W)).astype(np.float32)
    cm = ((rng.random((N, C)) > 0.5) * 2.0).astype(np.float32)
    gp = rng.standard_normal((N, C, H // 2, W // 2)).astype(np.float32)
    f = F.leaky_relu(torch.from_numpy(yv) * torch.from_numpy(scale)[None, :, None, None]
                     + torch.from_numpy(shift)[None, :, None, None], 0.01).requires_grad_()
    (F.max_pool2d(f, 2) * torch.from_numpy(gp)).sum().backward()
    ref = f.grad.numpy() + ga[:, 1:1 + C] + gb * cm[:, :, None, None]
    d = {k: be.arr(v) for k, v in dict(y=yv, scale=scale, shift=shift, ga=ga, gb=gb, cm=cm, gp=gp).items()}
    out = be.zeros((N, C, H, W))
    s = be.src(d["y"], C, scale=d["scale"], shift=d["shift"])
    be.call("wsl_feat_grad_combine", s, be.ptr(d["ga"]) + H * W * 4, (C + 2) * H * W, be.ptr(d["gb"]), C * H * W,
            be.ptr(d["cm"]), be.ptr(d["gp"]), be.ptr(out), N, H, W, be.stream)
    assert rel_err(be.np(out), ref) < 1e-6


def test_bilinear_golden(be):
    g = golden("g1_pool_up")
    for tag in ("u1", "u2", "u3", "u4"):
        x, r = g[f"{tag}_x"], g[f"{tag}_r"]
        N, C, h, w = x.shape
        dx_, dr = be.arr(x), be.arr(r)
        out, du = be.zeros((N, C, 2 * h, 2 * w)), be.zeros(x.shape)
        be.call("wsl_bilinear_up2_fwd", be.ptr(dx_), be.ptr(out), C * 4 * h * w, N, C, h, w, be.stream)
        be.call("wsl_bilinear_up2_bwd", be.ptr(dr), C * 4 * h * w, be.ptr(du), N, C, h, w, be.stream)
        assert rel_err(be.np(out), g[f"{tag}_y"]) < 1e-6, tag
        assert rel_err(be.np(du), g[f"{tag}_dx"]) < 1e-5, tag


@pytest.mark.parametrize("shape", [(2, 3, 16, 16), (1, 2, 24, 32), (1, 2, 8, 128), (1, 1, 40, 64), (2, 2, 5, 7), (1, 2, 6, 12),
                                   (1, 2, 32, 64), (1, 1, 64, 32), (1, 2, 32, 16), (1, 1, 64, 128)])   # (h % 32 == 0: the 32-row transpose tiles)
def test_bilinear_fast_and_fallback_shapes(be, shape):
    """power-of-two widths >= 16 take the float4 / LDS-tile kernels, anything else the reference kernels"""
    N, C, h, w = shape
    rng = np.random.default_rng(h * 131 + w)
    x = torch.from_numpy(rng.standard_normal(shape).astype(np.float32)).requires_grad_()
    r = rng.standard_normal((N, C, 2 * h, 2 * w)).astype(np.float32)
    y = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    (y * torch.from_numpy(r)).sum().backward()
    dx_, dr = be.arr(x.detach().numpy()), be.arr(r)
    out, du = be.zeros((N, C, 2 * h, 2 * w)), be.zeros(shape)
    be.call("wsl_bilinear_up2_fwd", be.ptr(dx_), be.ptr(out), C * 4 * h * w, N, C, h, w, be.stream)
    be.call("wsl_bilinear_up2_bwd", be.ptr(dr), C * 4 * h * w, be.ptr(du), N, C, h, w, be.stream)
    # fp32: the source coordinate scale*x carries ~1e-5 absolute rounding at x ~ 255, and the device contracts a*b+c
    assert rel_err(be.np(out), y.detach().numpy()) < 2e-5
    assert rel_err(be.np(du), x.grad.numpy()) < 1e-5


def test_src_materialize_and_eval_affine(be):
    rng = np.random.default_rng(9)
    N, C, H, W = 2, 4, 6, 5
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    gamma, beta = (rng.random(C) + 0.5).astype(np.float32), rng.standard_normal(C).astype(np.float32)
    rm, rv = rng.standard_normal(C).astype(np.float32), (rng.random(C) + 0.5).astype(np.float32)
    d = [be.arr(a) for a in (x, gamma, beta, rm, rv)]
    sc, sh, out = be.zeros((C,)), be.zeros((C,)), be.zeros((N, C, H, W))
    be.call("wsl_bn_eval_affine", *[be.ptr(a) for a in d[1:]], 1e-5, C, be.ptr(sc), be.ptr(sh), be.stream)
    be.call("wsl_src_materialize", be.src(d[0], C, scale=sc, shift=sh), be.ptr(out), C * H * W, N, H, W, be.stream)
    ref = F.leaky_relu(F.batch_norm(torch.from_numpy(x), torch.from_numpy(rm), torch.from_numpy(rv),
                                    torch.from_numpy(gamma), torch.from_numpy(beta), False, 0.1, 1e-5), 0.01)
    assert rel_err(be.np(out), ref.numpy()) < 1e-5


def _bn_setup(rng, N, C, H, W, with_mask):
    y = rng.standard_normal((N, C, H, W)).astype(np.float32)
    gamma, beta = (rng.random(C) + 0.5).astype(np.float32), (rng.standard_normal(C) * 0.3).astype(np.float32)
    mean = y.mean((0, 2, 3)).astype(np.float32)
    invstd = (1.0 / np.sqrt(y.var((0, 2, 3)) + 1e-5)).astype(np.float32)
    scale = (gamma * invstd).astype(np.float32)
    shift = (beta - mean * scale).astype(np.float32)
    emask = (rng.random((N, C, H, W)) > 0.3).astype(np.uint8) if with_mask else None
    return y, gamma, beta, mean, invstd, np.concatenate([mean, invstd, scale, shift]).astype(np.float32), emask


# (N, H, W, Cdy, Cg, ks, mask, fused): which kernel emits the statistics
FUSED_CASES = [
    (2, 16, 32, 32, 32, 3, True, 1),     # conv_wino2r (32-channel blocks)
    (1, 8, 64, 16, 16, 3, False, 1),     # conv_wino2r 8 x 64 tiles (the 16-channel layers at full width)
    (2, 16, 16, 64, 32, 3, True, 1),     # conv_wino2r 16 x 16 tiles
    (2, 16, 32, 64, 32, 1, False, 1),    # direct lean kernel, 1x1 (decoder conv1x1 data gradient), 32-channel block
    (1, 16, 64, 4, 16, 3, False, 1),     # narrow-K kernel (classifier data gradient 4 -> 16)
    (2, 16, 32, 32, 64, 1, False, 2),    # 64 output channels: a 64-channel block (64 accumulators) has no epilogue -> the caller
                                         # falls back; on a 256-CU device the plan shrinks the block to 32 -> fused (2 = either)
    (2, 8, 32, 16, 16, 3, True, 1),      # conv_wino2r 8 x 32 tiles x 16 channels (16-channel block below width 64)
    (1, 6, 10, 8, 8, 3, False, 0),       # odd shape, raw weights: no epilogue
]


@pytest.mark.parametrize("case", FUSED_CASES)
def test_dgrad_with_bn_backward_statistics_equals_the_two_pass_form(be, case):
    """wsl_conv2d_dgrad_bn + wsl_bnact_bwd_finish (statistics from the convolution's epilogue) against the plain data gradient
    followed by wsl_bnact_bwd (stand-alone reduction pass): same dy, dgamma, dbeta up to the summation order"""
    import ctypes as C_
    N, H, W, Cdy, Cg, ks, with_mask, expect_fused = case
    rng = np.random.default_rng(sum(case[:6]) * 7 + 1)
    dyv = (rng.standard_normal((N, Cdy, H, W)) * 0.1).astype(np.float32)
    w = (rng.standard_normal((Cdy, Cg, ks, ks)) * 0.2).astype(np.float32)      # layer Cg -> Cdy: its data gradient maps dy -> g
    y, gamma, beta, mean, invstd, st, emask = _bn_setup(rng, N, Cg, H, W, with_mask)
    es = 1.0 / 0.7
    d = {k: be.arr(v) for k, v in dict(dy=dyv, w=w, y=y, gamma=gamma, beta=beta, mean=mean, invstd=invstd, st=st).items()}
    dm = be.arr(emask) if with_mask else None
    src = be.src(d["dy"], Cdy)
    fast = bool(be.lib.wsl_conv2d_fast_ok(C_.byref(src), None, None, 0, W)) and W % 4 == 0
    if fast and be.lib.wsl_conv2d_wino_ok(N, H, W, Cdy, 0, Cg, ks):
        wp, wmode = be.zeros((16 * Cdy * Cg,)), 5
        be.call("wsl_conv2d_pack_weights", be.ptr(d["w"]), be.ptr(wp), Cdy, Cg, ks, 3, be.stream)
    elif fast:
        wp, wmode = be.zeros((ks * ks * Cdy * Cg,)), 3
        be.call("wsl_conv2d_pack_weights", be.ptr(d["w"]), be.ptr(wp), Cdy, Cg, ks, 1, be.stream)
    else:
        wp, wmode = d["w"], 1
    nws = be.lib.wsl_bnact_bwd_ws_bytes(N, Cg, H, W)
    # --- two-pass reference
    g_ref, dy_ref = be.zeros((N, Cg, H, W)), be.zeros((N, Cg, H, W))
    dg_ref, db_ref, ws = be.zeros((Cg,)), be.zeros((Cg,)), be.ws(nws)
    be.call("wsl_conv2d_fwd", src, be.src(), be.ptr(wp), None, be.ptr(g_ref), Cg * H * W, N, H, W, Cg, ks, wmode, None, None,
            be.stream)
    be.call("wsl_bnact_bwd", be.ptr(g_ref), Cg * H * W, be.ptr(d["y"]), be.ptr(d["mean"]), be.ptr(d["invstd"]), be.ptr(d["gamma"]),
            be.ptr(d["beta"]), be.ptr(dm) if with_mask else None, es, be.ptr(dy_ref), be.ptr(dg_ref), be.ptr(db_ref), N, Cg, H, W,
            be.ptr(ws), nws, be.stream)
    # --- fused
    g, dy = be.zeros((N, Cg, H, W)), be.zeros((N, Cg, H, W))
    dg, db, part, coef = be.zeros((Cg,)), be.zeros((Cg,)), be.ws(nws), be.zeros((2 * Cg,))
    fused = C_.c_int(-1)
    be.call("wsl_conv2d_dgrad_bn", src, be.ptr(wp), be.ptr(g), Cg * H * W, N, H, W, Cg, ks, wmode, be.ptr(d["y"]), be.ptr(d["st"]),
            be.ptr(dm) if with_mask else None, es, be.ptr(part), C_.byref(fused), be.stream)
    assert np.array_equal(be.np(g), be.np(g_ref))                           # the gradient itself: same kernel, same bits
    assert fused.value in ((0, 1) if expect_fused == 2 else (expect_fused,)), (case, fused.value)
    if fused.value:
        nblk = be.lib.wsl_conv2d_stat_blocks(N, H, W, Cdy, Cg, ks)
        be.call("wsl_bnact_bwd_finish", be.ptr(g), Cg * H * W, be.ptr(d["y"]), be.ptr(d["mean"]), be.ptr(d["invstd"]),
                be.ptr(d["gamma"]), be.ptr(d["beta"]), be.ptr(dm) if with_mask else None, es, be.ptr(dy), be.ptr(dg), be.ptr(db),
                N, Cg, H, W, be.ptr(part), nblk, 1, be.ptr(coef), 8 * Cg, be.stream)
        assert rel_err(be.np(dg), be.np(dg_ref)) < 2e-6 and rel_err(be.np(db), be.np(db_ref)) < 2e-6
        assert rel_err(be.np(dy), be.np(dy_ref)) < 2e-6
    # --- round 6: the d form -- the epilogue writes dz = g * keep * scale * leaky'(z) in place of g where the dispatched kernel can
    # (*fused == 2: the raw-source Winograd kernel), and wsl_bnact_bwd_finish_d_amax reads dz and y only.  Same dy, dgamma, dbeta.
    g2, dy2 = be.zeros((N, Cg, H, W)), be.zeros((N, Cg, H, W))
    dg2, db2, part2, coef2 = be.zeros((Cg,)), be.zeros((Cg,)), be.ws(nws), be.zeros((2 * Cg,))
    fused2 = C_.c_int(-1)
    be.call("wsl_conv2d_dgrad_bn_d", src, be.ptr(wp), be.ptr(g2), Cg * H * W, N, H, W, Cg, ks, wmode, be.ptr(d["y"]), be.ptr(d["st"]),
            be.ptr(dm) if with_mask else None, es, be.ptr(part2), C_.byref(fused2), be.stream)
    assert (fused2.value == 0) == (fused.value == 0), (fused.value, fused2.value)
    assert fused2.value == (2 if wmode == 5 and fused.value else fused.value), (case, fused2.value)
    if fused2.value == 2:
        z = y * st[2 * Cg:3 * Cg][None, :, None, None] + st[3 * Cg:][None, :, None, None]
        keep = (emask.astype(np.float32) * np.float32(es)) if with_mask else np.float32(1.0)
        dz_ref = be.np(g_ref) * keep * np.where(z > 0, np.float32(1.0), np.float32(0.01))
        assert rel_err(be.np(g2), dz_ref) < 1e-6
        nblk = be.lib.wsl_conv2d_stat_blocks(N, H, W, Cdy, Cg, ks)
        be.call("wsl_bnact_bwd_finish_d_amax", be.ptr(g2), Cg * H * W, be.ptr(d["y"]), be.ptr(d["mean"]), be.ptr(d["invstd"]),
                be.ptr(d["gamma"]), be.ptr(d["beta"]), be.ptr(dy2), be.ptr(dg2), be.ptr(db2), N, Cg, H, W, be.ptr(part2), nblk, 1,
                be.ptr(coef2), 8 * Cg, None, be.stream)
        assert rel_err(be.np(dg2), be.np(dg_ref)) < 2e-6 and rel_err(be.np(db2), be.np(db_ref)) < 2e-6
        assert rel_err(be.np(dy2), be.np(dy_ref)) < 2e-6
    elif fused2.value == 1:
        assert np.array_equal(be.np(g2), be.np(g_ref))


@pytest.mark.parametrize("shape,pool", [((2, 3, 8, 12), True), ((1, 4, 7, 10), True), ((2, 2, 6, 6), False)])
def test_fan_in_with_bn_backward_statistics_equals_the_two_pass_form(be, shape, pool):
    """wsl_feat_grad_combine_bn + wsl_bnact_bwd_finish against wsl_feat_grad_combine + wsl_bnact_bwd"""
    N, C, H, W = shape
    rng = np.random.default_rng(H * 17 + W)
    y, gamma, beta, mean, invstd, st, _ = _bn_setup(rng, N, C, H, W, False)
    ga = rng.standard_normal((N, C, H, W)).astype(np.float32)
    gb = rng.standard_normal((N, C, H, W)).astype(np.float32)
    cm = ((rng.random((N, C)) > 0.5) * 2.0).astype(np.float32)
    gp = rng.standard_normal((N, C, H // 2, W // 2)).astype(np.float32)
    d = {k: be.arr(v) for k, v in dict(y=y, gamma=gamma, beta=beta, mean=mean, invstd=invstd, scale=st[2 * C:3 * C], shift=st[3 * C:],
                                       ga=ga, gb=gb, cm=cm, gp=gp).items()}
    f = be.src(d["y"], C, scale=d["scale"], shift=d["shift"])
    nws = be.lib.wsl_bnact_bwd_ws_bytes(N, C, H, W)
    outs = []
    for fused in (False, True):
        g, dy, dg, db, ws, coef = be.zeros(shape), be.zeros(shape), be.zeros((C,)), be.zeros((C,)), be.ws(nws), be.zeros((2 * C,))
        args = [f, be.ptr(d["ga"]), C * H * W, be.ptr(d["gb"]), C * H * W, be.ptr(d["cm"]), be.ptr(d["gp"]) if pool else None,
                be.ptr(g), N, H, W]
        bn = [be.ptr(g), C * H * W, be.ptr(d["y"]), be.ptr(d["mean"]), be.ptr(d["invstd"]), be.ptr(d["gamma"]), be.ptr(d["beta"]),
              None, 1.0, be.ptr(dy), be.ptr(dg), be.ptr(db), N, C, H, W]
        if fused:
            be.call("wsl_feat_grad_combine_bn", *args, be.ptr(d["mean"]), be.ptr(d["invstd"]), be.ptr(ws), be.stream)
            be.call("wsl_bnact_bwd_finish", *bn, be.ptr(ws), be.lib.wsl_feat_grad_combine_blocks(N, H, W), 0, be.ptr(coef), 8 * C,
                    be.stream)
        else:
            be.call("wsl_feat_grad_combine", *args, be.stream)
            be.call("wsl_bnact_bwd", *bn, be.ptr(ws), nws, be.stream)
        outs.append([be.np(t).copy() for t in (g, dy, dg, db)])
    assert np.array_equal(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1:], outs[1][1:]):
        assert rel_err(b, a) < 2e-6
    # stage 2 that also leaves max |dy| (larger workspace: one partial maximum per workgroup)
    nfw = be.lib.wsl_bnact_bwd_finish_ws_bytes(N, C, H, W, 1)
    assert nfw > 8 * C == be.lib.wsl_bnact_bwd_finish_ws_bytes(N, C, H, W, 0)
    dy2, wsf, slots = be.zeros(shape), be.ws(nfw), be.arr(np.full(64, 0x7f7fffff, np.uint32).view(np.float32))
    bn[9] = be.ptr(dy2)
    be.call("wsl_bnact_bwd_finish_amax", *bn, be.ptr(ws), be.lib.wsl_feat_grad_combine_blocks(N, H, W), 0, be.ptr(wsf), nfw,
            be.ptr(slots), be.stream)
    assert np.array_equal(be.np(dy2), outs[1][1])
    assert be.np(slots).view(np.uint32).max() == np.abs(outs[1][1]).max().view(np.uint32)


@pytest.mark.parametrize("nblk", [37, 2048 + 37])
def test_bn_stats_finalize_four_and_sixteen_waves(be, nblk):
    """wsl_bn_stats_finalize merges per-tile (sum, M2, count) partials, one workgroup per channel; from 2048 partials per channel on it
    runs 16 waves instead of 4 (the kernel is a chain of memory round trips).  Both forms against a numpy fp64 Chan merge, with empty
    slots (count 0, garbage data) in between."""
    C = 5
    rng = np.random.default_rng(nblk)
    cnt = rng.integers(100, 400, nblk).astype(np.float32)
    dead = rng.random(nblk) < 0.1
    cnt[dead] = 0
    mean_b = rng.standard_normal((C, nblk)) * 0.5 + np.linspace(-3, 3, C)[:, None]
    var_b = rng.random((C, nblk)) + 0.1
    part = np.empty((C, nblk, 2), np.float32)
    part[..., 0] = mean_b * cnt
    part[..., 1] = var_b * cnt
    part[:, dead] = 1e30                                                     # must be ignored
    live = ~dead
    n = cnt[live].astype(np.float64).sum()
    s = part[:, live, 0].astype(np.float64).sum(1)
    mean = s / n
    mb = part[:, live, 0].astype(np.float64) / cnt[live]
    m2 = (part[:, live, 1].astype(np.float64) + cnt[live] * (mb - mean[:, None]) ** 2).sum(1)
    gamma, beta = rng.random(C).astype(np.float32) + 0.5, rng.standard_normal(C).astype(np.float32)
    d = [be.arr(a) for a in (part, cnt, gamma, beta)]
    rm, rv = be.arr(np.zeros(C, np.float32)), be.arr(np.ones(C, np.float32))
    o = [be.zeros((C,)) for _ in range(4)]
    be.call("wsl_bn_stats_finalize", be.ptr(d[0]), be.ptr(d[1]), nblk, C, be.ptr(d[2]), be.ptr(d[3]), 1e-5, 0.1, be.ptr(rm), be.ptr(rv),
            None, *[be.ptr(a) for a in o], be.stream)
    invstd = 1 / np.sqrt(m2 / n + 1e-5)
    assert rel_err(be.np(o[0]), mean) < 1e-6
    assert rel_err(be.np(o[1]), invstd) < 1e-6
    assert rel_err(be.np(o[2]), gamma * invstd) < 1e-6
    assert rel_err(be.np(o[3]), beta - mean * gamma * invstd) < 1e-5
    assert rel_err(be.np(rm), 0.1 * mean) < 1e-6
    assert rel_err(be.np(rv), 0.9 + 0.1 * m2 / (n - 1)) < 1e-6
