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


@pytest.mark.parametrize("shape", [(2, 3, 16, 16), (1, 2, 24, 32), (1, 2, 8, 128), (1, 1, 40, 64), (2, 2, 5, 7), (1, 2, 6, 12)])
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
