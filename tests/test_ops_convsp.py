"""Split-precision conv path (f16 hi/lo operands, three MFMA passes, fp32 accumulate; include/wsl_hip.h "split-precision conv
path") through the C ABI: forward / data gradient / weight gradient against stock torch fp32 autograd at the SAME 1e-4 criteria
as the f32 kernels (tests/test_ops_conv.py), plus the deviation from an fp64 truth next to the f32 kernels' own.
`be` runs every case on the host emulator (CPU) and, with -m gpu, on the MI355X."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import close, rel_err
from test_ops_conv import virt_input

TOL = 1e-4

CASES = [  # N, H, W, Ca, Cb, Co, transforms on a
    (2, 8, 32, 16, 0, 16, True),
    (1, 16, 32, 16, 16, 32, True),
    (1, 8, 64, 32, 0, 16, False),
    (2, 16, 16, 16, 0, 64, True),
    (1, 16, 16, 32, 32, 32, True),
    (1, 8, 32, 48, 0, 64, "bn"),
    (1, 16, 16, 64, 0, 16, False),
    (3, 8, 32, 16, 16, 128, True),
    (1, 8, 128, 16, 16, 16, True),        # more than one tile per row at 16 output channels
    (2, 16, 16, 32, 0, 64, "bn"),        # 16-column tiles with a 64-wide output block (the 16 x 16 level's form)
    (1, 8, 32, 32, 16, 32, "bn_cm"),     # loader switches outside the four specialised staging forms
]
# sizes at which the GPU launches take the shapes of the full-size step: 64-channel output blocks with streamed weight blocks,
# several tiles per persistent workgroup, weight-gradient runs of several tiles per split (the emulator reaches the same code with
# its single "CU"; it runs the small cases only for time)
BIG = [
    (32, 64, 64, 64, 0, 64, "bn"),
    (32, 32, 64, 32, 32, 64, True),
    (48, 32, 32, 128, 0, 128, "bn"),
    (8, 128, 128, 32, 0, 32, True),
    (8, 256, 256, 16, 16, 16, True),      # the full-resolution decoder layer: wide tiles, persistent workgroups over many tiles
    (64, 16, 16, 128, 0, 256, "bn"),     # the deepest encoder layer as the step runs it: 8 x 16 tiles, 64-wide blocks
]


def _amax_bits(be):
    return be.zeros((4,), np.int64)     # one max |w| word (the library writes / reads the first uint32)


def _set_amax(be, slot, value):
    """a tensor's maximum as its producer leaves it: WSL_SP_AMAX_SLOTS (64) partial maxima, here one of them non-zero"""
    a = np.zeros(32, np.int64)
    a.view(np.uint32)[37] = np.float32(value).view(np.uint32)
    return be.arr(a)


@pytest.mark.gpu
@pytest.mark.parametrize("case", BIG)
def test_sp_conv_big_gpu(case):
    from conftest import get_backend
    torch.set_num_threads(min(32, torch.get_num_threads()))
    test_sp_conv_fwd_dgrad_wgrad(get_backend("hip"), case)


@pytest.mark.parametrize("case", CASES)
def test_sp_conv_fwd_dgrad_wgrad(be, case):
    N, H, W, Ca, Cb, Co, tr = case
    rng = np.random.default_rng(1000 + (CASES + BIG).index(case))     # (hash() of a tuple holding a str changes from process to process)
    xa = rng.standard_normal((N, Ca, H, W)).astype(np.float32)
    xb = (rng.standard_normal((N, Cb, H, W)) * 3).astype(np.float32) if Cb else None
    Ci = Ca + Cb
    w = (rng.standard_normal((Co, Ci, 3, 3)) * 0.07).astype(np.float32)
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
    if tr == "bn_cm":      # BatchNorm source with channel multipliers but no keep mask: no network has it -- the staging code's generic form
        cmask = ((rng.random((N, Ca)) > 0.5) * 2.0).astype(np.float32)
    va = virt_input(xa, scale, shift, emask, es, cmask)
    vin = (torch.cat([va, torch.from_numpy(xb)], 1) if Cb else va).requires_grad_()
    wt = torch.from_numpy(w).requires_grad_()
    bt = torch.from_numpy(bias).requires_grad_()
    y_ref = F.conv2d(vin, wt, bt, padding=1)
    r = (rng.standard_normal((N, Co, H, W)) * 3e-5).astype(np.float32)     # a gradient-sized tensor: exercises the dynamic scale
    (y_ref * torch.from_numpy(r)).sum().backward()
    y64 = F.conv2d(vin.detach().double(), wt.detach().double(), bt.detach().double(), padding=1).numpy()

    d = {k: (be.arr(v) if v is not None else None) for k, v in
         dict(xa=xa, xb=xb, w=w, bias=bias, scale=scale, shift=shift, emask=emask, cmask=cmask, r=r).items()}
    sa = be.src(d["xa"], Ca, scale=d["scale"], shift=d["shift"], emask=d["emask"], es=es, cmask=d["cmask"])
    sb = be.src(d["xb"], Cb) if Cb else be.src()
    y = be.zeros((N, Co, H, W))
    assert be.lib.wsl_sp_conv2d_ok(sa, sb, be.ptr(y), Co * H * W, N, H, W, Co, 3) == 1
    nbytes = be.lib.wsl_sp_weight_image_bytes(Co, Ci)
    assert nbytes == 40 * Ci * Co
    img, wmax = be.ws(nbytes), _amax_bits(be)
    be.call("wsl_sp_pack_weights", be.ptr(d["w"]), be.ptr(img), be.ptr(wmax), Co, Ci, 0, be.stream)
    assert be.np(wmax).view(np.uint32)[0] == np.abs(w).max().view(np.uint32)
    nblk = be.lib.wsl_sp_conv2d_stat_blocks(N, H, W, Ci, Co)
    part, cnt = be.zeros((Co, nblk, 2)), be.zeros((nblk,))
    be.call("wsl_sp_conv2d_fwd", sa, sb, be.ptr(img), be.ptr(wmax), None, be.ptr(d["bias"]), be.ptr(y), Co * H * W, N, H, W, Co,
            be.ptr(part), be.ptr(cnt), be.stream)
    got = be.np(y)
    assert close(got, y_ref.detach().numpy(), TOL)
    # distance from the fp64 truth: the split path vs stock torch fp32 (must be the same class, not 1e-4)
    e_sp, e_f32 = rel_err(got, y64), rel_err(y_ref.detach().numpy(), y64)
    print(f"SP-FWD {case}: split {e_sp:.2e}  torch-fp32 {e_f32:.2e} from the fp64 truth")
    assert e_sp < 6 * max(e_f32, 2e-7)
    # BatchNorm statistics from the epilogue partials
    assert float(be.np(cnt).sum()) == N * H * W
    mean, invstd, sc, sh = (be.zeros((Co,)) for _ in range(4))
    g1, b0 = be.arr(np.ones(Co, np.float32)), be.arr(np.zeros(Co, np.float32))
    be.call("wsl_bn_stats_finalize", be.ptr(part), be.ptr(cnt), nblk, Co, be.ptr(g1), be.ptr(b0), 1e-5, 0.1, None, None, None,
            be.ptr(mean), be.ptr(invstd), be.ptr(sc), be.ptr(sh), be.stream)
    yr = y_ref.detach()
    assert rel_err(be.np(mean), yr.mean((0, 2, 3)).numpy()) < 1e-5
    assert rel_err(be.np(invstd), (1 / torch.sqrt(yr.var((0, 2, 3), unbiased=False) + 1e-5)).numpy()) < 1e-5

    # ---- data gradient: dL/d(vin) from r, scaled from max |r| as its producer would leave it
    rmax = _set_amax(be, None, np.abs(r).max())
    imgd, wmaxd = be.ws(be.lib.wsl_sp_weight_image_bytes(Ci, Co)), _amax_bits(be)
    be.call("wsl_sp_pack_weights", be.ptr(d["w"]), be.ptr(imgd), be.ptr(wmaxd), Ci, Co, 1, be.stream)
    sr = be.src(d["r"], Co)
    dx = be.zeros((N, Ci, H, W))
    assert be.lib.wsl_sp_conv2d_ok(sr, be.src(), be.ptr(dx), Ci * H * W, N, H, W, Ci, 3) == 1
    be.call("wsl_sp_conv2d_fwd", sr, be.src(), be.ptr(imgd), be.ptr(wmaxd), be.ptr(rmax), None, be.ptr(dx), Ci * H * W, N, H, W, Ci,
            None, None, be.stream)
    assert close(be.np(dx), vin.grad.numpy(), TOL)
    # a scale chosen 2^6 too small (an out-of-date maximum) changes nothing but the low halves that turn subnormal
    rmax2 = _set_amax(be, None, np.abs(r).max() * 64)
    dx2 = be.zeros((N, Ci, H, W))
    be.call("wsl_sp_conv2d_fwd", sr, be.src(), be.ptr(imgd), be.ptr(wmaxd), be.ptr(rmax2), None, be.ptr(dx2), Ci * H * W, N, H, W, Ci,
            None, None, be.stream)
    assert rel_err(be.np(dx2), be.np(dx)) < 1e-6

    # ---- weight / bias gradient
    nws = be.lib.wsl_sp_conv2d_wgrad_ws_bytes(N, H, W, Ci, Co)
    ws, dw, db = be.ws(nws), be.zeros((Co, Ci, 3, 3)), be.zeros((Co,))
    from wsl4mis_amd import _lib
    pend = _lib.WslWgradPending()
    import ctypes as C
    be.call("wsl_sp_conv2d_wgrad_partial", sa, sb, be.ptr(d["r"]), Co * H * W, be.ptr(rmax), be.ptr(dw), be.ptr(db), N, H, W, Co,
            be.ptr(ws), nws, C.byref(pend), be.stream)
    be.call("wsl_wgrad_reduce_batch", C.byref(pend), 1, be.stream)
    assert close(be.np(dw), wt.grad.numpy(), TOL)
    assert close(be.np(db), bt.grad.numpy(), TOL)


# (N, H, W, Ca, Cb, Co, persistent workgroups forced): workgroups that walk several tiles each in the interleaved, XCD-grouped order
# (splits % 8 == 0) and in contiguous runs (otherwise); 32 x 32 and 16 x 16 channel blocks, tiles 4 x 32 and 8 x 16
SP_WALKS = [(3, 16, 64, 32, 0, 32, 8), (5, 8, 32, 16, 0, 16, 8), (3, 16, 16, 32, 32, 32, 16), (3, 16, 64, 32, 0, 32, 5),
            (2, 16, 64, 64, 0, 64, 32)]


@pytest.mark.parametrize("case", SP_WALKS)
def test_sp_wgrad_tile_walk_orders(be_route, case):
    be = be_route   # forces the number of persistent workgroups: emulator / experiments build
    import ctypes as C
    from wsl4mis_amd import _lib
    N, H, W, Ca, Cb, Co, wgs = case
    Ci = Ca + Cb
    rng = np.random.default_rng(sum(case))
    xa = rng.standard_normal((N, Ca, H, W)).astype(np.float32)
    xb = rng.standard_normal((N, Cb, H, W)).astype(np.float32) if Cb else None
    scale, shift = (rng.standard_normal(Ca) * 0.5 + 1).astype(np.float32), (rng.standard_normal(Ca) * 0.3).astype(np.float32)
    va = virt_input(xa, scale, shift, None, 1.0, None)
    vin = torch.cat([va, torch.from_numpy(xb)], 1) if Cb else va
    wt, bt = torch.zeros(Co, Ci, 3, 3, requires_grad=True), torch.zeros(Co, requires_grad=True)
    r = (rng.standard_normal((N, Co, H, W)) * 3e-5).astype(np.float32)
    (F.conv2d(vin, wt, bt, padding=1) * torch.from_numpy(r)).sum().backward()
    d = {k: (be.arr(v) if v is not None else None) for k, v in dict(xa=xa, xb=xb, scale=scale, shift=shift, r=r).items()}
    sa = be.src(d["xa"], Ca, scale=d["scale"], shift=d["shift"])
    sb = be.src(d["xb"], Cb) if Cb else be.src()
    rmax = _set_amax(be, None, np.abs(r).max())
    be.call("wsl_debug_wgrad_workgroups", wgs)
    try:
        nws = be.lib.wsl_sp_conv2d_wgrad_ws_bytes(N, H, W, Ci, Co)
        ws, dw, db = be.ws(nws), be.zeros((Co, Ci, 3, 3)), be.zeros((Co,))
        pend = _lib.WslWgradPending()
        be.call("wsl_sp_conv2d_wgrad_partial", sa, sb, be.ptr(d["r"]), Co * H * W, be.ptr(rmax), be.ptr(dw), be.ptr(db), N, H, W, Co,
                be.ptr(ws), nws, C.byref(pend), be.stream)
        be.call("wsl_wgrad_reduce_batch", C.byref(pend), 1, be.stream)
        assert close(be.np(dw), wt.grad.numpy(), TOL)
        assert close(be.np(db), bt.grad.numpy(), TOL)
    finally:
        be.call("wsl_debug_wgrad_workgroups", 0)


def test_sp_raw_source_beyond_the_static_range(be):
    """the second source of a decoder block's first convolution is the RAW upsampled tensor (networks/unet.py:63-68) -- not
    BatchNorm-normalised, so nothing bounds it by the static activation scale 2^4 (|v| < 4094).  Its maximum is tracked by the
    upsampling pass (wsl_bilinear_up2_fwd_amax: max |u| bounds the bilinear output) and handed to the forward conv and the weight
    gradient as `in_amax`: results stay at the 1e-4 criterion for values far beyond 4094, and are bit-identical to the untracked call
    when the source is small."""
    import ctypes as C
    from wsl4mis_amd import _lib
    rng = np.random.default_rng(77)
    N, H, W, Ca, Cb, Co = 2, 8, 32, 16, 16, 32
    for big in (False, True):
        h, w_ = H // 2, W // 2
        u = (rng.standard_normal((N, Cb, h, w_)) * (3.0e4 if big else 2.0)).astype(np.float32)       # the 1x1-conv output that is upsampled
        xa = rng.standard_normal((N, Ca, H, W)).astype(np.float32)
        scale, shift = (rng.standard_normal(Ca) * 0.5 + 1).astype(np.float32), (rng.standard_normal(Ca) * 0.3).astype(np.float32)
        wgt = (rng.standard_normal((Co, Ca + Cb, 3, 3)) * 0.07).astype(np.float32)
        r = (rng.standard_normal((N, Co, H, W)) * 3e-5).astype(np.float32)
        d = {k: be.arr(v) for k, v in dict(u=u, xa=xa, scale=scale, shift=shift, w=wgt, r=r).items()}
        up, slots = be.zeros((N, Cb, H, W)), be.arr(np.zeros(32, np.int64))
        nws = be.lib.wsl_bilinear_up2_fwd_amax_ws_bytes(N, Cb, h, w_)
        ws = be.ws(nws)
        be.call("wsl_bilinear_up2_fwd_amax", be.ptr(d["u"]), be.ptr(up), Cb * H * W, N, Cb, h, w_, be.ptr(ws), nws, be.ptr(slots), be.stream)
        up_np = be.np(up)
        ref_up = F.interpolate(torch.from_numpy(u), scale_factor=2, mode="bilinear", align_corners=True).numpy()
        assert close(up_np, ref_up, 1e-5)
        tracked = be.np(slots).view(np.uint32).view(np.float32).max()
        assert tracked == np.abs(u).max() and tracked >= np.abs(up_np).max()
        va = virt_input(xa, scale, shift, None, 1.0, None)
        vin = torch.cat([va, torch.from_numpy(up_np)], 1).requires_grad_()
        wt = torch.from_numpy(wgt).requires_grad_()
        y_ref = F.conv2d(vin, wt, None, padding=1)
        (y_ref * torch.from_numpy(r)).sum().backward()
        sa, sb = be.src(d["xa"], Ca, scale=d["scale"], shift=d["shift"]), be.src(up, Cb)
        img, wmax = be.ws(be.lib.wsl_sp_weight_image_bytes(Co, Ca + Cb)), _amax_bits(be)
        be.call("wsl_sp_pack_weights", be.ptr(d["w"]), be.ptr(img), be.ptr(wmax), Co, Ca + Cb, 0, be.stream)
        ys = []
        for amax in (be.ptr(slots), None):
            y = be.zeros((N, Co, H, W))
            be.call("wsl_sp_conv2d_fwd", sa, sb, be.ptr(img), be.ptr(wmax), amax, None, be.ptr(y), Co * H * W, N, H, W, Co, None, None, be.stream)
            ys.append(be.np(y).copy())
        assert close(ys[0], y_ref.detach().numpy(), TOL), rel_err(ys[0], y_ref.detach().numpy())
        if big:
            assert not close(ys[1], y_ref.detach().numpy(), TOL)          # the untracked call saturates: what the tracking is for
        else:
            assert np.array_equal(ys[0], ys[1])                            # inside the static range nothing changes
        rmax = _set_amax(be, None, np.abs(r).max())
        nwg = be.lib.wsl_sp_conv2d_wgrad_ws_bytes(N, H, W, Ca + Cb, Co)
        wsg, dw, db = be.ws(nwg), be.zeros((Co, Ca + Cb, 3, 3)), be.zeros((Co,))
        pend = _lib.WslWgradPending()
        be.call("wsl_sp_conv2d_wgrad_partial_amax", sa, sb, be.ptr(d["r"]), Co * H * W, be.ptr(rmax), be.ptr(slots), be.ptr(dw), be.ptr(db),
                N, H, W, Co, be.ptr(wsg), nwg, C.byref(pend), be.stream)
        be.call("wsl_wgrad_reduce_batch", C.byref(pend), 1, be.stream)
        assert close(be.np(dw), wt.grad.numpy(), TOL), rel_err(be.np(dw), wt.grad.numpy())


def test_sp_not_eligible(be):
    x = be.zeros((1, 8, 8, 32))
    s = be.src(x, 8)
    assert be.lib.wsl_sp_conv2d_ok(s, be.src(), None, 0, 1, 8, 32, 16, 3) == 0        # Ci % 16
    x2 = be.zeros((1, 16, 8, 32))
    s2 = be.src(x2, 16)
    assert be.lib.wsl_sp_conv2d_ok(s2, be.src(), None, 0, 1, 8, 32, 16, 1) == 0       # 1x1
    assert be.lib.wsl_sp_conv2d_ok(s2, be.src(), None, 0, 1, 8, 32, 16, 3) == 1
    # BatchNorm coefficients are fetched as float4 pairs: a scale / shift array off a 16-byte boundary leaves the layer to the f32 kernels
    coef = be.zeros((40,))
    s3 = be.src(x2, 16, scale=coef, shift=coef)
    assert be.lib.wsl_sp_conv2d_ok(s3, be.src(), None, 0, 1, 8, 32, 16, 3) == 1
    s3.scale = be.ptr(coef) + 4
    assert be.lib.wsl_sp_conv2d_ok(s3, be.src(), None, 0, 1, 8, 32, 16, 3) == 0
    with pytest.raises(Exception, match="eligible"):
        be.call("wsl_sp_conv2d_fwd", s, be.src(), be.ptr(x), be.ptr(x), None, None, be.ptr(x), 16 * 8 * 32, 1, 8, 32, 16, None, None,
                be.stream)


def test_lds_residency_of_the_steps_layer_shapes(be):
    """LDS is handed out in units of 1280 bytes on gfx950 (round 4: a 256-byte table took the 16-wide blocks of 32 input channels from 42
    to 43 units and with that from three workgroups per CU to two: +21 % on that layer).  The layer shapes of the benchmark step that sit at
    such an edge, through the library's own residency computation: 16-wide blocks three per CU, everything else two; and nothing the step
    launches is left with ONE workgroup per CU."""
    import ctypes as C
    f = be.lib.wsl_debug_sp_conv_residency
    f.restype = C.c_int
    f.argtypes = [C.c_int] * 6 + [C.POINTER(C.c_int)] * 3 + [C.POINTER(C.c_size_t), C.POINTER(C.c_int)]

    def residency(N, H, W, Ci, Co, bn=0):
        th, tw, cot, per = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        lds = C.c_size_t()
        assert f(N, H, W, Ci, Co, bn, C.byref(th), C.byref(tw), C.byref(cot), C.byref(lds), C.byref(per)) == 0
        return th.value, tw.value, cot.value, lds.value, per.value

    # the full-resolution 16-wide blocks: exactly 42 units with 32 input channels (the coefficient table lives in the image planes' padding)
    assert residency(64, 256, 256, 16, 16)[2:] == (16, 43520, 3)
    th, tw, cot, lds, per = residency(64, 256, 256, 32, 16)
    assert (cot, lds, per) == (16, 42 * 1280, 3), (cot, lds, per)
    enc = [(16, 32, 128), (32, 32, 128), (32, 64, 64), (64, 64, 64), (64, 128, 32), (128, 128, 32), (128, 256, 16), (256, 256, 16)]
    dec = [(256, 128, 32), (128, 64, 64), (64, 32, 128)]
    for Ci, Co, S in enc + dec:
        for ci, co in ((Ci, Co), (Co, Ci)):                      # forward and data-gradient launch of the layer
            for bn in (0, 1):
                cot, lds, per = residency(64, S, S, ci, co, bn)[2:]
                assert per >= 2 and per * ((lds + 1279) // 1280) * 1280 <= 160 * 1024, (ci, co, S, bn, cot, lds, per)
    assert f(64, 256, 256, 20, 16, 0, None, None, None, None, None) == 1      # no split kernel for this shape
