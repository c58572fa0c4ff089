"""Whole-network parity through the C ABI (wsl_net_forward / wsl_head_fwd_bwd / wsl_net_backward) against the golden
vectors generated from the reference's own UNet / UNet_CCT + ours_proposed loss (tests/golden/make_golden.py)."""
import ctypes as C

import numpy as np
import pytest

from conftest import close, golden, grad_tol, labelmap_mismatch, mixed_err, rel_err
from netutil import check_grads, det_arenas, grad_l2, net_desc, ptr_array, strict_grads

TOL = 1e-4


def run_net(be, tag, net, full=True, precision=0):
    g = golden(f"g2_{tag}")
    N, H, W = (int(v) for v in g["cfg"])
    d = net_desc(net, N, H, W, precision=precision)
    params, bufs, nbt, ents = det_arenas(be.lib, d, 2022)
    dp, db, dn = be.arr(params), be.arr(bufs), be.arr(nbt)
    x, lab = be.arr(g["x"]), be.arr(g["label"])
    em = [be.arr(g[f"emask{i}"]) for i in range(5)]
    cm = [be.arr(g[f"cmask{i}"]) for i in range(5)] if net == "unet_cct" else None
    pem, pcm = ptr_array(be, em), ptr_array(be, cm)
    nws = be.lib.wsl_net_ws_bytes(C.byref(d))
    ws = be.ws(nws)
    lm = be.zeros((N, 4, H, W))
    la = be.zeros((N, 4, H, W)) if cm else None
    be.call("wsl_net_forward", C.byref(d), be.ptr(dp), be.ptr(db), be.ptr(dn), be.ptr(x), pem, pcm, 1, be.ptr(lm),
            be.ptr(la) if cm else None, be.ptr(ws), nws, be.stream)
    assert close(be.np(lm), g["logits_main"], TOL), (rel_err(be.np(lm), g["logits_main"]), mixed_err(be.np(lm), g["logits_main"]))
    if cm:
        assert close(be.np(la), g["logits_aux"], TOL), (rel_err(be.np(la), g["logits_aux"]), mixed_err(be.np(la), g["logits_aux"]))
    # loss head
    out, pseudo = be.zeros((4,)), be.zeros((N, H, W), np.int64)
    dz1, dz2 = be.zeros((N, 4, H, W)), be.zeros((N, 4, H, W))
    nl = be.lib.wsl_loss_ws_bytes(N, 4, H * W)
    lws = be.ws(nl)
    be.call("wsl_head_fwd_bwd", be.ptr(lm), be.ptr(la) if cm else None, be.ptr(lab), 4, float(g["beta"]), 0.5, 1.0,
            be.ptr(out), be.ptr(pseudo) if cm else None, be.ptr(dz1), be.ptr(dz2) if cm else None, N, 4, H * W,
            be.ptr(lws), nl, be.stream)
    o = be.np(out)
    assert rel_err(o[0], g["loss_parts"][0]) < TOL
    if cm:
        assert rel_err(o[1:3], g["loss_parts"][1:3]) < TOL
        labelmap_mismatch(f"test_net {tag} pseudo-label map ({be.name})", be.np(pseudo), g["pseudo"], allow_px=2)
    grads = be.zeros(params.shape)
    be.call("wsl_net_backward", C.byref(d), be.ptr(dp), be.ptr(x), pem, pcm, be.ptr(dz1), be.ptr(dz2) if cm else None,
            be.ptr(grads), be.ptr(ws), nws, 0, be.stream)
    if strict_grads(g):
        bad = check_grads(g, be.np(grads), ents, grad_tol, prefix="g.")
        assert not bad, bad[:8]
    else:
        # A pre-activation of this fixture lies within fp32 noise of the LeakyReLU kink (margins recorded by the
        # generator): one sign flip makes the gradient jump, so element-wise 1e-4 parity is undefined here -- for any two
        # fp32 implementations.  The forward above stays strict; the gradients get a kink-tolerant bound.
        assert grad_l2(g, be.np(grads), ents) < 2e-2
    # BatchNorm buffers after the step
    hb = be.np(db)
    for n, kind, shape, off in ents:
        if kind == 1:
            assert rel_err(hb[off:off + shape[0]], g[f"b.{n}"]) < TOL, n
    assert np.all(be.np(dn) == 4)
    if not full:      # (CPU time on the emulator: the eval forward and the split backward are covered by the other cases)
        return
    # eval-mode forward (running stats, dropout off; the aux branch's dropout2d stays on in the reference, but the
    # golden eval output is the main branch, which does not see it)
    lm2 = be.zeros((N, 4, H, W))
    cm_zero = [be.zeros(tuple(c.shape)) for c in cm] if cm else None     # keep alive while the call runs
    cm0 = ptr_array(be, cm_zero)
    be.call("wsl_net_forward", C.byref(d), be.ptr(dp), be.ptr(db), be.ptr(dn), be.ptr(x), None, cm0, 0, be.ptr(lm2),
            be.ptr(la) if cm else None, be.ptr(ws), nws, be.stream)
    assert rel_err(be.np(lm2), g["logits_eval"]) < TOL
    # split backward (decoders, then encoder) gives the same gradients as phase 0
    dp2, db2, dn2 = be.arr(params), be.arr(bufs), be.arr(nbt)
    be.call("wsl_net_forward", C.byref(d), be.ptr(dp2), be.ptr(db2), be.ptr(dn2), be.ptr(x), pem, pcm, 1, be.ptr(lm),
            be.ptr(la) if cm else None, be.ptr(ws), nws, be.stream)
    grads2 = be.zeros(params.shape)
    for phase in (1, 2):
        be.call("wsl_net_backward", C.byref(d), be.ptr(dp), be.ptr(x), pem, pcm, be.ptr(dz1), be.ptr(dz2) if cm else None,
                be.ptr(grads2), be.ptr(ws), nws, phase, be.stream)
    assert np.array_equal(be.np(grads), be.np(grads2))


def test_unet_cct_16_emul_and_gpu(be):
    run_net(be, "cct16", "unet_cct", full=(be.name == "hip"))


def test_unet_32_emul_and_gpu(be):
    run_net(be, "unet32", "unet")


@pytest.mark.gpu
def test_unet_cct_32_gpu():
    from conftest import get_backend
    run_net(get_backend("hip"), "cct32", "unet_cct")


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["cct64", "cct48x80"])
def test_unet_cct_larger_gpu(tag):
    from conftest import get_backend
    run_net(get_backend("hip"), tag, "unet_cct")


# ---- the same fixtures on the split-precision conv path (f16 hi / lo operands, three MFMA passes, fp32 accumulate): same criteria
def test_unet_cct_16_split_emul_and_gpu(be):
    run_net(be, "cct16", "unet_cct", full=(be.name == "hip"), precision=1)


@pytest.mark.gpu
@pytest.mark.parametrize("tag,net", [("unet32", "unet"), ("cct32", "unet_cct"), ("cct64", "unet_cct"), ("cct48x80", "unet_cct")])
def test_net_split_gpu(tag, net):
    from conftest import get_backend
    run_net(get_backend("hip"), tag, net, precision=1)


def test_layout_matches_reference_state_dict(be):
    g = golden("g0_init")
    from netutil import entries
    for net in ("unet", "unet_cct"):
        d = net_desc(net, 1, 16, 16)
        ents = entries(be.lib, d)
        assert [e[0] for e in ents] == list(g[f"{net}_keys"])
        assert [str(tuple(int(v) for v in e[2])) for e in ents] == list(g[f"{net}_shapes"])
    assert be.lib.wsl_net_param_count(C.byref(net_desc("unet_cct", 1, 16, 16))) == 2447064
    assert be.lib.wsl_net_param_count(C.byref(net_desc("unet", 1, 16, 16))) == 1813764
    assert be.lib.wsl_net_encoder_param_count(C.byref(net_desc("unet_cct", 1, 16, 16))) == 1180464
