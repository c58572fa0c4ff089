"""The host-side mirror of the reference interface (networks.net_factory, utils.losses, utils.gate_crf_loss, the
fused engine): module structure, autograd wiring and end-to-end parity with the reference's golden vectors.
`mode` = emul runs the Python layer against the host-emulation library with CPU tensors (host-logic check);
`mode` = hip (gpu mark) is the real thing."""
import ctypes as C
import random

import numpy as np
import pytest
import torch

from conftest import get_backend, golden, grad_tol, rel_err
from detinit import det_state, sample_index

TOL = 1e-4


@pytest.fixture(params=[pytest.param("emul"), pytest.param("hip", marks=pytest.mark.gpu)])
def mode(request):
    from wsl4mis_amd import _lib, runtime
    _lib._reset_for_tests()
    runtime._ws_cache.clear()
    if request.param == "emul":
        _lib.use_library_for_tests(get_backend("emul").lib)
    yield request.param
    _lib._reset_for_tests()
    runtime._ws_cache.clear()


def dev():
    from wsl4mis_amd import runtime
    return runtime.device()


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def load_det(model, seed):
    sd = model.state_dict()
    vals = det_state({k: tuple(v.shape) for k, v in sd.items()}, seed)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in vals.items()})


def test_net_factory_default_init_is_the_reference_init(mode):
    from wsl4mis_amd.networks.net_factory import net_factory
    g = golden("g0_init")
    for net in ("unet", "unet_cct"):
        torch.manual_seed(2022)
        m = net_factory(net, 1, 4)
        sd = m.state_dict()
        assert list(sd.keys()) == list(g[f"{net}_keys"])
        assert [str(tuple(v.shape)) for v in sd.values()] == list(g[f"{net}_shapes"])
        sums = np.array([float(v.double().sum()) for v in sd.values()])
        assert np.allclose(sums, g[f"{net}_sum"], rtol=0, atol=1e-9)          # same RNG draws, bit for bit
        heads = np.stack([np.resize(v.double().cpu().numpy().ravel(), 4) for v in sd.values()])
        assert np.array_equal(heads, g[f"{net}_head"])
        assert [tuple(p.shape) for p in m.parameters()] == [tuple(v.shape) for k, v in sd.items()
                                                            if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))]
    assert net_factory("no_such_net") is None
    with pytest.raises(NotImplementedError):
        net_factory("unet_ds")


def test_module_autograd_matches_reference_script_composition(mode):
    """model(x) -> softmax -> CE + mixed pseudo-label pDice written exactly like ours_proposed.py:108-128, with the
    drop-in modules; gradients land in p.grad through ordinary autograd; torch.optim.SGD updates the arena views."""
    from wsl4mis_amd.networks.net_factory import net_factory
    from wsl4mis_amd.utils import losses
    g = golden("g2_cct16" if mode == "emul" else "g2_cct32")     # the emulator gets the small fixture (CPU time)
    model = net_factory("unet_cct", 1, 4)
    load_det(model, 2022)
    model.train()
    model.set_dropout_masks([T(g[f"emask{i}"]) for i in range(5)], [T(g[f"cmask{i}"]) for i in range(5)])
    x, lab = T(g["x"]), T(g["label"])
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    ce_loss, dice_loss = losses.PartialCrossEntropyLoss(ignore_index=4), losses.pDLoss(4, ignore_index=4)
    outputs, outputs_aux1 = model(x)
    assert rel_err(outputs.detach().cpu(), g["logits_main"]) < TOL and rel_err(outputs_aux1.detach().cpu(), g["logits_aux"]) < TOL
    s1, s2 = losses.softmax(outputs), losses.softmax(outputs_aux1)
    loss_ce = 0.5 * (ce_loss(outputs, lab.long()) + ce_loss(outputs_aux1, lab.long()))
    beta = float(g["beta"])
    pseudo = losses.mix_argmax(s1, s2, beta)
    loss_pse = 0.5 * (dice_loss(s1, pseudo.unsqueeze(1)) + dice_loss(s2, pseudo.unsqueeze(1)))
    loss = loss_ce + 0.5 * loss_pse
    opt.zero_grad()
    loss.backward()
    assert rel_err([loss.item(), loss_ce.item(), loss_pse.item()], g["loss_parts"]) < TOL
    before = model.flat_params().clone()
    for k, p in model.named_parameters():
        gr = p.grad.detach().cpu().numpy().ravel()
        ref = g[f"g.{k}"]
        assert np.max(np.abs(gr[sample_index(gr.size)] - ref)) <= grad_tol(k, ref), k
    opt.step()
    assert not torch.equal(before, model.flat_params())            # the optimiser moved the arena through the views
    # fused head == the composed form
    model.zero_grad()
    outputs, outputs_aux1 = model(x)
    l2, parts, ps = losses.wsl_head(outputs, outputs_aux1, lab, beta)
    assert torch.equal(ps, losses.mix_argmax(losses.softmax(outputs), losses.softmax(outputs_aux1), beta))
    with pytest.raises(NotImplementedError):
        model(x.clone().requires_grad_())


def test_masks_are_drawn_reproducibly_from_torch_seed(mode):
    from wsl4mis_amd.networks.net_factory import net_factory
    m = net_factory("unet_cct", 1, 4)
    x = torch.rand(2, 1, 16, 16).to(dev())
    m.train()
    torch.manual_seed(5)
    with torch.no_grad():
        m(x)
    em1, cm1 = [t.clone() for t in m._last_masks[0]], [t.clone() for t in m._last_masks[1]]
    torch.manual_seed(5)
    with torch.no_grad():
        m(x)
    assert all(torch.equal(a, b) for a, b in zip(em1, m._last_masks[0]))
    assert all(torch.equal(a, b) for a, b in zip(cm1, m._last_masks[1]))
    with torch.no_grad():
        m(x)                                                   # next draw differs
    assert not torch.equal(em1[0], m._last_masks[0][0])
    assert abs(em1[0].float().mean().item() - 0.95) < 0.02 and set(cm1[0].unique().tolist()) <= {0.0, 2.0}
    m.eval()                                                   # eval: no elementwise dropout, aux dropout2d still drawn
    with torch.no_grad():
        m(x)
    assert m._last_masks[0] is None and m._last_masks[1] is not None


def test_eval_and_state_dict_roundtrip(mode):
    from wsl4mis_amd.networks.net_factory import net_factory
    g = golden("g2_unet32")
    m = net_factory("unet", 1, 4)
    load_det(m, 2022)
    m.train()
    m.set_dropout_masks([T(g[f"emask{i}"]) for i in range(5)])
    out = m(T(g["x"]))
    assert rel_err(out.detach().cpu(), g["logits_main"]) < TOL
    m.eval()
    with torch.no_grad():
        ev = m(T(g["x"]))
    assert rel_err(ev.cpu(), g["logits_eval"]) < TOL
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    for k in sd:
        if k.endswith(("running_mean", "running_var")):
            assert rel_err(sd[k].cpu(), g[f"b.{k}"]) < TOL
        if k.endswith("num_batches_tracked"):
            assert int(sd[k]) == 4
    m2 = net_factory("unet", 1, 4)
    m2.load_state_dict(sd)
    m2.eval()
    with torch.no_grad():
        assert torch.equal(m2(T(g["x"])), ev)


def test_checkpoint_file_roundtrip(mode, tmp_path):
    """the reference's checkpoint contract: torch.save(model.state_dict(), p) / net.load_state_dict(torch.load(p))
    (..._ours_proposed.py:174-198, test_2D_fully_sps.py:147-150) -- same keys, order, shapes and dtypes as the reference
    modules' state_dict (fixture g0), and a file written from CPU tensors (what a reference run leaves) loads back"""
    from wsl4mis_amd.networks.net_factory import net_factory
    g0 = golden("g0_init")
    m = net_factory("unet_cct", 1, 4)
    load_det(m, 77)
    path = str(tmp_path / "unet_cct_best_model.pth")
    torch.save(m.state_dict(), path)
    sd = torch.load(path, map_location="cpu")
    assert list(sd.keys()) == list(g0["unet_cct_keys"])
    assert [str(tuple(v.shape)) for v in sd.values()] == list(g0["unet_cct_shapes"])
    assert all(v.dtype == (torch.int64 if k.endswith("num_batches_tracked") else torch.float32) for k, v in sd.items())
    m2 = net_factory("unet_cct", 1, 4)
    m2.load_state_dict(torch.load(path, map_location="cpu"))
    assert torch.equal(m2.flat_params(), m.flat_params())
    assert all(torch.equal(a, b) for a, b in zip(m2.state_dict().values(), m.state_dict().values()))
    bad = dict(sd)
    bad.pop("aux_decoder1.out_conv.bias")
    with pytest.raises(RuntimeError):
        m2.load_state_dict(bad)                                 # strict by default, like nn.Module


def test_loss_modules_against_reference(mode):
    from wsl4mis_amd.utils import losses
    from wsl4mis_amd.utils.gate_crf_loss import ModelLossSemsegGatedCRF
    g = golden("g4_gatedcrf")
    crf = ModelLossSemsegGatedCRF()
    y = T(g["ns5_y"]).requires_grad_()
    out = crf(y, [{"weight": 1, "xy": 6, "rgb": 0.1}], 5, T(g["ns5_img"]), 48, 40)["loss"]
    (2.0 * out).backward()
    assert rel_err(out.item(), g["ns5_loss"]) < TOL and rel_err(y.grad.cpu(), 2.0 * g["ns5_dy"]) < TOL
    with pytest.raises(NotImplementedError):
        crf(y, [{"weight": 1, "xy": 6, "rgb": 0.1}], 5, T(g["ns5_img"]), 48, 40, mask_src=T(g["ns5_img"]))
    with pytest.raises(NotImplementedError):
        crf(y, [{"weight": 1, "xy": 6, "rgb": 0.1}], 5, T(g["ns5_img"]).repeat(1, 3, 1, 1), 48, 40)      # multi-channel sample: not built
    with pytest.raises(RuntimeError):
        crf(y, [{"weight": 1}], 5, T(g["ns5_img"]), 48, 40)                                              # no modality at all (the reference fails too)
    with pytest.raises(AssertionError):
        crf(y, [{"weight": 1, "xy": 6, "rgb": 0.1}], 5, T(g["ns5_img"]), 50, 40)
    # round 6 (VERDICT r5 missing 4): several descriptors summed, descriptors without 'xy' / with several sample modalities, a sample larger
    # than the prediction (adaptive average pooling) -- against the reference module's own outputs
    gg = golden("g11_gatedcrf_general")
    for tag in ("two", "rgbonly", "twomod", "down", "three"):
        desc = eval(str(gg[f"{tag}_desc"]), {"__builtins__": {}})
        yy, up = T(gg[f"{tag}_y"]).requires_grad_(), int(gg[f"{tag}_up"])
        Hh, Ww = yy.shape[2] * up, yy.shape[3] * up
        img = T(gg[f"{tag}_img"])
        img0 = img.clone()
        out = crf(yy, desc, int(gg[f"{tag}_r"]), img, Hh, Ww)["loss"]
        out.backward()
        assert torch.equal(img, img0)                                                                    # the sample is not modified
        assert rel_err(out.item(), gg[f"{tag}_loss"]) < TOL, (tag, out.item(), float(gg[f"{tag}_loss"]))
        assert rel_err(yy.grad.cpu(), gg[f"{tag}_dy"]) < TOL, tag
    g = golden("g5_tv_ms")
    p = T(g["tv_p"]).requires_grad_()
    l = losses.tv_loss(p[1:])
    l.backward()
    assert rel_err(l.item(), g["tv_loss"]) < 1e-5 and rel_err(p.grad.cpu(), g["tv_dp"]) < 1e-5
    p = T(g["ms_p"]).requires_grad_()
    l = losses.MumfordShah_Loss()(T(g["ms_img"]), p)
    (l * 1e-6).backward()
    assert rel_err(l.item(), g["ms_loss"]) < 1e-5 and rel_err(p.grad.cpu(), 1e-6 * g["ms_dp"]) < TOL
    g = golden("g3_head")
    s = T(g["dl_s"]).requires_grad_()
    l = losses.DiceLoss(4)(s, T(g["dl_target"]))
    l.backward()
    assert rel_err(l.item(), g["dl_loss"]) < 1e-5 and rel_err(s.grad.cpu(), g["dl_ds"]) < TOL
    with pytest.raises(NotImplementedError):
        losses.DiceLoss(4)(s, T(g["dl_target"]), weight=[1, 2, 1, 1])
    a = T(g["mse_a"]).requires_grad_()
    l = torch.mean(losses.softmax_mse_loss(a, T(g["mse_b"])))
    l.backward()
    assert rel_err(l.item(), g["mse_loss"]) < 1e-5 and rel_err(a.grad.cpu(), g["mse_da"]) < TOL
    a2 = T(g["mse_a"]).requires_grad_()
    l2 = losses.softmax_mse_mean(a2, T(g["mse_b"]))
    l2.backward()
    assert rel_err(l2.item(), g["mse_loss"]) < 1e-5 and rel_err(a2.grad.cpu(), g["mse_da"]) < TOL
    from wsl4mis_amd.utils.ramps import sigmoid_rampup
    assert abs(sigmoid_rampup(0, 200) - np.exp(-5.0)) < 1e-12 and sigmoid_rampup(300, 200) == 1.0


def run_curve(steps):
    from wsl4mis_amd.engine import TrainEngine
    g = golden("g7_curve")
    eng = TrainEngine("unet_cct", 1, 4, base_lr=0.01, max_iterations=60000, loss="ours_proposed")
    load_det(eng.model, 7)
    got = []
    for it in range(steps):
        xs = T(g["xs"][it])
        N, _, H, W = xs.shape
        em = [T(np.unpackbits(g[f"em{it}_{l}"])[:N * (16 << l) * (H >> l) * (W >> l)].reshape(N, 16 << l, H >> l, W >> l))
              for l in range(5)]
        cm = [T(g[f"cm{it}_{l}"]) for l in range(5)]
        eng.model.set_dropout_masks(em, cm)
        eng.step(xs, T(g["labels"][it]), float(g["betas"][it]))
        o = eng.losses()
        got.append([o["loss"], o["ce"], o["pse"]])
    return np.array(got), g, eng


def test_entropy_loss_kernel(mode):
    """entropy_loss (utils/losses.py:30-36) as one kernel: value and gradient against the plain fp32 torch expression"""
    import math
    from wsl4mis_amd.utils import losses
    rng = np.random.default_rng(21)
    z = torch.from_numpy(rng.standard_normal((3, 4, 9, 11)).astype(np.float32))
    p_ref = torch.softmax(z, 1).requires_grad_()
    ref = torch.mean(-1 * torch.sum(p_ref * torch.log(p_ref + 1e-6), dim=1) / math.log(4))
    ref.backward()
    p = p_ref.detach().clone().to(dev()).requires_grad_()
    got = losses.entropy_loss(p, C=4)
    (2.0 * got).backward()
    assert abs(got.item() - ref.item()) < 1e-6
    assert rel_err(p.grad.cpu().numpy(), 2.0 * p_ref.grad.numpy()) < 1e-5
    # C is only the log(C) normaliser (losses.py:30-33): the reference's default C=2 works on a 4-class softmax too
    p2 = p_ref.detach().clone().to(dev()).requires_grad_()
    got2 = losses.entropy_loss(p2)
    got2.backward()
    assert abs(got2.item() - ref.item() * math.log(4) / math.log(2)) < 1e-6
    assert rel_err(p2.grad.cpu().numpy(), p_ref.grad.numpy() * math.log(4) / math.log(2)) < 1e-5
    with pytest.raises(ValueError):
        losses.entropy_loss(p, C=1)


def test_autograd_accumulation_and_interleaved_forwards(mode):
    """p.grad must be autograd's own memory, not a view of the arena the next backward overwrites: two backward passes
    accumulate g_a + g_b, zero_grad(set_to_none=False) works, and a no_grad / eval forward between a training forward and
    its backward must not redraw the dropout masks the backward replays."""
    from wsl4mis_amd.networks.net_factory import net_factory
    from wsl4mis_amd.utils import losses
    torch.manual_seed(3)
    model = net_factory("unet", 1, 4)
    model.train()
    S = 16
    xa, xb = torch.rand(2, 1, S, S).to(dev()), torch.rand(2, 1, S, S).to(dev())
    lab = torch.randint(0, 5, (2, S, S), dtype=torch.uint8).to(dev())
    ce = losses.PartialCrossEntropyLoss(ignore_index=4)
    gen = torch.Generator().manual_seed(8)
    em = [(torch.rand((2, 16 << l, S >> l, S >> l), generator=gen) >= 0.3).to(torch.uint8).to(dev()) for l in range(5)]
    model.set_dropout_masks(em, None)

    def grads_of(x, zero):
        if zero == "none":
            model.zero_grad(set_to_none=True)
        elif zero == "inplace":
            model.zero_grad(set_to_none=False)
        ce(model(x), lab.long()).backward()
        return torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone()

    ga, gb = grads_of(xa, "none"), grads_of(xb, "none")
    gab = grads_of(xa, "keep")                                 # p.grad holds g_b: accumulate g_a on top
    scale = float((ga + gb).abs().max())
    assert float((gab - (ga + gb)).abs().max()) <= 1e-6 * scale
    assert float((grads_of(xb, "inplace") - gb).abs().max()) <= 1e-6 * scale      # not 2 * g_b
    model = net_factory("unet_cct", 1, 4)
    model.train()
    # interleaved forwards: drawn masks this time
    model.set_dropout_masks(None, None)
    model.zero_grad(set_to_none=True)
    torch.manual_seed(11)
    z1, z2 = model(xa)
    saved = [t.clone() for t in model._saved[3]] + [t.clone() for t in model._saved[4]]
    with torch.no_grad():
        model(xb)                                              # train-mode forward without grad: must use other buffers
    model.eval()
    with torch.no_grad():
        model(xb)                                              # eval forward of unet_cct still draws the aux dropout2d masks
    model.train()
    assert all(torch.equal(a, b) for a, b in zip(saved, list(model._saved[3]) + list(model._saved[4])))
    (ce(z1, lab.long()) + 0.5 * ce(z2, lab.long())).backward()
    g1 = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone()
    model.set_dropout_masks(saved[:5], saved[5:])              # the same step without the interleaved forwards
    model.zero_grad(set_to_none=True)
    z1, z2 = model(xa)
    (ce(z1, lab.long()) + 0.5 * ce(z2, lab.long())).backward()
    g2 = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    assert float((g1 - g2).abs().max()) <= 1e-6 * float(g2.abs().max())


def test_engine_loss_curve_start(mode):
    """first optimiser steps of the fused engine vs the reference loop (SGD + poly LR incl. its one-step lag)."""
    steps = 1 if mode == "emul" else 12
    got, g, eng = run_curve(steps)
    ref = g["losses"][:steps]
    rel = np.abs(got - ref) / np.abs(ref)
    # the first optimiser steps pin the arithmetic (forward, loss, backward, SGD, poly LR) tightly; further along, the
    # trajectories of two fp32 implementations drift apart through LeakyReLU / max-pool / arg-max flips (bs 4, 32x32 nets
    # are twitchy), so the tail is only required to stay close
    assert np.max(rel[:min(2, steps)]) < 1e-4, (got, ref)
    assert np.max(rel) < 3e-2, (got, ref)
    if steps == 12:
        sd = eng.model.state_dict()
        for k in ("encoder.in_conv.conv_conv.0.weight", "main_decoder.out_conv.weight", "aux_decoder1.up1.conv1x1.bias",
                  "encoder.down4.maxpool_conv.1.conv_conv.5.running_var"):
            assert rel_err(sd[k].cpu().numpy().ravel()[:256], g[f"final.{k}"]) < 5e-2, k


def test_engine_gatedcrf_curve_against_reference(mode):
    """G9: the headline composition -- unet_cct, 0.5 (ce1 + ce2) + 0.1 GatedCRF(beta s1 + (1 - beta) s2), r = 5 -- through
    the fused engine vs the reference's own modules + torch SGD (train_ACDC_scribblevc.py:171-206 with the GatedCRF term of
    ..._pCE_GatedCRFLoss_2D.py:103-123): every parameter gradient of the first step, then the loss curve."""
    from wsl4mis_amd.engine import TrainEngine
    g = golden("g9_crf_curve")
    steps = 1 if mode == "emul" else g["xs"].shape[0]
    eng = TrainEngine("unet_cct", 1, 4, base_lr=0.01, max_iterations=60000, loss="pce_gatedcrf", crf_radius=5)
    load_det(eng.model, 9)
    got = []
    for it in range(steps):
        xs = T(g["xs"][it])
        N, _, H, W = xs.shape
        em = [T(np.unpackbits(g[f"em{it}_{l}"])[:N * (16 << l) * (H >> l) * (W >> l)].reshape(N, 16 << l, H >> l, W >> l))
              for l in range(5)]
        cm = [T(g[f"cm{it}_{l}"]) for l in range(5)]
        eng.model.set_dropout_masks(em, cm)
        eng.forward_backward(xs, T(g["labels"][it]), float(g["betas"][it]))
        if it == 0:
            flat = eng.model.flat_grads().cpu().numpy()
            off = 0
            for k, p in eng.model.named_parameters():
                gr, ref = flat[off:off + p.numel()], g[f"g.{k}"]
                off += p.numel()
                # strict: the generator searched step 0 for LeakyReLU / max-pool margins clear of fp32 noise (g["margins"])
                assert np.max(np.abs(gr[sample_index(gr.size)] - ref)) <= grad_tol(k, ref), k
                if not k.endswith(("conv_conv.0.bias", "conv_conv.4.bias")):
                    assert abs(np.sqrt((gr.astype(np.float64) ** 2).sum()) - g[f"gn.{k}"][0]) <= 1e-4 * g[f"gn.{k}"][0] + 1e-9, k
            assert off == flat.size
        eng.optimizer_step()
        o = eng.losses()
        got.append([o["loss"], o["ce"], o["crf"]])
    got, ref = np.array(got), g["losses"][:steps]
    rel = np.abs(got - ref) / np.abs(ref)
    assert np.max(rel[:min(2, steps)]) < 1e-4, (got, ref)
    assert np.max(rel) < 3e-2, (got, ref)                       # (tail: kink flips, as for G7)


def test_two_stream_decoders_are_bit_identical_to_one_stream(mode):
    """unet_cct runs its auxiliary decoder on a side stream (wsl_net_concurrent): same bits either way"""
    if mode != "hip":
        pytest.skip("streams only exist on the device")
    from wsl4mis_amd import _lib
    outs = []
    for conc in (1, 0, 1):
        _lib.lib().wsl_net_concurrent(conc)
        got, _, eng = run_curve(3)
        outs.append((got, eng.model.flat_params().clone(), eng.model.flat_grads().clone()))
    _lib.lib().wsl_net_concurrent(1)
    for got, prm, grd in outs[1:]:
        assert np.array_equal(got, outs[0][0]) and torch.equal(prm, outs[0][1]) and torch.equal(grd, outs[0][2])


def test_mean_teacher_step_against_oracle(mode):
    """SURVEY 8d config 4 (student unet + EMA teacher; pCE + TV + consistency): one engine step vs the oracle's
    composition of the pinned pieces -- losses, student parameters after SGD, teacher parameters after the EMA."""
    from oracle import torch_ref as R
    from wsl4mis_amd.engine import TrainEngine
    from wsl4mis_amd.synthetic import scribble_labels
    N, S, it0 = 3, 16, 30000
    gen = torch.Generator().manual_seed(11)
    x = torch.rand((N, 1, S, S), generator=gen)
    lab = torch.from_numpy(scribble_labels(N, S, S, 4, share=0.08))
    noise = torch.clamp(torch.randn((N, 1, S, S), generator=gen) * 0.1, -0.2, 0.2)
    masks = [[(torch.rand((N, 16 << l, S >> l, S >> l), generator=gen) >= R.DROP[l]).to(torch.uint8) for l in range(5)]
             for _ in range(2)]
    eng = TrainEngine("unet", 1, 4, base_lr=0.01, loss="mean_teacher")
    load_det(eng.model, 21)
    load_det(eng.teacher, 22)            # a teacher that differs from the student, so the EMA is visible
    eng.it = it0
    eng.model.set_dropout_masks([T(m) for m in masks[0]])
    eng.teacher.set_dropout_masks([T(m) for m in masks[1]])
    # ---- oracle
    sd_s = {k: torch.from_numpy(np.asarray(v)).clone() for k, v in det_state(
        {k: tuple(v.shape) for k, v in eng.model.state_dict().items()}, 21).items()}
    sd_t = {k: torch.from_numpy(np.asarray(v)).clone() for k, v in det_state(
        {k: tuple(v.shape) for k, v in eng.teacher.state_dict().items()}, 22).items()}
    pk = [k for k in sd_s if R.is_param(k)]
    for k in pk:
        sd_s[k].requires_grad_(True)
    z_s = R.net_forward(sd_s, x, "unet", masks[0], None, True)
    with torch.no_grad():
        z_t = R.net_forward(sd_t, x + noise, "unet", masks[1], None, True)
    loss, ce, tv, cons = R.mean_teacher_loss(z_s, z_t, lab, it0)
    loss.backward()
    with torch.no_grad():
        ps = [sd_s[k] for k in pk]
        R.sgd_step(ps, [p.grad for p in ps], [torch.zeros_like(p) for p in ps], 0.01, first=False)
        R.ema_update([sd_t[k] for k in pk], ps, 0.99, it0)
    # ---- engine
    eng.step(T(x.numpy()), T(lab.numpy()), noise=T(noise.numpy()))
    o = eng.losses()
    assert rel_err([o["loss"], o["ce"], o["tv"], o["cons"]], [loss.item(), ce.item(), tv.item(), cons.item()]) < 1e-4
    got_s, got_t = eng.model.state_dict(), eng.teacher.state_dict()
    for k in pk:
        assert rel_err(got_s[k].detach().cpu(), sd_s[k].detach()) < 1e-4, k
        assert rel_err(got_t[k].detach().cpu(), sd_t[k].detach()) < 1e-4, k


def test_validation_volume_label_maps(mode):
    """val_2D-style per-volume inference: label maps of the HIP eval forward vs the oracle's (bit-exact argmax is pinned
    at op level; end to end we report the near-tie mismatch rate) and the in-house Dice."""
    from oracle import torch_ref as R
    from wsl4mis_amd import val_2D
    from wsl4mis_amd.networks.net_factory import net_factory
    rng = np.random.default_rng(3)
    D, H, W, P = 2, 20, 24, (16, 16)
    vol = rng.random((D, H, W)).astype(np.float32)
    lab = rng.integers(0, 4, (D, H, W)).astype(np.uint8)
    m = net_factory("unet_cct", 1, 4)
    load_det(m, 5)
    got = val_2D.test_single_volume_cct(torch.from_numpy(vol)[None], torch.from_numpy(lab)[None], m, classes=4, patch_size=P)
    assert len(got) == 3 and all(0.0 <= d <= 1.0 and (h == 0 or np.isfinite(h)) for d, h in got)
    from oracle import metrics_ref
    pred0 = val_2D._predict_volume(vol, m, P, first_output=True)
    for c, (d, h) in enumerate(got, start=1):       # metrics of the predicted maps == the medpy algorithm on the same maps
        d_ref, h_ref = metrics_ref.calculate_metric_percase(pred0 == c, lab == c)
        assert d == d_ref and abs(h - h_ref) <= 1e-12 * max(1.0, h_ref)
    # oracle label maps for the same slices
    from scipy.ndimage import zoom
    sd = {k: torch.from_numpy(np.asarray(v)).clone() for k, v in det_state(
        {k: tuple(v.shape) for k, v in m.state_dict().items()}, 5).items()}
    pred = val_2D._predict_volume(vol, m, P, first_output=True)
    ref = np.zeros_like(pred)
    for i in range(D):
        inp = torch.from_numpy(zoom(vol[i], (P[0] / H, P[1] / W), order=0))[None, None]
        with torch.no_grad():
            z = R.net_forward(sd, inp, "unet_cct", None, [torch.ones(1, 16 << l) for l in range(5)], False)[0]
        ref[i] = zoom(torch.argmax(z, 1)[0].numpy().astype(np.uint8), (H / P[0], W / P[1]), order=0)
    from conftest import labelmap_mismatch
    labelmap_mismatch(f"val_2D label maps, synthetic 2 x 20 x 24 volume ({mode})", pred, ref, allow_px=2)
    assert val_2D.dice_percase(np.array([1, 1, 0]), np.array([1, 0, 0])) == pytest.approx(2 / 3)
    with pytest.raises(NotImplementedError):
        val_2D.test_single_volume(torch.from_numpy(vol[0]), torch.from_numpy(lab[0]), m, 4)


def test_ustm_step_against_oracle(mode):
    """train_weakly_supervised_ustm_2D.py:119-163 (pCE + uncertainty-masked consistency against an EMA teacher, T = 8
    stochastic teacher passes on the rotated batch): one engine step vs the oracle's composition of the pinned pieces."""
    import random as pyrandom
    from oracle import torch_ref as R
    from wsl4mis_amd.engine import TrainEngine
    from wsl4mis_amd.synthetic import scribble_labels
    N, S, it0, max_it = 4, 16, 20000, 30000       # (N = 4: two samples make the deepest BatchNorm ill-conditioned)
    gen = torch.Generator().manual_seed(13)
    x = torch.rand((N, 1, S, S), generator=gen)
    lab = torch.from_numpy(scribble_labels(N, S, S, 4, share=0.08))
    noises = [torch.clamp(torch.randn((N, 1, S, S), generator=gen) * 0.1, -0.2, 0.2)] + \
             [torch.clamp(torch.randn((2 * N, 1, S, S), generator=gen) * 0.1, -0.2, 0.2) for _ in range(4)]

    def mk_masks(n):
        return [(torch.rand((n, 16 << l, S >> l, S >> l), generator=gen) >= R.DROP[l]).to(torch.uint8) for l in range(5)]
    m_s, m_t = mk_masks(N), {N: mk_masks(N), 2 * N: mk_masks(2 * N)}
    eng = TrainEngine("unet", 1, 4, base_lr=0.01, max_iterations=max_it, loss="ustm")
    load_det(eng.model, 31)
    load_det(eng.teacher, 32)
    SHARP = 60.0                        # a confident teacher head, so that the entropy threshold splits the pixels
    tsd = eng.teacher.state_dict()
    tsd["decoder.out_conv.weight"] = tsd["decoder.out_conv.weight"] * SHARP
    eng.teacher.load_state_dict(tsd)
    eng.it = it0
    eng.model.set_dropout_masks([T(m) for m in m_s])
    eng.teacher.set_dropout_masks(lambda n, h, w: ([T(m) for m in m_t[n]], None))
    pyrandom.seed(77)
    k = pyrandom.Random(77).randrange(0, 4)
    # ---- oracle
    sd_s = {kk: torch.from_numpy(np.asarray(v)).clone() for kk, v in det_state(
        {kk: tuple(v.shape) for kk, v in eng.model.state_dict().items()}, 31).items()}
    sd_t = {kk: torch.from_numpy(np.asarray(v)).clone() for kk, v in det_state(
        {kk: tuple(v.shape) for kk, v in eng.teacher.state_dict().items()}, 32).items()}
    sd_t["decoder.out_conv.weight"] = sd_t["decoder.out_conv.weight"] * SHARP
    pk = [kk for kk in sd_s if R.is_param(kk)]
    for kk in pk:
        sd_s[kk].requires_grad_(True)
    z_s = R.net_forward(sd_s, x, "unet", m_s, None, True)
    xr = torch.rot90(x, k, [2, 3])
    with torch.no_grad():
        z_t = R.net_forward(sd_t, xr + noises[0], "unet", m_t[N], None, True)
        preds = [R.net_forward(sd_t, xr.repeat(2, 1, 1, 1) + noises[1 + i], "unet", m_t[2 * N], None, True) for i in range(4)]
    loss, ce, cons, n_mask = R.ustm_loss(z_s, z_t, preds, lab, k, it0, max_it)
    loss.backward()
    assert 0 < float(n_mask) < N * S * S                      # the threshold actually splits the pixels
    # ---- engine
    eng.forward_backward(T(x), T(lab), 0.5, noise=[T(n) for n in noises])
    o = eng.losses()
    assert rel_err([o["loss"], o["ce"], o["cons"]], [loss.item(), ce.item(), cons.item()]) < TOL
    assert o["n_certain"] == float(n_mask)
    flat = eng.model.flat_grads().cpu().numpy()
    off = 0
    for kk in pk:
        g = sd_s[kk].grad.numpy().ravel()
        assert np.max(np.abs(flat[off:off + g.size] - g)) <= grad_tol(kk, g), kk
        off += g.size


@pytest.mark.parametrize("kind", ["pce_tv", "pce_ms", "pce_entropy", "ce_dice"])
def test_regularised_pce_steps_against_oracle(mode, kind):
    """the single-branch pCE + regulariser scripts (TV / Mumford-Shah / entropy minimisation): losses and gradients of one
    engine step vs the oracle's pinned pieces composed like the scripts"""
    import math
    from oracle import torch_ref as R
    from wsl4mis_amd.engine import TrainEngine
    from wsl4mis_amd.synthetic import scribble_labels
    N, S = 4, 16
    gen = torch.Generator().manual_seed(5)
    x = torch.rand((N, 1, S, S), generator=gen)
    lab = torch.from_numpy(scribble_labels(N, S, S, 4, share=0.08))
    if kind == "ce_dice":                                       # dense labels (fully supervised / random-walker pseudo labels)
        lab = torch.randint(0, 4, (N, S, S), generator=torch.Generator().manual_seed(6)).to(torch.uint8)
    masks = [(torch.rand((N, 16 << l, S >> l, S >> l), generator=gen) >= R.DROP[l]).to(torch.uint8) for l in range(5)]
    eng = TrainEngine("unet", 1, 4, loss=kind)
    load_det(eng.model, 41)
    eng.model.set_dropout_masks([T(m) for m in masks])
    sd = {k: torch.from_numpy(np.asarray(v)).clone() for k, v in det_state(
        {k: tuple(v.shape) for k, v in eng.model.state_dict().items()}, 41).items()}
    pk = [k for k in sd if R.is_param(k)]
    for k in pk:
        sd[k].requires_grad_(True)
    z = R.net_forward(sd, x, "unet", masks, None, True)
    s = torch.softmax(z, 1)
    ce = R.ce_ignore(z, lab)
    if kind == "pce_tv":
        reg, w = R.tv_loss(s[1:]), 1e-2
    elif kind == "pce_ms":
        reg, w = R.mumford_shah(x, s), 1e-6
    elif kind == "ce_dice":
        reg, w = R.dice(s, lab.long().unsqueeze(1)), 0.5        # train_fully_supervised_2D.py:100-102
        ce_w = 0.5
    else:
        reg, w = torch.mean(-1 * torch.sum(s * torch.log(s + 1e-6), dim=1) / math.log(4)), 0.1      # losses.py:30-36
    ce_w = 0.5 if kind == "ce_dice" else 1.0
    (ce_w * ce + w * reg).backward()
    eng.forward_backward(T(x), T(lab), 0.5)
    o = eng.losses()
    assert rel_err([o["loss"], o["ce"], o["dice" if kind == "ce_dice" else "reg"]],
                   [(ce_w * ce + w * reg).item(), ce.item(), reg.item()]) < TOL
    flat = eng.model.flat_grads().cpu().numpy()
    off = 0
    for k in pk:
        g = sd[k].grad.numpy().ravel()
        assert np.max(np.abs(flat[off:off + g.size] - g)) <= grad_tol(k, g), k
        off += g.size
