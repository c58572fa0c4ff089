"""BASELINE.json's full problem size (unet_cct, 64 slices of 256x256 per GPU): size-independent properties of the path
(the direct comparisons with the oracle at this size -- fp32 and fp64 -- live in tests/test_error_budget.py):
  * per-sample independence of the eval forward (a batch equals its halves, bit for bit),
  * linearity of the backward in the logit gradients,
  * linearity of the GatedCRF message in y, the loss head's closed forms (uniform logits -> ln 4, valid-pixel count),
  * run-to-run bit-reproducibility of whole optimiser steps (no atomics anywhere)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
N, S = 64, 256


@pytest.fixture(scope="module")
def setup():
    from wsl4mis_amd import _lib, runtime
    from wsl4mis_amd.networks.net_factory import net_factory
    from wsl4mis_amd.synthetic import batch
    _lib._reset_for_tests()
    runtime._ws_cache.clear()
    dev = torch.device("cuda", 0)
    torch.manual_seed(2022)
    model = net_factory("unet_cct", 1, 4)
    x, lab = batch(N, S, S, 2022, dev)
    return model, x, lab


def test_eval_forward_is_per_sample(setup):
    model, x, _ = setup
    model.eval()
    zeros = [torch.zeros((N, c), device=x.device) + 2.0 for c in (16, 32, 64, 128, 256)]      # keep every channel (x2)
    with torch.no_grad():
        model.set_dropout_masks(None, zeros)
        full = [t.clone() for t in model(x)]
        halves = []
        for h in (slice(0, N // 2), slice(N // 2, N)):
            model.set_dropout_masks(None, [z[h] for z in zeros])
            halves.append([t.clone() for t in model(x[h])])
    model.set_dropout_masks(None, None)
    for b in range(2):
        assert torch.equal(full[b], torch.cat([halves[0][b], halves[1][b]], 0))
    assert torch.isfinite(full[0]).all() and float(full[0].abs().max()) > 0


def test_backward_is_linear_in_the_logit_gradients(setup):
    model, x, _ = setup
    model.train()
    torch.manual_seed(7)
    model._run_forward(x, keep_for_backward=True)
    g = [[torch.randn((N, 4, S, S), device=x.device) * 1e-3 for _ in range(2)] for _ in range(2)]

    def bw(ga, gb):
        model._run_backward(x, [ga, gb], phase=0)
        return model.flat_grads().clone()

    g1, g2 = bw(*g[0]), bw(*g[1])
    g12 = bw(0.5 * g[0][0] - 2.0 * g[1][0], 0.5 * g[0][1] - 2.0 * g[1][1])
    ref = 0.5 * g1 - 2.0 * g2
    # fp32 sums over 4.2 M pixels per weight: the two sides round differently -- the 1e-4 parity bar is the bound
    assert float((g12 - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    assert torch.equal(bw(*g[0]), g1)                                         # and deterministic


def test_loss_head_closed_forms_and_crf_linearity(setup):
    from wsl4mis_amd.utils import losses
    from wsl4mis_amd.utils.gate_crf_loss import ModelLossSemsegGatedCRF
    _, x, lab = setup
    z = torch.zeros((N, 4, S, S), device=x.device)
    ce = losses.PartialCrossEntropyLoss(ignore_index=4)(z, lab.long())
    assert abs(float(ce) - math.log(4.0)) < 1e-6
    loss, parts, pseudo = losses.wsl_head(z, z, lab, 0.3)
    assert torch.equal(pseudo, torch.zeros_like(pseudo))                      # ties -> lowest class index
    n_valid = int((lab != 4).sum())
    assert n_valid > 0 and abs(float(parts[-1]) - n_valid) < 0.5
    crf = ModelLossSemsegGatedCRF()
    kd = [{"weight": 1, "xy": 6, "rgb": 0.1}]
    ys = [torch.softmax(torch.randn((N, 4, S, S), device=x.device), 1).requires_grad_() for _ in range(2)]
    grads = []
    for y in ys + [(0.25 * ys[0] + 0.75 * ys[1]).detach().requires_grad_()]:
        out = crf(y, kd, 5, x, S, S)["loss"]
        out.backward()
        grads.append(y.grad.clone())                                          # dL/dy = -2 msg / (N H W): linear in y
    ref = 0.25 * grads[0] + 0.75 * grads[1]
    assert float((grads[2] - ref).abs().max()) <= 1e-4 * float(ref.abs().max())


@pytest.mark.parametrize("precision", ["f32", "split_f16x3"])
def test_whole_step_is_bit_reproducible(setup, precision):
    """Whole optimiser steps at the benchmark size, both decoder streams, FIVE repetitions, for both conv precisions: losses and every
    parameter bit-identical.  This is the level at which round 3's split path failed (wrong AND different from run to run: the f32
    kernels beside the other decoder's f16-MFMA kernels computed a wrong packed FMA -- profiles/r4_sp_root_cause.md); the
    disassembly scan of tests/test_abi.py guards the cause, this test the symptom."""
    from wsl4mis_amd.engine import TrainEngine
    _, x, lab = setup

    def run():
        torch.manual_seed(2022)
        eng = TrainEngine("unet_cct", 1, 4, base_lr=0.01, max_iterations=60000, loss="pce_gatedcrf", crf_radius=5,
                          conv_precision=precision)
        assert eng.concurrent
        torch.manual_seed(99)
        for b in (0.3, 0.7):
            eng.step(x, lab, b)
        return eng.losses(), eng.model.flat_params().clone()

    l1, p1 = run()
    assert all(math.isfinite(v) for v in l1.values())
    for rep in range(4):
        l2, p2 = run()
        assert l1 == l2 and torch.equal(p1, p2), (precision, rep, l1, l2, int((p1 != p2).sum()))
