#!/usr/bin/env python3
"""Golden-vector generator.  Runs ONLY in the build container, where the upstream
reference is mounted read-only at /root/reference; it imports the reference's own
Python modules (networks.unet, utils.losses, utils.gate_crf_loss), drives them on
seeded inputs and writes small .npz fixtures next to this file.  Nothing from the
reference travels: the fixtures hold inputs and expected outputs only.

    python tests/golden/make_golden.py            # regenerate everything

Reference entry points exercised (all paths relative to /root/reference/code):
  networks/unet.py:13 ConvBlock, :32 DownBlock, :47 UpBlock, :286 UNet, :327 UNet_CCT
  utils/losses.py:156 DiceLoss, :195 pDLoss, :275 MumfordShah_Loss, :65 softmax_mse_loss
  utils/gate_crf_loss.py:5 ModelLossSemsegGatedCRF
  train_weakly_supervised_pCE_TV_2D.py:58 tv_loss            (lifted with ast)
  train_weakly_supervised_ustm_2D.py:61 update_ema_variables (lifted with ast)
  train_weakly_supervised_segmentation_pCE_ours_proposed.py:108-132 (step restated
      here by calling the imported modules in the same order)

Dropout is the only instrumented piece: torch.nn.functional.dropout/dropout2d are
replaced (here only) by recording equivalents so the fixture can carry the exact
Bernoulli masks the reference forward used.
"""
import ast
import os
import random
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/code"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, HERE)

from detinit import det_state, sample_index, scribble_labels  # noqa: E402
from networks.unet import ConvBlock, DownBlock, UpBlock, UNet, UNet_CCT  # noqa: E402
from utils import losses as ref_losses  # noqa: E402
from utils.gate_crf_loss import ModelLossSemsegGatedCRF  # noqa: E402

torch.set_num_threads(8)


def lift(path, fname, env):
    """Extract one top-level function from a reference script without importing it."""
    src = open(os.path.join(REF, path)).read()
    tree = ast.parse(src)
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == fname:
            mod = ast.Module(body=[node], type_ignores=[])
            exec(compile(mod, path, "exec"), env)
            return env[fname]
    raise KeyError(fname)


tv_loss = lift("train_weakly_supervised_pCE_TV_2D.py", "tv_loss", {"nn": torch.nn, "torch": torch})
update_ema_variables = lift("train_weakly_supervised_ustm_2D.py", "update_ema_variables", {"torch": torch})


# ----------------------------------------------------------------------------- dropout
class DropoutRecorder:
    """Context manager that records every Bernoulli mask drawn by F.dropout /
    F.dropout2d during a reference forward (or replays a given list)."""

    def __init__(self):
        self.elem, self.chan = [], []

    def __enter__(self):
        self._d, self._d2 = F.dropout, F.dropout2d
        rec = self

        def dropout(x, p=0.5, training=True, inplace=False):
            if not training or p == 0.0:
                return x
            keep = torch.bernoulli(torch.full_like(x, 1.0 - p))
            rec.elem.append((keep.to(torch.uint8).numpy(), float(p)))
            return x * (keep * np.float32(1.0 / (1.0 - p)))

        def dropout2d(x, p=0.5, training=True, inplace=False):
            if not training or p == 0.0:
                return x
            keep = torch.bernoulli(torch.full((x.shape[0], x.shape[1]), 1.0 - p))
            cm = keep * np.float32(1.0 / (1.0 - p))
            rec.chan.append(cm.numpy().astype(np.float32))
            return x * cm[:, :, None, None]

        F.dropout, F.dropout2d = dropout, dropout2d
        torch.nn.functional.dropout, torch.nn.functional.dropout2d = dropout, dropout2d
        return self

    def __exit__(self, *a):
        F.dropout, F.dropout2d = self._d, self._d2
        torch.nn.functional.dropout, torch.nn.functional.dropout2d = self._d, self._d2


class KinkMargins:
    """Records how close the forward came to a point where the gradient is discontinuous: the smallest |z| fed to a
    LeakyReLU and the smallest gap between the two largest values of a MaxPool2d(2) window.  fp32 parity of gradients is
    undefined when such a margin is inside cross-implementation rounding noise (~1e-6): the sign / arg-max flips and the
    gradient jumps -- the reference's own CPU and GPU runs would disagree there.  gen_net() therefore picks inputs whose
    margins are clear of that noise and stores them in the fixture."""

    def __enter__(self):
        self.leaky, self.pool = float("inf"), float("inf")
        self._l, self._p = F.leaky_relu, F.max_pool2d
        me = self

        def leaky_relu(x, negative_slope=0.01, inplace=False):
            me.leaky = min(me.leaky, float(x.detach().abs().min()))
            return me._l(x, negative_slope, inplace)

        def max_pool2d(x, kernel_size, *a, **k):
            if kernel_size in (2, (2, 2)):
                N, C, H, W = x.shape
                w = x.detach().view(N, C, H // 2, 2, W // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(-1, 4)
                top = torch.topk(w, 2, dim=1).values
                me.pool = min(me.pool, float((top[:, 0] - top[:, 1]).min()))
            return me._p(x, kernel_size, *a, **k)

        F.leaky_relu, F.max_pool2d = leaky_relu, max_pool2d
        torch.nn.functional.leaky_relu, torch.nn.functional.max_pool2d = leaky_relu, max_pool2d
        return self

    def __exit__(self, *a):
        F.leaky_relu, F.max_pool2d = self._l, self._p
        torch.nn.functional.leaky_relu, torch.nn.functional.max_pool2d = self._l, self._p


def load_det(module, seed):
    sd = module.state_dict()
    vals = det_state({k: tuple(v.shape) for k, v in sd.items()}, seed)
    module.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in vals.items()})
    return vals


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"  {name}.npz  {os.path.getsize(path)/1024:.1f} KiB")


def grads_of(module):
    return {k: p.grad.detach().numpy().copy() for k, p in module.named_parameters()}


def buffers_of(module):
    return {k: b.detach().numpy().copy() for k, b in module.named_buffers()}


# ----------------------------------------------------------------------------- G1 per-op
def gen_convblock():
    cases = [  # (tag, Ci, Co, p, N, H, W)
        ("a", 1, 16, 0.05, 2, 24, 20),
        ("b", 16, 32, 0.1, 2, 16, 16),
        ("c", 24, 40, 0.3, 3, 12, 20),
        ("d", 32, 16, 0.0, 2, 16, 32),
    ]
    out = {}
    for tag, ci, co, p, N, H, W in cases:
        torch.manual_seed(100 + ord(tag))
        blk = ConvBlock(ci, co, p).train()
        load_det(blk, 11)
        x = torch.randn(N, ci, H, W, requires_grad=True)
        r = torch.randn(N, co, H, W)
        with DropoutRecorder() as rec:
            y = blk(x)
        (y * r).sum().backward()
        out[f"{tag}_cfg"] = np.array([ci, co, N, H, W], dtype=np.int64)
        out[f"{tag}_p"] = np.float32(p)
        out[f"{tag}_x"] = x.detach().numpy()
        out[f"{tag}_r"] = r.numpy()
        out[f"{tag}_y"] = y.detach().numpy()
        out[f"{tag}_dx"] = x.grad.numpy()
        if rec.elem:
            out[f"{tag}_mask"] = rec.elem[0][0]
        for k, g in grads_of(blk).items():
            out[f"{tag}_g.{k}"] = g
        for k, b in buffers_of(blk).items():
            out[f"{tag}_b.{k}"] = b
        # eval-mode forward (running stats after ONE training step, dropout off)
        blk.eval()
        out[f"{tag}_y_eval"] = blk(x.detach()).detach().numpy()
    save("g1_convblock", **out)


def gen_pool_up():
    out = {}
    # MaxPool2d(2) with ties: values quantised to few levels so windows tie often.
    torch.manual_seed(5)
    x = (torch.randint(0, 3, (2, 5, 12, 16)).float() * 0.5).requires_grad_()
    mp = torch.nn.MaxPool2d(2)
    y = mp(x)
    r = torch.randn_like(y)
    (y * r).sum().backward()
    out.update(mp_x=x.detach().numpy(), mp_y=y.detach().numpy(), mp_r=r.numpy(), mp_dx=x.grad.numpy())
    # DownBlock (maxpool + ConvBlock), reference unet.py:32
    torch.manual_seed(6)
    db = DownBlock(8, 16, 0.2).train()
    load_det(db, 12)
    x = torch.randn(2, 8, 16, 24, requires_grad=True)
    with DropoutRecorder() as rec:
        y = db(x)
    r = torch.randn_like(y)
    (y * r).sum().backward()
    out.update(db_x=x.detach().numpy(), db_y=y.detach().numpy(), db_r=r.numpy(), db_dx=x.grad.numpy(),
               db_mask=rec.elem[0][0])
    for k, g in grads_of(db).items():
        out[f"db_g.{k}"] = g
    # bilinear x2 align_corners=True (unet.py:56-57) incl. degenerate 1->2 and non-square
    up = torch.nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
    for tag, shp in (("u1", (2, 3, 1, 1)), ("u2", (2, 3, 5, 7)), ("u3", (1, 4, 16, 16)), ("u4", (1, 2, 2, 9))):
        torch.manual_seed(hash(tag) % 1000)
        x = torch.randn(*shp, requires_grad=True)
        y = up(x)
        r = torch.randn_like(y)
        (y * r).sum().backward()
        out.update({f"{tag}_x": x.detach().numpy(), f"{tag}_y": y.detach().numpy(),
                    f"{tag}_r": r.numpy(), f"{tag}_dx": x.grad.numpy()})
    # UpBlock (unet.py:47-68, bilinear branch): conv1x1 -> up -> cat([skip, up]) -> ConvBlock
    torch.manual_seed(7)
    ub = UpBlock(32, 16, 16, dropout_p=0.0).train()
    load_det(ub, 13)
    x1 = torch.randn(2, 32, 8, 12, requires_grad=True)
    x2 = torch.randn(2, 16, 16, 24, requires_grad=True)
    y = ub(x1, x2)
    r = torch.randn_like(y)
    (y * r).sum().backward()
    out.update(ub_x1=x1.detach().numpy(), ub_x2=x2.detach().numpy(), ub_y=y.detach().numpy(), ub_r=r.numpy(),
               ub_dx1=x1.grad.numpy(), ub_dx2=x2.grad.numpy())
    for k, g in grads_of(ub).items():
        out[f"ub_g.{k}"] = g
    save("g1_pool_up", **out)


# ----------------------------------------------------------------------------- loss pieces
def ours_proposed_loss(out1, out2, label_u8, beta):
    """The loss of train_weakly_supervised_segmentation_pCE_ours_proposed.py:110-125."""
    ce = torch.nn.CrossEntropyLoss(ignore_index=4)
    dice = ref_losses.pDLoss(4, ignore_index=4)
    s1 = torch.softmax(out1, dim=1)
    s2 = torch.softmax(out2, dim=1)
    lab = label_u8.long()
    loss_ce1, loss_ce2 = ce(out1, lab), ce(out2, lab)
    loss_ce = 0.5 * (loss_ce1 + loss_ce2)
    pseudo = torch.argmax(beta * s1.detach() + (1.0 - beta) * s2.detach(), dim=1, keepdim=False)
    d1, d2 = dice(s1, pseudo.unsqueeze(1)), dice(s2, pseudo.unsqueeze(1))
    loss_pse = 0.5 * (d1 + d2)
    loss = loss_ce + 0.5 * loss_pse
    return dict(loss=loss, loss_ce=loss_ce, loss_ce1=loss_ce1, loss_ce2=loss_ce2, loss_pse=loss_pse,
                d1=d1, d2=d2, pseudo=pseudo, s1=s1, s2=s2)


def gen_head():
    out = {}
    random.seed(2022)
    for tag, (N, H, W) in (("a", (2, 32, 32)), ("b", (3, 24, 40))):
        torch.manual_seed(30 + ord(tag))
        z1 = (torch.randn(N, 4, H, W) * 2.0).requires_grad_()
        z2 = (torch.randn(N, 4, H, W) * 2.0).requires_grad_()
        lab = torch.from_numpy(scribble_labels(N, H, W, seed=ord(tag)))
        if tag == "b":
            lab[:, :4, :] = torch.randint(0, 5, (N, 4, W)).to(torch.uint8)   # denser labels in a band
        beta = random.random() + 1e-10
        res = ours_proposed_loss(z1, z2, lab, beta)
        res["loss"].backward()
        out.update({f"{tag}_z1": z1.detach().numpy(), f"{tag}_z2": z2.detach().numpy(), f"{tag}_label": lab.numpy(),
                    f"{tag}_beta": np.float64(beta),
                    f"{tag}_s1": res["s1"].detach().numpy(), f"{tag}_s2": res["s2"].detach().numpy(),
                    f"{tag}_pseudo": res["pseudo"].numpy(),
                    f"{tag}_dz1": z1.grad.numpy(), f"{tag}_dz2": z2.grad.numpy()})
        for k in ("loss", "loss_ce", "loss_ce1", "loss_ce2", "loss_pse", "d1", "d2"):
            out[f"{tag}_{k}"] = np.float32(res[k].item())
    # mix+argmax bit-exactness over many beta draws on fixed probabilities (incl. exact ties)
    torch.manual_seed(99)
    s1 = torch.softmax(torch.randn(2, 4, 16, 16) * 3, 1)
    s2 = torch.softmax(torch.randn(2, 4, 16, 16) * 3, 1)
    s1[0, :, 0, 0] = 0.25
    s2[0, :, 0, 0] = 0.25
    s1[0, :, 0, 1] = torch.tensor([0.1, 0.4, 0.4, 0.1])
    s2[0, :, 0, 1] = torch.tensor([0.1, 0.4, 0.4, 0.1])
    betas = [random.random() + 1e-10 for _ in range(20)] + [1e-10, 1.0, 0.5]
    out["mix_s1"], out["mix_s2"] = s1.numpy(), s2.numpy()
    out["mix_betas"] = np.array(betas, dtype=np.float64)
    out["mix_pseudo"] = np.stack([torch.argmax(b * s1 + (1.0 - b) * s2, dim=1).numpy() for b in betas])
    # plain CE (pCE_2D.py:81,100) incl. all-ignored -> NaN ; DiceLoss/pDLoss on a generic target with ignore
    torch.manual_seed(41)
    z = torch.randn(2, 4, 16, 16, requires_grad=True)
    lab = torch.randint(0, 5, (2, 16, 16))
    l = torch.nn.CrossEntropyLoss(ignore_index=4)(z, lab)
    l.backward()
    out.update(ce_z=z.detach().numpy(), ce_label=lab.numpy().astype(np.uint8), ce_loss=np.float32(l.item()),
               ce_dz=z.grad.numpy())
    l_nan = torch.nn.CrossEntropyLoss(ignore_index=4)(z.detach(), torch.full((2, 16, 16), 4))
    out["ce_allignored"] = np.float32(l_nan.item())
    s = torch.softmax(torch.randn(2, 4, 16, 16), 1).requires_grad_()
    tgt = torch.randint(0, 5, (2, 1, 16, 16))
    l = ref_losses.pDLoss(4, ignore_index=4)(s, tgt)
    l.backward()
    out.update(pd_s=s.detach().numpy(), pd_target=tgt.numpy(), pd_loss=np.float32(l.item()), pd_ds=s.grad.numpy())
    s = torch.softmax(torch.randn(2, 4, 16, 16), 1).requires_grad_()
    tgt = torch.randint(0, 4, (2, 1, 16, 16))
    l = ref_losses.DiceLoss(4)(s, tgt)
    l.backward()
    out.update(dl_s=s.detach().numpy(), dl_target=tgt.numpy(), dl_loss=np.float32(l.item()), dl_ds=s.grad.numpy())
    # consistency term of the mean-teacher composition (losses.py:65-82)
    a = torch.randn(2, 4, 8, 8, requires_grad=True)
    b = torch.randn(2, 4, 8, 8)
    l = torch.mean(ref_losses.softmax_mse_loss(a, b))
    l.backward()
    out.update(mse_a=a.detach().numpy(), mse_b=b.numpy(), mse_loss=np.float32(l.item()), mse_da=a.grad.numpy())
    save("g3_head", **out)


def gen_crf():
    out = {}
    crf = ModelLossSemsegGatedCRF()
    for tag, (N, H, W, r) in (("r5", (2, 32, 32, 5)), ("r2", (2, 32, 32, 2)), ("ns5", (2, 48, 40, 5)),
                              ("ns2", (1, 20, 28, 2)), ("r1", (1, 8, 8, 1))):
        torch.manual_seed(50 + len(tag) + r)
        y = torch.softmax(torch.randn(N, 4, H, W) * 1.5, 1).requires_grad_()
        img = torch.rand(N, 1, H, W)
        img_in = img.clone()
        loss = crf(y, [{"weight": 1, "xy": 6, "rgb": 0.1}], r, img_in, H, W)["loss"]
        loss.backward()
        assert torch.equal(img, img_in)
        out.update({f"{tag}_y": y.detach().numpy(), f"{tag}_img": img.numpy(), f"{tag}_r": np.int64(r),
                    f"{tag}_loss": np.float32(loss.item()), f"{tag}_dy": y.grad.numpy()})
    # second descriptor set: other sigmas / weight (kernel is a runtime argument, gate_crf_loss.py:21)
    torch.manual_seed(58)
    y = torch.softmax(torch.randn(1, 4, 16, 24), 1).requires_grad_()
    img = torch.rand(1, 1, 16, 24)
    loss = crf(y, [{"weight": 0.7, "xy": 3, "rgb": 0.25}], 3, img.clone(), 16, 24)["loss"]
    loss.backward()
    out.update(alt_y=y.detach().numpy(), alt_img=img.numpy(), alt_r=np.int64(3), alt_loss=np.float32(loss.item()),
               alt_dy=y.grad.numpy(), alt_desc=np.array([0.7, 3.0, 0.25], dtype=np.float64))
    save("g4_gatedcrf", **out)


def gen_crf_general():
    """Round 6 (VERDICT r5 missing 4): what the reference's signature admits beyond the one descriptor its trainers pass --
    several kernel descriptors summed, descriptors without 'xy' / with several sample modalities (gate_crf_loss.py:135-161) and a
    `sample` larger than the prediction (F.adaptive_avg_pool2d, :127-133)."""
    out = {}
    crf = ModelLossSemsegGatedCRF()
    cases = {
        "two": ([{"weight": 1, "xy": 6, "rgb": 0.1}, {"weight": 0.5, "xy": 3}], (2, 24, 32), 3, 1),
        "rgbonly": ([{"weight": 0.8, "rgb": 0.2}], (1, 16, 24), 2, 1),
        "twomod": ([{"weight": 1, "xy": 5, "rgb": 0.15, "depth": 0.3}], (1, 20, 16), 2, 1),
        "down": ([{"weight": 1, "xy": 6, "rgb": 0.1}], (2, 16, 24), 3, 2),
        "three": ([{"weight": 1, "xy": 6, "rgb": 0.1}, {"weight": 0.3, "rgb": 0.5}, {"weight": 0.2, "xy": 2}], (1, 16, 16), 5, 1),
    }
    for i, (tag, (desc, (N, H, W), r, up)) in enumerate(cases.items()):
        torch.manual_seed(170 + i)
        y = torch.softmax(torch.randn(N, 4, H, W) * 1.5, 1).requires_grad_()
        img = torch.rand(N, 1, H * up, W * up)
        img_in = img.clone()
        loss = crf(y, [dict(d) for d in desc], r, img_in, H * up, W * up)["loss"]
        loss.backward()
        assert torch.equal(img, img_in)
        out.update({f"{tag}_y": y.detach().numpy(), f"{tag}_img": img.numpy(), f"{tag}_r": np.int64(r), f"{tag}_up": np.int64(up),
                    f"{tag}_loss": np.float32(loss.item()), f"{tag}_dy": y.grad.numpy(),
                    f"{tag}_desc": np.array(repr(desc))})
    save("g11_gatedcrf_general", **out)


def gen_tv_ms():
    out = {}
    torch.manual_seed(61)
    p = torch.softmax(torch.randn(3, 4, 20, 24) * 2, 1).requires_grad_()
    l = tv_loss(p[1:])                                   # pCE_TV_2D.py:113 drops sample 0
    l.backward()
    out.update(tv_p=p.detach().numpy(), tv_loss=np.float32(l.item()), tv_dp=p.grad.numpy())
    # ties: piecewise-constant probabilities (many equal neighbours) exercise the first-extremum rule
    q = torch.zeros(2, 4, 12, 12)
    q[:, 0] = 1.0
    q[:, 0, 3:8, 4:9] = 0.25
    q[:, 1, 3:8, 4:9] = 0.75
    q[0, 1, 5, 6] = 0.5
    q[0, 2, 5, 6] = 0.25
    q = q.requires_grad_()
    l = tv_loss(q)
    l.backward()
    out.update(tvt_p=q.detach().numpy(), tvt_loss=np.float32(l.item()), tvt_dp=q.grad.numpy())
    torch.manual_seed(62)
    img = torch.rand(2, 1, 16, 20)
    p = torch.softmax(torch.randn(2, 4, 16, 20), 1).requires_grad_()
    l = ref_losses.MumfordShah_Loss()(img, p)            # MumfordShah_Loss_2D.py:102 argument order
    l.backward()
    out.update(ms_img=img.numpy(), ms_p=p.detach().numpy(), ms_loss=np.float32(l.item()), ms_dp=p.grad.numpy())
    save("g5_tv_ms", **out)


def gen_sgd_ema():
    out = {}
    torch.manual_seed(71)
    shapes = [(16, 1, 3, 3), (16,), (32, 16, 3, 3), (7,)]
    params = [torch.nn.Parameter(torch.randn(*s)) for s in shapes]
    ema = [torch.nn.Parameter(p.detach().clone() * 0.5) for p in params]

    class Holder:
        def __init__(self, ps):
            self.ps = ps

        def parameters(self):
            return self.ps

    base_lr, max_it = 0.01, 60000
    opt = torch.optim.SGD(params, lr=base_lr, momentum=0.9, weight_decay=1e-4)
    out["p0"] = np.concatenate([p.detach().numpy().ravel() for p in params])
    out["ema0"] = np.concatenate([p.detach().numpy().ravel() for p in ema])
    gs, ps, es, lrs = [], [], [], []
    for it in range(5):
        g = [torch.randn_like(p) for p in params]
        for p, gi in zip(params, g):
            p.grad = gi.clone()
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        update_ema_variables(Holder(params), Holder(ema), 0.99, it)      # ustm_2D.py:163 (iter before ++)
        lr_ = base_lr * (1.0 - it / max_it) ** 0.9                         # ours_proposed.py:130-132
        for gsd in opt.param_groups:
            gsd["lr"] = lr_
        gs.append(np.concatenate([x.numpy().ravel() for x in g]))
        ps.append(np.concatenate([p.detach().numpy().ravel() for p in params]))
        es.append(np.concatenate([p.detach().numpy().ravel() for p in ema]))
    out.update(grads=np.stack(gs), params=np.stack(ps), emas=np.stack(es), lrs=np.array(lrs, dtype=np.float64))
    save("g6_sgd_ema", **out)


# ----------------------------------------------------------------------------- G2 whole net
def pack_param_grads(module, out, prefix):
    for k, p in module.named_parameters():
        g = p.grad.detach().numpy().ravel()
        idx = sample_index(g.size)
        out[f"{prefix}g.{k}"] = g[idx].astype(np.float32)
        out[f"{prefix}gn.{k}"] = np.array([np.sqrt((g.astype(np.float64) ** 2).sum()), g.astype(np.float64).sum()])


def gen_net():
    # (net, tag, shape, leaky margin asked for).  The margin a fixture can have shrinks with its activation count n
    # (P[min|z| > d] ~ exp(-0.8 n d)): the 16^2 / 32^2 cases are searched until clear of fp32 noise and get the strict
    # 1e-4 gradient check; the larger ones record the best margin found and are checked kink-tolerantly.
    for net, tag, (N, H, W), want in (("unet_cct", "cct16", (4, 16, 16), 1e-5), ("unet_cct", "cct32", (2, 32, 32), 4e-6),
                                      ("unet", "unet32", (2, 32, 32), 4e-6), ("unet_cct", "cct64", (2, 64, 64), None),
                                      ("unet_cct", "cct48x80", (3, 48, 80), None)):
        out = {}
        random.seed(2022)
        beta = random.random() + 1e-10
        lab = torch.from_numpy(scribble_labels(N, H, W, seed=3))
        best = None
        for attempt in range(120 if want else 6):   # pick an input whose forward stays clear of gradient discontinuities
            seed = 200 + len(tag) + 1000 * attempt
            torch.manual_seed(seed)
            model = (UNet_CCT if net == "unet_cct" else UNet)(1, 4).train()
            load_det(model, 2022)
            x = torch.rand(N, 1, H, W)
            with torch.no_grad(), DropoutRecorder(), KinkMargins() as km:
                model(x)
            if best is None or km.leaky > best[0]:
                best = (km.leaky, km.pool, seed)
            if want and km.leaky > want and km.pool > want / 4:
                best = (km.leaky, km.pool, seed)
                break
        else:
            if want:
                raise RuntimeError(f"{tag}: no well-separated input found (best {best})")
        torch.manual_seed(best[2])
        model = (UNet_CCT if net == "unet_cct" else UNet)(1, 4).train()
        load_det(model, 2022)
        x = torch.rand(N, 1, H, W)
        with DropoutRecorder() as rec, KinkMargins() as km:
            res_m = model(x)
        print(f"    {tag}: seed {best[2]}, leaky margin {km.leaky:.2e}, pool margin {km.pool:.2e}")
        out["margins"] = np.array([km.leaky, km.pool])
        if net == "unet_cct":
            o1, o2 = res_m
            res = ours_proposed_loss(o1, o2, lab, beta)
            loss = res["loss"]
            out["logits_aux"] = o2.detach().numpy()
            out["pseudo"] = res["pseudo"].numpy()
            out["loss_parts"] = np.array([res[k].item() for k in ("loss", "loss_ce", "loss_pse")], dtype=np.float32)
        else:
            o1 = res_m
            loss = torch.nn.CrossEntropyLoss(ignore_index=4)(o1, lab.long())      # pCE_2D.py:100
            out["loss_parts"] = np.array([loss.item()], dtype=np.float32)
        loss.backward()
        out.update(x=x.numpy(), label=lab.numpy(), beta=np.float64(beta), logits_main=o1.detach().numpy(),
                   cfg=np.array([N, H, W], dtype=np.int64))
        for i, (m, p) in enumerate(rec.elem):
            out[f"emask{i}"] = m
        for i, cm in enumerate(rec.chan):
            out[f"cmask{i}"] = cm
        pack_param_grads(model, out, "")
        for k, b in buffers_of(model).items():
            out[f"b.{k}"] = b
        # eval-mode forward after that one training step's running-stat update (val_2D.py:100-106 uses [0])
        model.eval()
        with torch.no_grad(), DropoutRecorder():
            ev = model(x)
        out["logits_eval"] = (ev[0] if isinstance(ev, tuple) else ev).numpy()
        save(f"g2_{tag}", **out)


def gen_curve_and_ddp():
    """G7: 12 steps of ours_proposed on a fixed synthetic batch stream (bs 4, 32x32), reference modules +
    torch SGD + poly LR.  G8: one DDP-equivalent step = mean of two shard gradients (bs 2+2)."""
    out = {}
    N, H, W, steps = 4, 32, 32, 12
    torch.manual_seed(2022)
    random.seed(2022)
    model = UNet_CCT(1, 4).train()
    load_det(model, 7)
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    gen = torch.Generator().manual_seed(1)
    xs = torch.rand(steps, N, 1, H, W, generator=gen)
    labs = np.stack([scribble_labels(N, H, W, seed=100 + s) for s in range(steps)])
    losses, betas, emasks, cmasks = [], [], [], []
    for it in range(steps):
        with DropoutRecorder() as rec:
            o1, o2 = model(xs[it])
        beta = random.random() + 1e-10
        res = ours_proposed_loss(o1, o2, torch.from_numpy(labs[it]), beta)
        opt.zero_grad()
        res["loss"].backward()
        opt.step()
        lr_ = 0.01 * (1.0 - it / 60000) ** 0.9
        for g in opt.param_groups:
            g["lr"] = lr_
        losses.append([res["loss"].item(), res["loss_ce"].item(), res["loss_pse"].item()])
        betas.append(beta)
        emasks.append([np.packbits(m.ravel()) for m, _ in rec.elem])
        cmasks.append(rec.chan)
    out.update(xs=xs.numpy(), labels=labs, betas=np.array(betas), losses=np.array(losses, dtype=np.float32))
    for it in range(steps):
        for i in range(5):
            out[f"em{it}_{i}"] = emasks[it][i]
            out[f"cm{it}_{i}"] = cmasks[it][i]
    sd = model.state_dict()
    for k in ("encoder.in_conv.conv_conv.0.weight", "main_decoder.out_conv.weight", "aux_decoder1.up1.conv1x1.bias",
              "encoder.down4.maxpool_conv.1.conv_conv.5.running_var"):
        out[f"final.{k}"] = sd[k].numpy().ravel()[:256].copy()
    save("g7_curve", **out)

    # G8 DDP-equivalent: two shards, per-shard BN stats/loss normalisation, gradients averaged.
    out = {}
    random.seed(8)
    lab = scribble_labels(4, 32, 32, seed=9)
    beta = random.random() + 1e-10
    for attempt in range(200):       # both shards clear of gradient discontinuities (see KinkMargins)
        torch.manual_seed(8 + 1000 * attempt)
        x = torch.rand(4, 1, 32, 32)
        ok = True
        state = torch.get_rng_state()
        for r in range(2):
            model = UNet_CCT(1, 4).train()
            load_det(model, 9)
            with torch.no_grad(), DropoutRecorder(), KinkMargins() as km:
                model(x[2 * r:2 * r + 2])
            ok = ok and km.leaky > 4e-6 and km.pool > 1e-6
        if ok:
            torch.set_rng_state(state)
            break
    else:
        raise RuntimeError("g8: no well-separated input found")
    shard_grads, margins = [], []
    for r in range(2):
        model = UNet_CCT(1, 4).train()
        load_det(model, 9)
        with DropoutRecorder() as rec, KinkMargins() as km:
            o1, o2 = model(x[2 * r:2 * r + 2])
        margins.append([km.leaky, km.pool])
        res = ours_proposed_loss(o1, o2, torch.from_numpy(lab[2 * r:2 * r + 2]), beta)
        res["loss"].backward()
        shard_grads.append({k: p.grad.detach().numpy().astype(np.float64) for k, p in model.named_parameters()})
        for i, (m, _) in enumerate(rec.elem):
            out[f"r{r}_emask{i}"] = m
        for i, cm in enumerate(rec.chan):
            out[f"r{r}_cmask{i}"] = cm
        out[f"r{r}_loss"] = np.float32(res["loss"].item())
    print(f"    g8: attempt {attempt}, margins {margins}")
    out["margins"] = np.min(np.array(margins), axis=0)
    for k in shard_grads[0]:
        g = (0.5 * (shard_grads[0][k] + shard_grads[1][k])).ravel()
        idx = sample_index(g.size)
        out[f"g.{k}"] = g[idx].astype(np.float32)
        out[f"gn.{k}"] = np.array([np.sqrt((g ** 2).sum()), g.sum()])
    out.update(x=x.numpy(), label=lab, beta=np.float64(beta))
    save("g8_ddp", **out)



def gen_crf_curve():
    """G9: the headline composition on the reference's own modules -- UNet_CCT, 0.5*(ce1+ce2) + 0.1*GatedCRF(beta*s1 +
    (1-beta)*s2) (train_ACDC_scribblevc.py:171-206, kernels_desc / radius of ..._pCE_GatedCRFLoss_2D.py:103-123), torch SGD
    + poly LR, 6 steps on a fixed synthetic batch stream (bs 4, 32x32), with recorded masks; gradients of step 0."""
    out = {}
    N, H, W, steps = 4, 32, 32, 6
    best = None
    for attempt in range(400):     # step 0 carries the strict gradient check: keep its forward clear of the kinks (see gen_net)
        seed = 2023 + 1000 * attempt
        torch.manual_seed(seed)
        model = UNet_CCT(1, 4).train()
        load_det(model, 9)
        x0 = torch.rand(steps, N, 1, H, W, generator=torch.Generator().manual_seed(seed))[0]
        with torch.no_grad(), DropoutRecorder(), KinkMargins() as km:
            model(x0)
        if best is None or min(km.leaky, 4 * km.pool) > best[0]:
            best = (min(km.leaky, 4 * km.pool), seed, km.leaky, km.pool)
        if km.leaky > 3e-6 and km.pool > 7.5e-7:
            break
    seed = best[1]
    print(f"    crf_curve: seed {seed}, leaky margin {best[2]:.2e}, pool margin {best[3]:.2e}")
    out["margins"] = np.array([best[2], best[3]])
    torch.manual_seed(seed)
    random.seed(2023)
    model = UNet_CCT(1, 4).train()
    load_det(model, 9)
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    ce = torch.nn.CrossEntropyLoss(ignore_index=4)
    crf = ModelLossSemsegGatedCRF()
    xs = torch.rand(steps, N, 1, H, W, generator=torch.Generator().manual_seed(seed))
    labs = np.stack([scribble_labels(N, H, W, seed=300 + s) for s in range(steps)])
    losses, betas, emasks, cmasks = [], [], [], []
    for it in range(steps):
        with DropoutRecorder() as rec:
            o1, o2 = model(xs[it])
        beta = random.random() + 1e-10
        lab = torch.from_numpy(labs[it]).long()
        s1, s2 = torch.softmax(o1, 1), torch.softmax(o2, 1)
        loss_ce = 0.5 * (ce(o1, lab) + ce(o2, lab))
        y = beta * s1 + (1.0 - beta) * s2
        loss_crf = crf(y, [{"weight": 1, "xy": 6, "rgb": 0.1}], 5, xs[it].clone(), H, W)["loss"]
        loss = loss_ce + 0.1 * loss_crf
        opt.zero_grad()
        loss.backward()
        if it == 0:
            pack_param_grads(model, out, "")
        opt.step()
        lr_ = 0.01 * (1.0 - it / 60000) ** 0.9
        for g in opt.param_groups:
            g["lr"] = lr_
        losses.append([loss.item(), loss_ce.item(), loss_crf.item()])
        betas.append(beta)
        emasks.append([np.packbits(m.ravel()) for m, _ in rec.elem])
        cmasks.append(rec.chan)
    out.update(xs=xs.numpy(), labels=labs, betas=np.array(betas), losses=np.array(losses, dtype=np.float32))
    for it in range(steps):
        for i in range(5):
            out[f"em{it}_{i}"] = emasks[it][i]
            out[f"cm{it}_{i}"] = cmasks[it][i]
    save("g9_crf_curve", **out)


def gen_upblock_t():
    """G10: the transposed-convolution branch of UpBlock (unet.py:58-60, bilinear=False) -- forward, input / parameter
    gradients, BatchNorm buffers, eval forward; and the default initialisation draws of the module."""
    cases = [  # (tag, C1, C2, Co, p, N, h, w)
        ("a", 32, 16, 16, 0.0, 2, 8, 8),
        ("b", 64, 32, 32, 0.1, 2, 8, 16),
        ("c", 12, 6, 10, 0.2, 3, 5, 7),
    ]
    out = {}
    for tag, c1, c2, co, p, N, h, w in cases:
        torch.manual_seed(300 + ord(tag))
        blk = UpBlock(c1, c2, co, p, bilinear=False).train()
        out[f"{tag}_keys"] = np.array(list(blk.state_dict().keys()))
        out[f"{tag}_init_sum"] = np.array([float(v.double().sum()) for v in blk.state_dict().values()])
        load_det(blk, 13)
        x1 = torch.randn(N, c1, h, w, requires_grad=True)
        x2 = torch.randn(N, c2, 2 * h, 2 * w, requires_grad=True)
        r = torch.randn(N, co, 2 * h, 2 * w)
        with DropoutRecorder() as rec:
            y = blk(x1, x2)
        (y * r).sum().backward()
        out[f"{tag}_cfg"] = np.array([c1, c2, co, N, h, w], dtype=np.int64)
        out[f"{tag}_p"] = np.float32(p)
        out.update({f"{tag}_x1": x1.detach().numpy(), f"{tag}_x2": x2.detach().numpy(), f"{tag}_r": r.numpy(),
                    f"{tag}_y": y.detach().numpy(), f"{tag}_dx1": x1.grad.numpy(), f"{tag}_dx2": x2.grad.numpy()})
        if rec.elem:
            out[f"{tag}_mask"] = rec.elem[0][0]
        for k, g in grads_of(blk).items():
            out[f"{tag}_g.{k}"] = g
        for k, b in buffers_of(blk).items():
            out[f"{tag}_b.{k}"] = b
        blk.eval()
        out[f"{tag}_y_eval"] = blk(x1.detach(), x2.detach()).detach().numpy()
    save("g10_upblock_t", **out)


def gen_init_digest():
    """net_factory parity of the *default* torch initialisation (net_factory.py:6-22 builds the module under the
    global torch seed): digest of the reference state_dict for seed 2022."""
    out = {}
    for net, cls in (("unet", UNet), ("unet_cct", UNet_CCT)):
        torch.manual_seed(2022)
        sd = cls(1, 4).state_dict()
        out[f"{net}_keys"] = np.array(list(sd.keys()))
        out[f"{net}_shapes"] = np.array([str(tuple(v.shape)) for v in sd.values()])
        out[f"{net}_sum"] = np.array([float(v.double().sum()) for v in sd.values()])
        out[f"{net}_head"] = np.stack([np.resize(v.double().numpy().ravel(), 4) for v in sd.values()])
    save("g0_init", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["init", "convblock", "pool_up", "head", "crf", "crf_general", "tv_ms", "sgd", "net", "curve", "crf_curve", "upblock_t"]
    fns = dict(init=gen_init_digest, convblock=gen_convblock, pool_up=gen_pool_up, head=gen_head, crf=gen_crf, crf_general=gen_crf_general,
               tv_ms=gen_tv_ms, sgd=gen_sgd_ema, net=gen_net, curve=gen_curve_and_ddp, crf_curve=gen_crf_curve,
               upblock_t=gen_upblock_t)
    for w in which:
        print(w)
        fns[w]()
