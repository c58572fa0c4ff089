"""Drop-in for the reference's utils/losses.py entry points that sit on the hot path, each backed by a HIP kernel
behind the C ABI (include/wsl_hip.h).  Same names, arguments and return types as the reference; arguments the kernels
do not implement raise NotImplementedError instead of being ignored.

    pDLoss(n_classes, ignore_index)(inputs, target, weight=None)         ref: utils/losses.py:195-232
    DiceLoss(n_classes)(inputs, target, weight=None, softmax=False)      ref: utils/losses.py:156-192
    MumfordShah_Loss()(image, prediction)                                ref: utils/losses.py:275-309
    softmax_mse_loss(input_logits, target_logits, sigmoid=False)         ref: utils/losses.py:65-82
    entropy_loss(p, C=2)                                                 ref: utils/losses.py:30-36
plus the pieces the trainers take from torch / define inline:
    PartialCrossEntropyLoss(ignore_index)(logits, target)                ref: ...pCE_2D.py:81,100 (CrossEntropyLoss)
    tv_loss(prediction)                                                  ref: ...pCE_TV_2D.py:58-65
    softmax(logits), mix_argmax(s1, s2, beta), wsl_head(...)             ref: ...pCE_ours_proposed.py:110-125
"""
import math

import torch
import torch.nn as nn

from .. import _lib
from .. import runtime as rt


def _lws(N, C, HW):
    n = rt.L().wsl_loss_ws_bytes(N, C, HW)
    return rt.workspace("loss", n), n


def _scalar(dev):
    return torch.empty(1, dtype=torch.float32, device=dev)


def _label(t):
    """uint8 or int64 labels, contiguous; returns (tensor, is_i64)."""
    if t.dtype == torch.uint8:
        return t.contiguous(), 0
    if t.dtype == torch.int64:
        return t.contiguous(), 1
    raise _lib.WslError(f"labels must be uint8 or int64, got {t.dtype}")


# --------------------------------------------------------------------------------------------------- softmax / CE
class _Softmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z):
        z = rt.f32c(z, "logits")
        N, C = z.shape[:2]
        s = torch.empty_like(z)
        rt.call("wsl_softmax_fwd", rt.ptr(z), rt.ptr(s), N, C, z[0, 0].numel(), rt.stream())
        ctx.save_for_backward(s)
        return s

    @staticmethod
    def backward(ctx, ds):
        (s,) = ctx.saved_tensors
        ds = rt.f32c(ds, "grad")
        dz = torch.empty_like(s)
        rt.call("wsl_softmax_bwd", rt.ptr(s), rt.ptr(ds), rt.ptr(dz), s.shape[0], s.shape[1], s[0, 0].numel(), rt.stream())
        return dz


def softmax(logits):
    """torch.softmax(logits, dim=1) on the HIP path."""
    return _Softmax.apply(logits)


class _CE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, target, ignore):
        z = rt.f32c(z, "logits")
        lab, i64 = _label(target)
        N, C = z.shape[:2]
        HW = z[0, 0].numel()
        ws, n = _lws(N, C, HW)
        loss, dz = _scalar(z.device), torch.empty_like(z)
        rt.call("wsl_ce_fwd_bwd", rt.ptr(z), rt.ptr(lab), i64, int(ignore), rt.ptr(loss), rt.ptr(dz), 1.0, N, C, HW,
                rt.ptr(ws), n, rt.stream())
        ctx.save_for_backward(dz)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dz,) = ctx.saved_tensors
        return dz * g, None, None


class PartialCrossEntropyLoss(nn.Module):
    """torch.nn.CrossEntropyLoss(ignore_index=...) call semantics: loss(logits [N,C,H,W], target [N,H,W])."""

    def __init__(self, ignore_index=4):
        super().__init__()
        self.ignore_index = ignore_index

    def forward(self, logits, target):
        return _CE.apply(logits, target, self.ignore_index)


# --------------------------------------------------------------------------------------------------- dice family
class _PDice(torch.autograd.Function):
    @staticmethod
    def forward(ctx, s, target, n_classes, ignore):
        s = rt.f32c(s, "inputs")
        lab, i64 = _label(target)
        N, C = s.shape[:2]
        if C != n_classes or lab.numel() != N * s[0, 0].numel():
            raise AssertionError("predict & target shape do not match")
        HW = s[0, 0].numel()
        ws, n = _lws(N, C, HW)
        loss, sums = _scalar(s.device), torch.empty(3 * C, dtype=torch.float32, device=s.device)
        rt.call("wsl_pdice_fwd", rt.ptr(s), rt.ptr(lab), i64, int(ignore), rt.ptr(loss), rt.ptr(sums), N, C, HW, rt.ptr(ws),
                n, rt.stream())
        ctx.save_for_backward(s, lab, sums)
        ctx.meta = (i64, int(ignore))
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        s, lab, sums = ctx.saved_tensors
        i64, ignore = ctx.meta
        ds = torch.empty_like(s)
        go = g.reshape(1).to(torch.float32).contiguous()
        rt.call("wsl_pdice_bwd", rt.ptr(s), rt.ptr(lab), i64, ignore, rt.ptr(sums), rt.ptr(go), rt.ptr(ds), s.shape[0],
                s.shape[1], s[0, 0].numel(), rt.stream())
        return ds, None, None, None


def _unit_weight(weight, n):
    if weight is not None and any(float(w) != 1.0 for w in weight):
        raise NotImplementedError("per-class weights other than 1 are not built (no reference trainer passes them)")


class pDLoss(nn.Module):
    def __init__(self, n_classes, ignore_index):
        super().__init__()
        self.n_classes, self.ignore_index = n_classes, ignore_index

    def forward(self, inputs, target, weight=None):
        _unit_weight(weight, self.n_classes)
        return _PDice.apply(inputs, target, self.n_classes, self.ignore_index)


class DiceLoss(nn.Module):
    def __init__(self, n_classes):
        super().__init__()
        self.n_classes = n_classes

    def forward(self, inputs, target, weight=None, softmax=False):
        _unit_weight(weight, self.n_classes)
        if softmax:
            inputs = _Softmax.apply(inputs)
        return _PDice.apply(inputs, target, self.n_classes, -1)


def mix_argmax(s1, s2, beta):
    """argmax(beta*s1 + (1-beta)*s2, dim=1) -> int64 [N,H,W], bit-exact with torch (ours_proposed.py:117-120)."""
    s1, s2 = rt.f32c(s1.detach(), "s1"), rt.f32c(s2.detach(), "s2")
    N, C = s1.shape[:2]
    out = torch.empty((N,) + tuple(s1.shape[2:]), dtype=torch.int64, device=s1.device)
    rt.call("wsl_mix_argmax", rt.ptr(s1), rt.ptr(s2), float(beta), rt.ptr(out), N, C, s1[0, 0].numel(), rt.stream())
    return out


# --------------------------------------------------------------------------------------------------- fused head
class _Head(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z1, z2, label, ignore, beta, w_pse):
        z1 = rt.f32c(z1, "logits1")
        z2 = rt.f32c(z2, "logits2") if z2 is not None else None
        if label.dtype != torch.uint8:
            label = label.to(torch.uint8)
        label = label.contiguous()
        N, C = z1.shape[:2]
        HW = z1[0, 0].numel()
        ws, n = _lws(N, C, HW)
        out = torch.empty(4, dtype=torch.float32, device=z1.device)
        pseudo = torch.empty((N,) + tuple(z1.shape[2:]), dtype=torch.int64, device=z1.device) if z2 is not None else None
        dz1 = torch.empty_like(z1)
        dz2 = torch.empty_like(z1) if z2 is not None else None
        rt.call("wsl_head_fwd_bwd", rt.ptr(z1), rt.ptr(z2), rt.ptr(label), int(ignore), float(beta), float(w_pse), 1.0,
                rt.ptr(out), rt.ptr(pseudo), rt.ptr(dz1), rt.ptr(dz2), N, C, HW, rt.ptr(ws), n, rt.stream())
        ctx.save_for_backward(dz1, dz2)
        ctx.mark_non_differentiable(out)
        if pseudo is not None:
            ctx.mark_non_differentiable(pseudo)
        return out[0], out, pseudo

    @staticmethod
    def backward(ctx, g, _g_out, _g_pseudo):
        dz1, dz2 = ctx.saved_tensors
        return dz1 * g, (dz2 * g if dz2 is not None else None), None, None, None, None


def wsl_head(logits1, logits2, label_u8, beta, ignore_index=4, w_pse=0.5):
    """Fused loss of `ours_proposed` (one pass over the logits forward, one backward):
       loss = 0.5*(CE(l1)+CE(l2)) + w_pse * 0.5*(pDice(s1,pseudo)+pDice(s2,pseudo)),  pseudo = argmax(beta*s1+(1-beta)*s2).
    Returns (loss, parts[4] = {loss, ce, pse, n_valid} on the device, pseudo int64).  logits2=None: plain pCE."""
    return _Head.apply(logits1, logits2, label_u8, ignore_index, beta, w_pse)


# --------------------------------------------------------------------------------------------------- TV / MS / MSE
class _TV(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p):
        p = rt.f32c(p, "prediction")
        N, C, H, W = p.shape
        ws, n = _lws(N, C, H * W)
        loss, dp = _scalar(p.device), torch.empty_like(p)
        rt.call("wsl_tv_fwd_bwd", rt.ptr(p), 0, rt.ptr(loss), rt.ptr(dp), 1.0, N, C, H, W, rt.ptr(ws), n, rt.stream())
        ctx.save_for_backward(dp)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        return ctx.saved_tensors[0] * g


def tv_loss(predication):
    """Module-level tv_loss of the TV trainer; call it on outputs_soft[1:] exactly as the reference does."""
    return _TV.apply(predication)


class _MS(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, p):
        image, p = rt.f32c(image, "image"), rt.f32c(p, "prediction")
        N, C, H, W = p.shape
        if image.shape != (N, 1, H, W):
            raise NotImplementedError("MumfordShah_Loss is built for a single-channel image [N,1,H,W]")
        ws, n = _lws(N, C, H * W)
        loss, dp = _scalar(p.device), torch.empty_like(p)
        rt.call("wsl_mumford_shah_fwd_bwd", rt.ptr(image), rt.ptr(p), rt.ptr(loss), rt.ptr(dp), 1.0, N, C, H, W, rt.ptr(ws),
                n, rt.stream())
        ctx.save_for_backward(dp)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        return None, ctx.saved_tensors[0] * g


class MumfordShah_Loss(nn.Module):
    def forward(self, image, prediction):
        if image.requires_grad:
            raise NotImplementedError("gradient with respect to the image is not built")
        return _MS.apply(image, prediction)


class _MSEMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = rt.f32c(a, "input_logits"), rt.f32c(b.detach(), "target_logits")
        N, C = a.shape[:2]
        HW = a[0, 0].numel()
        ws, n = _lws(N, C, HW)
        loss, da = _scalar(a.device), torch.empty_like(a)
        rt.call("wsl_softmax_mse_fwd_bwd", rt.ptr(a), rt.ptr(b), rt.ptr(loss), rt.ptr(da), 1.0, N, C, HW, rt.ptr(ws), n,
                rt.stream())
        ctx.save_for_backward(da)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        return ctx.saved_tensors[0] * g, None


def softmax_mse_mean(input_logits, target_logits):
    """torch.mean(softmax_mse_loss(a, b)) fused (the form train_mean_teacher_2D.py:164-166 uses)."""
    return _MSEMean.apply(input_logits, target_logits)


def softmax_mse_loss(input_logits, target_logits, sigmoid=False):
    """Elementwise (softmax(a)-softmax(b))**2 map like the reference; gradients flow to `input_logits` only."""
    assert input_logits.size() == target_logits.size()
    if sigmoid:
        raise NotImplementedError("sigmoid=True is not built (no reference trainer on the 2-D path passes it)")
    return (_Softmax.apply(input_logits) - _Softmax.apply(target_logits).detach()) ** 2


class _Entropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, C):
        p = rt.f32c(p, "p")
        N, Cn, HW = p.shape[0], p.shape[1], p[0, 0].numel()
        if int(C) < 2:
            raise ValueError(f"entropy_loss: C={C} (the log(C) normaliser needs C >= 2)")
        ws, n = _lws(N, Cn, HW)
        loss, dp = _scalar(p.device), torch.empty_like(p)
        # like the reference, C is only the normaliser (default 2 whatever the channel count; losses.py:30-33)
        rt.call("wsl_entropy_fwd_bwd", rt.ptr(p), rt.ptr(loss), rt.ptr(dp), 1.0, N, Cn, HW, int(C), rt.ptr(ws), n, rt.stream())
        ctx.save_for_backward(dp)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dp,) = ctx.saved_tensors
        return dp * g, None


def entropy_loss(p, C=2):
    """mean(-sum_c p log(p+1e-6)) / log(C)  (ref: utils/losses.py:30-36), one fused kernel for value and gradient."""
    return _Entropy.apply(p, C)
