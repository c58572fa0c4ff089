"""Drop-in for utils/gate_crf_loss.py: ModelLossSemsegGatedCRF with the reference's forward signature
(ref: utils/gate_crf_loss.py:20-124).  One stencil kernel computes the pairwise messages; nothing of the reference's
[N,C,(2r+1)^2,H,W] unfolded tensors is ever materialised.  Options no trainer of the reference passes raise
NotImplementedError (never silently ignored)."""
import torch

from .. import _lib
from .. import runtime as rt
from .losses import _lws, _scalar


class _CRF(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, img, radius, weight, sxy, srgb):
        y, img = rt.f32c(y, "y_hat_softmax"), rt.f32c(img, "sample")
        N, C, H, W = y.shape
        ws, n = _lws(N, C, H * W)
        loss, msg = _scalar(y.device), torch.empty_like(y)
        rt.call("wsl_gatedcrf_fwd", rt.ptr(y), rt.ptr(img), rt.ptr(msg), rt.ptr(loss), N, C, H, W, int(radius), float(sxy),
                float(srgb), float(weight), rt.ptr(ws), n, rt.stream())
        ctx.save_for_backward(msg)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (msg,) = ctx.saved_tensors
        N, C, H, W = msg.shape
        dy = torch.empty_like(msg)
        go = g.reshape(1).to(torch.float32).contiguous()
        rt.call("wsl_gatedcrf_bwd", rt.ptr(msg), rt.ptr(go), 1.0, rt.ptr(dy), N, C, H, W, rt.stream())
        return dy, None, None, None, None, None


class ModelLossSemsegGatedCRF(torch.nn.Module):
    def forward(self, y_hat_softmax, kernels_desc, kernels_radius, sample, height_input, width_input, mask_src=None,
                mask_dst=None, compatibility=None, custom_modality_downsamplers=None, out_kernels_vis=False):
        assert y_hat_softmax.dim() == 4, 'Prediction must be a NCHW batch'
        N, C, height_pred, width_pred = y_hat_softmax.shape
        assert width_input % width_pred == 0 and height_input % height_pred == 0 and \
            width_input * height_pred == height_input * width_pred, \
            f'[{width_input}x{height_input}] !~= [{width_pred}x{height_pred}]'
        for name, val in (("mask_src", mask_src), ("mask_dst", mask_dst), ("compatibility", compatibility),
                          ("custom_modality_downsamplers", custom_modality_downsamplers)):
            if val is not None:
                raise NotImplementedError(f"ModelLossSemsegGatedCRF: `{name}` is not built (Potts model, no masks)")
        if out_kernels_vis:
            raise NotImplementedError("ModelLossSemsegGatedCRF: out_kernels_vis is not built")
        if len(kernels_desc) != 1 or set(kernels_desc[0]) != {"weight", "xy", "rgb"}:
            raise NotImplementedError("ModelLossSemsegGatedCRF: exactly one {'weight','xy','rgb'} kernel descriptor is "
                                      f"built (the one every reference trainer passes), got {kernels_desc}")
        if tuple(sample.shape) != (N, 1, height_pred, width_pred):
            raise NotImplementedError("ModelLossSemsegGatedCRF: `sample` must be a single-channel image at prediction "
                                      f"resolution [N,1,{height_pred},{width_pred}], got {tuple(sample.shape)}")
        d = kernels_desc[0]
        return {"loss": _CRF.apply(y_hat_softmax, sample, kernels_radius, d["weight"], d["xy"], d["rgb"])}
