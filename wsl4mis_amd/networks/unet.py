"""HIP-backed UNet / UNet_CCT with the reference's module interface (ref: networks/unet.py:286-303, 327-346).

Same constructor arguments, same forward return ((N,C,H,W) logits, or a pair for UNet_CCT), same state_dict keys /
shapes / order (202 entries for unet_cct) and parameters() order, so reference checkpoints load both ways and
`optim.SGD(model.parameters())`, `loss.backward()`, `model.train()/eval()` work unchanged.

MI355X-first differences of form: all parameters are views into ONE flat fp32 arena (so the optimiser and the RCCL
all-reduce see a single buffer), buffers likewise, and a forward is ONE autograd node that enqueues the whole HIP
kernel sequence through the C ABI (wsl_net_forward / wsl_net_backward) instead of ~150 ATen ops.
"""
import ctypes as C
import math

import torch
import torch.nn as nn

from .. import _lib
from .. import runtime as rt

_FT = (16, 32, 64, 128, 256)
_DROP = (0.05, 0.1, 0.2, 0.3, 0.5)


class _NetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, x, *params):
        x = rt.f32c(x, "input")     # the kernels address x as dense NCHW: keep the tensor the forward actually read
        outs = mod._run_forward(x, keep_for_backward=True)
        ctx.mod, ctx.x, ctx.token = mod, x, mod._fwd_token
        return outs if len(outs) > 1 else outs[0]

    @staticmethod
    def backward(ctx, *gouts):
        mod = ctx.mod
        if ctx.token != mod._fwd_token:
            raise _lib.WslError("backward() after a newer training forward of the same module: the activations kept "
                                "in the module's workspace were overwritten (use a second model instance)")
        mod._run_backward(ctx.x, gouts)
        # autograd keeps what is returned here as p.grad and later ACCUMULATES into it in place: hand out views of a private
        # copy, never of the arena the next wsl_net_backward overwrites (the fused TrainEngine reads the arena directly)
        flat = mod._grad_arena.clone()
        return (None, None) + tuple(flat[off:off + n].view(shape) for _, off, n, shape in mod._plist)


class _HipUNet(nn.Module):
    _n_dec = 1
    PRECISIONS = {"f32": 0, "split_f16x3": 1}

    def __init__(self, in_chns, class_num, conv_precision="f32"):
        super().__init__()
        self.in_chns, self.class_num = int(in_chns), int(class_num)
        # "split_f16x3" (opt-in): the 3x3 layers with >= 16 channels on both sides run on the f16 matrix cores with f16 hi / lo
        # operands, three MFMA passes and fp32 accumulation (include/wsl_hip.h, "split-precision conv path"); "f32" is the default
        if conv_precision not in self.PRECISIONS:
            raise ValueError(f"conv_precision {conv_precision!r}: one of {sorted(self.PRECISIONS)}")
        self.conv_precision = conv_precision
        dev = rt.device()
        d0 = self._desc(1, 16, 16)
        L = rt.L()
        n_ent = L.wsl_net_num_entries(C.byref(d0))
        if n_ent <= 0:
            raise _lib.WslError(L.wsl_last_error().decode())
        self._entries = []
        for i in range(n_ent):
            e = _lib.WslNetEntry()
            rt.call("wsl_net_entry", C.byref(d0), i, C.byref(e))
            self._entries.append((e.name.decode(), e.kind, tuple(e.shape[k] for k in range(e.ndim)), e.offset))
        self.n_param = L.wsl_net_param_count(C.byref(d0))
        self.n_enc_param = L.wsl_net_encoder_param_count(C.byref(d0))
        n_buf = L.wsl_net_buffer_count(C.byref(d0))
        n_bn = sum(1 for e in self._entries if e[1] == 2)
        # flat arenas (device memory); 64-float padding keeps float4 paths legal for any tail
        self._param_arena = torch.zeros(self.n_param + 64, dtype=torch.float32, device=dev)
        self._grad_arena = torch.zeros(self.n_param + 64, dtype=torch.float32, device=dev)
        self._buf_arena = torch.zeros(n_buf + 64, dtype=torch.float32, device=dev)
        self._nbt = torch.zeros(n_bn, dtype=torch.int64, device=dev)
        self._build_tree()
        self._default_init()
        self._fwd_token = 0
        self._forced_masks = None
        self._last_masks = None
        self._mask_bufs = {}

    # ------------------------------------------------------------------ structure
    def _desc(self, N, H, W):
        return _lib.WslNetDesc(self.in_chns, self.class_num, self._n_dec, N, H, W, self.PRECISIONS[self.conv_precision], 0)

    def _build_tree(self):
        """Container modules named after the reference's attribute path, so state_dict() keys are identical."""
        self._plist = []
        for name, kind, shape, off in self._entries:
            *path, leaf = name.split(".")
            m = self
            for part in path:
                if part not in m._modules:
                    m.add_module(part, nn.Module())
                m = m._modules[part]
            n = int(math.prod(shape)) if shape else 1
            if kind == 0:
                p = nn.Parameter(self._param_arena[off:off + n].view(shape))
                m.register_parameter(leaf, p)
                self._plist.append((p, off, n, shape))
            elif kind == 1:
                m.register_buffer(leaf, self._buf_arena[off:off + n].view(shape))
            else:
                m.register_buffer(leaf, self._nbt[off])

    @torch.no_grad()
    def _default_init(self):
        """nn.Conv2d / nn.BatchNorm2d default initialisation drawn from torch's global CPU generator in the reference's
        construction order, so `torch.manual_seed(s); net_factory(...)` reproduces the reference's initial weights bit
        for bit (ref: net_factory.py:9-10 builds the module on the CPU, then .cuda())."""
        for name, kind, shape, off in self._entries:
            n = int(math.prod(shape)) if shape else 1
            if kind == 0 and len(shape) == 4:
                w = torch.empty(shape)
                nn.init.kaiming_uniform_(w, a=math.sqrt(5))
                self._param_arena[off:off + n].copy_(w.view(-1))
                self._pending_fan_in = shape[1] * shape[2] * shape[3]
            elif kind == 0:
                parts = name.split(".")
                is_bn = parts[-2] in ("1", "5")
                if is_bn:
                    self._param_arena[off:off + n].fill_(1.0 if parts[-1] == "weight" else 0.0)
                else:
                    bound = 1 / math.sqrt(self._pending_fan_in)
                    self._param_arena[off:off + n].copy_(torch.empty(shape).uniform_(-bound, bound))
            elif kind == 1:
                self._buf_arena[off:off + n].fill_(1.0 if name.endswith("running_var") else 0.0)
        self._nbt.zero_()

    def _ensure_arena(self):
        """Re-attach parameters that were re-allocated behind our back (model.to(...), p.data = ...)."""
        base = self._param_arena.data_ptr()
        for p, off, n, shape in self._plist:
            if p.data_ptr() != base + 4 * off:
                if p.device != self._param_arena.device or p.dtype != torch.float32:
                    raise _lib.WslError("model parameters were moved off the GPU / cast away from float32")
                with torch.no_grad():
                    self._param_arena[off:off + n].copy_(p.data.reshape(-1))
                    p.data = self._param_arena[off:off + n].view(shape)

    # ------------------------------------------------------------------ masks (host-side RNG = torch's, on the device)
    def set_dropout_masks(self, emasks, cmasks=None):
        """Inject the Bernoulli masks of the next forward(s) (parity tests replay the reference's); None = draw."""
        self._forced_masks = emasks if callable(emasks) else (None if emasks is None and cmasks is None else (emasks, cmasks))

    def _draw_masks(self, N, H, W, training, slot="infer"):
        """slot: 'train' for a forward whose masks a pending backward will read again, 'infer' for every other forward --
        two buffer sets, so a no_grad / eval forward between forward and backward cannot redraw the saved masks."""
        dev = self._param_arena.device
        if callable(self._forced_masks):                 # (tests) masks that depend on the batch shape of the forward
            em, cm = self._forced_masks(N, H, W)
        elif self._forced_masks is not None:
            em, cm = self._forced_masks
        else:
            em = cm = None
        want_e = training and em is None
        want_c = self._n_dec == 2 and cm is None     # F.dropout2d(x, 0.5) is active in eval mode too (unet.py:254-256,344)
        if want_e or want_c:
            key = (N, H, W)
            bufs = self._mask_bufs.get(slot)
            if bufs is None or bufs[0] != key:              # persistent mask buffers, refilled in place every forward
                bufs = (key,
                        [torch.empty((N, _FT[l], H >> l, W >> l), dtype=torch.uint8, device=dev) for l in range(5)],
                        [torch.empty((N, _FT[l]), dtype=torch.float32, device=dev) for l in range(5)])
                self._mask_bufs[slot] = bufs
            outs, probs, scales, isf = [], [], [], []
            if want_e:
                em = bufs[1]
                outs += em
                probs += [1.0 - p for p in _DROP]
                scales += [1.0] * 5
                isf += [0] * 5
            if want_c:
                cm = bufs[2]
                outs += cm
                probs += [0.5] * 5
                scales += [2.0] * 5
                isf += [1] * 5
            n = len(outs)
            # the seed comes from torch's (CPU) generator: torch.manual_seed() keeps runs reproducible, no device sync
            seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
            rt.call("wsl_draw_masks", n, rt.ptr_array(outs), (C.c_int64 * n)(*[t.numel() for t in outs]),
                    (C.c_float * n)(*probs), (C.c_float * n)(*scales), (C.c_int * n)(*isf), C.c_uint64(seed), rt.stream())
        return em, cm

    # ------------------------------------------------------------------ execution
    def _run_forward(self, x, keep_for_backward=False):
        x = rt.f32c(x, "input")
        if x.dim() != 4 or x.shape[1] != self.in_chns:
            raise _lib.WslError(f"expected input [N,{self.in_chns},H,W], got {tuple(x.shape)}")
        N, _, H, W = x.shape
        self._ensure_arena()
        d = self._desc(N, H, W)
        nws = rt.L().wsl_net_ws_bytes(C.byref(d))
        if nws == 0:
            raise _lib.WslError(rt.L().wsl_last_error().decode())
        training = self.training
        grad_mode = training and keep_for_backward      # (grad mode is off inside autograd.Function.forward)
        ws = rt.workspace(("net", id(self), "train" if grad_mode else "infer"), nws)
        em, cm = self._draw_masks(N, H, W, training, "train" if grad_mode else "infer")
        lm = torch.empty((N, self.class_num, H, W), dtype=torch.float32, device=x.device)
        la = torch.empty_like(lm) if self._n_dec == 2 else None
        rt.call("wsl_net_forward", C.byref(d), rt.ptr(self._param_arena), rt.ptr(self._buf_arena), rt.ptr(self._nbt),
                rt.ptr(x), rt.ptr_array(em) if training else None, rt.ptr_array(cm), int(training), rt.ptr(lm), rt.ptr(la),
                rt.ptr(ws), nws, rt.stream())
        self._last_masks = (em, cm)
        if grad_mode:
            self._fwd_token += 1
            self._saved = (d, ws, nws, em, cm)
        return (lm, la) if la is not None else (lm,)

    def _run_backward(self, x, gouts, phase=0):
        d, ws, nws, em, cm = self._saved
        g = [rt.f32c(t, "grad_output") if t is not None else None for t in gouts]
        if g[0] is None or (self._n_dec == 2 and g[1] is None):
            shape = (d.N, self.class_num, d.H, d.W)
            g = [t if t is not None else torch.zeros(shape, dtype=torch.float32, device=x.device) for t in
                 (g + [None])[:self._n_dec]]
        rt.call("wsl_net_backward", C.byref(d), rt.ptr(self._param_arena), rt.ptr(x), rt.ptr_array(em), rt.ptr_array(cm),
                rt.ptr(g[0]), rt.ptr(g[1]) if self._n_dec == 2 else None, rt.ptr(self._grad_arena), rt.ptr(ws), nws, phase,
                rt.stream())
        ga = self._grad_arena
        return tuple(ga[off:off + n].view(shape) for _, off, n, shape in self._plist)

    def forward(self, x):
        if x.requires_grad:
            raise NotImplementedError("gradient with respect to the input image is not built (no trainer of the "
                                      "reference's hot path asks for it)")
        if self.training and torch.is_grad_enabled():
            return _NetFn.apply(self, x, *[p for p, _, _, _ in self._plist])
        outs = self._run_forward(x)
        return outs if len(outs) > 1 else outs[0]

    # flat views for the fused trainer / data-parallel engine
    def flat_params(self):
        self._ensure_arena()
        return self._param_arena[:self.n_param]

    def flat_grads(self):
        return self._grad_arena[:self.n_param]

    def cuda(self, device=None):   # already resident; keep net_factory(...).cuda()-style call sites working
        return self


class UNet(_HipUNet):
    """ref: networks/unet.py:286-303 (in_chns, class_num) -> logits [N, class_num, H, W]."""
    _n_dec = 1

    def __init__(self, in_chns, class_num, conv_precision="f32"):
        super().__init__(in_chns, class_num, conv_precision)


class UNet_CCT(_HipUNet):
    """ref: networks/unet.py:327-346 (in_chns, class_num) -> (main_seg, aux_seg1); the aux decoder sees
    F.dropout2d(feature, 0.5) of all five encoder features, in train AND eval mode."""
    _n_dec = 2

    def __init__(self, in_chns, class_num, conv_precision="f32"):
        super().__init__(in_chns, class_num, conv_precision)


class _UpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, x1, x2, *params):
        x1, x2 = rt.f32c(x1, "x1"), rt.f32c(x2, "x2")     # the dense tensors the kernels read are the ones saved
        out = mod._run_forward(x1, x2, keep=True)
        ctx.mod, ctx.token = mod, mod._fwd_token
        ctx.save_for_backward(x1, x2)
        ctx.need = (x1.requires_grad, x2.requires_grad)
        return out

    @staticmethod
    def backward(ctx, gout):
        mod = ctx.mod
        if ctx.token != mod._fwd_token:
            raise _lib.WslError("backward() after a newer training forward of the same UpBlock (its workspace was overwritten)")
        x1, x2 = ctx.saved_tensors
        dx1, dx2, grads = mod._run_backward(x1, x2, gout, ctx.need)
        return (None, dx1, dx2) + grads


class UpBlock(nn.Module):
    """ref: networks/unet.py:47-68 -- `UpBlock(in_channels1, in_channels2, out_channels, dropout_p, bilinear=True)`,
    forward(x1, x2).  SURVEY 8f rank 4 (opt-in): the TRANSPOSED-CONVOLUTION branch (`bilinear=False`: ConvTranspose2d(k=2,
    s=2) -> cat([x2, x1]) -> ConvBlock) as a module of its own on the HIP kernels, with the reference's state_dict layout
    (up.weight [C1,C2,2,2], up.bias, conv.conv_conv.{0,1,4,5}.*) and default initialisation draws.  The reference's Decoder
    never selects this branch (it builds every UpBlock with the default bilinear=True -- that path lives inside UNet /
    UNet_CCT), so `bilinear=True` is not offered stand-alone."""

    def __init__(self, in_channels1, in_channels2, out_channels, dropout_p, bilinear=True):
        super().__init__()
        if bilinear:
            raise NotImplementedError("the bilinear UpBlock is built into UNet / UNet_CCT; stand-alone only bilinear=False is")
        self.bilinear = False
        self.c1, self.c2, self.co, self.p = int(in_channels1), int(in_channels2), int(out_channels), float(dropout_p)
        dev = rt.device()
        n = rt.L().wsl_upblock_t_param_count(C.byref(self._desc(1, 1, 1)))
        if n <= 0:
            raise _lib.WslError(rt.L().wsl_last_error().decode())
        self.n_param = int(n)
        self._param_arena = torch.zeros(self.n_param + 64, dtype=torch.float32, device=dev)
        self._grad_arena = torch.zeros(self.n_param + 64, dtype=torch.float32, device=dev)
        self._buf_arena = torch.zeros(4 * self.co + 64, dtype=torch.float32, device=dev)
        self._nbt = torch.zeros(2, dtype=torch.int64, device=dev)
        c1, c2, co = self.c1, self.c2, self.co
        layout = [("up.weight", (c1, c2, 2, 2)), ("up.bias", (c2,)), ("conv.conv_conv.0.weight", (co, 2 * c2, 3, 3)),
                  ("conv.conv_conv.0.bias", (co,)), ("conv.conv_conv.1.weight", (co,)), ("conv.conv_conv.1.bias", (co,)),
                  ("conv.conv_conv.4.weight", (co, co, 3, 3)), ("conv.conv_conv.4.bias", (co,)),
                  ("conv.conv_conv.5.weight", (co,)), ("conv.conv_conv.5.bias", (co,))]
        self._plist, off = [], 0
        for name, shape in layout:
            *path, leaf = name.split(".")
            m = self
            for part in path:
                if part not in m._modules:
                    m.add_module(part, nn.Module())
                m = m._modules[part]
            k = int(math.prod(shape))
            prm = nn.Parameter(self._param_arena[off:off + k].view(shape))
            m.register_parameter(leaf, prm)
            self._plist.append((prm, off, k, shape))
            if leaf == "bias" and path[-1] in ("1", "5"):           # BatchNorm: its buffers follow its parameters
                bi = 0 if path[-1] == "1" else 1
                m.register_buffer("running_mean", self._buf_arena[2 * bi * co:(2 * bi + 1) * co])
                m.register_buffer("running_var", self._buf_arena[(2 * bi + 1) * co:(2 * bi + 2) * co])
                m.register_buffer("num_batches_tracked", self._nbt[bi])
            off += k
        assert off == self.n_param
        self._default_init()
        self._fwd_token, self._forced_mask, self._saved = 0, None, None

    def _desc(self, N, h, w):
        return _lib.WslUpBlockDesc(self.c1, self.c2, self.co, N, h, w, self.p)

    @torch.no_grad()
    def _default_init(self):
        """the reference's construction-order draws: ConvTranspose2d (kaiming_uniform(a=sqrt(5)) with fan_in = C2 * 4 -- torch
        takes dim 1 of the [C1,C2,2,2] weight --, bias U(+-1/sqrt(fan_in))), then the ConvBlock's Conv2d / BatchNorm2d"""
        for prm, off, k, shape in self._plist:
            name = [n for n, q in self.named_parameters() if q is prm][0]
            if len(shape) == 4:
                w = torch.empty(shape)
                nn.init.kaiming_uniform_(w, a=math.sqrt(5))
                prm.copy_(w)
                fan_in = shape[1] * shape[2] * shape[3]
            elif name.split(".")[-2] in ("1", "5"):
                prm.fill_(1.0 if name.endswith("weight") else 0.0)
            else:
                bound = 1 / math.sqrt(fan_in)
                prm.copy_(torch.empty(shape).uniform_(-bound, bound))
        self._buf_arena.zero_()
        self._buf_arena[self.co:2 * self.co] = 1.0
        self._buf_arena[3 * self.co:4 * self.co] = 1.0
        self._nbt.zero_()

    def set_dropout_mask(self, emask):
        """inject the nn.Dropout keep mask [N, Co, 2h, 2w] uint8 of the next forward(s) (parity tests); None = draw"""
        self._forced_mask = emask

    def _ensure_arena(self):
        """Re-attach parameters that were re-allocated behind the module (p.data = ..., .double().float(), ...)."""
        base = self._param_arena.data_ptr()
        for prm, off, k, shape in self._plist:
            if prm.data_ptr() != base + 4 * off:
                if prm.device != self._param_arena.device or prm.dtype != torch.float32:
                    raise _lib.WslError("UpBlock parameters were moved off the GPU / cast away from float32")
                with torch.no_grad():
                    self._param_arena[off:off + k].copy_(prm.data.reshape(-1))
                    prm.data = self._param_arena[off:off + k].view(shape)

    def _run_forward(self, x1, x2, keep=False):
        x1, x2 = rt.f32c(x1, "x1"), rt.f32c(x2, "x2")
        self._ensure_arena()
        N, c1, h, w = x1.shape
        if c1 != self.c1 or tuple(x2.shape) != (N, self.c2, 2 * h, 2 * w):
            raise _lib.WslError(f"UpBlock: x1 {tuple(x1.shape)} / x2 {tuple(x2.shape)} do not fit ({self.c1}, {self.c2})")
        d = self._desc(N, h, w)
        nws = rt.L().wsl_upblock_t_ws_bytes(C.byref(d))
        ws = rt.workspace(("upblock", id(self), "train" if keep else "infer"), nws)
        training = self.training
        em = None
        if training and self.p > 0:
            em = self._forced_mask
            if em is None:
                em = torch.empty((N, self.co, 2 * h, 2 * w), dtype=torch.uint8, device=x1.device)
                seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
                rt.call("wsl_draw_masks", 1, rt.ptr_array([em]), (C.c_int64 * 1)(em.numel()), (C.c_float * 1)(1.0 - self.p),
                        (C.c_float * 1)(1.0), (C.c_int * 1)(0), C.c_uint64(seed), rt.stream())
        out = torch.empty((N, self.co, 2 * h, 2 * w), dtype=torch.float32, device=x1.device)
        rt.call("wsl_upblock_t_forward", C.byref(d), rt.ptr(self._param_arena), rt.ptr(self._buf_arena), rt.ptr(self._nbt),
                rt.ptr(x1), rt.ptr(x2), rt.ptr(em), int(training), rt.ptr(out), rt.ptr(ws), nws, rt.stream())
        if keep:
            self._fwd_token += 1
            self._saved = (d, ws, nws, em)
        return out

    def _run_backward(self, x1, x2, gout, need):
        d, ws, nws, em = self._saved
        g = rt.f32c(gout, "grad_output")
        x1, x2 = rt.f32c(x1, "x1"), rt.f32c(x2, "x2")
        dx1 = torch.empty(x1.shape, dtype=torch.float32, device=x1.device) if need[0] else None     # dense, whatever x's strides
        dx2 = torch.empty(x2.shape, dtype=torch.float32, device=x2.device) if need[1] else None
        rt.call("wsl_upblock_t_backward", C.byref(d), rt.ptr(self._param_arena), rt.ptr(x1), rt.ptr(x2), rt.ptr(em), rt.ptr(g),
                rt.ptr(self._grad_arena), rt.ptr(dx1), rt.ptr(dx2), rt.ptr(ws), nws, rt.stream())
        flat = self._grad_arena.clone()
        return dx1, dx2, tuple(flat[off:off + k].view(shape) for _, off, k, shape in self._plist)

    def forward(self, x1, x2):
        if self.training and torch.is_grad_enabled():
            return _UpFn.apply(self, x1, x2, *[p for p, _, _, _ in self._plist])
        # eval (or no-grad) forward: same convention as UNet / UNet_CCT.forward -- a gradient with respect to an INPUT is refused,
        # otherwise the result comes back without a graph (gradients in eval mode are not built: the backward kernels replay the
        # training-mode BatchNorm); wrap evaluation in torch.no_grad() as the reference's val_2D.py does
        if torch.is_grad_enabled() and (x1.requires_grad or x2.requires_grad):
            raise NotImplementedError("UpBlock: gradients in eval mode are not built; call .train() or wrap the forward in torch.no_grad()")
        if torch.is_grad_enabled() and not self.training and not getattr(self, "_warned_eval_grad", False) and \
                any(p.requires_grad for p, _, _, _ in self._plist):
            import warnings     # (a loss computed under .eval() with grad enabled would otherwise fail late or train nothing)
            warnings.warn("UpBlock.forward in eval mode returns a tensor WITHOUT an autograd graph although its parameters require grad: "
                          "parameter gradients in eval mode are not built -- call .train() for training or wrap evaluation in torch.no_grad()",
                          RuntimeWarning, stacklevel=2)
            self._warned_eval_grad = True
        return self._run_forward(x1, x2)
