"""Fused training engine: one optimiser step of the reference's 2-D weakly-supervised trainers as a fixed sequence of
C-ABI calls on one HIP stream, data-parallel over the GPUs of a node with RCCL.

Steps reproduced (reference file:line, all under code/):
  'ours_proposed'   train_weakly_supervised_segmentation_pCE_ours_proposed.py:105-132  (unet_cct)
  'pce'             train_weakly_supervised_pCE_2D.py:96-108 (unet) / dual-branch 0.5*(ce1+ce2) (unet_cct, config 1)
  'pce_gatedcrf'    train_weakly_supervised_pCE_GatedCRFLoss_2D.py:108-130 (unet);  unet_cct: 0.5*(ce1+ce2) +
                    0.1*GatedCRF(beta*s1+(1-beta)*s2) as in train_ACDC_scribblevc.py:171-206
  'pce_tv' | 'pce_ms' | 'pce_entropy'   pCE + 1e-2 tv_loss(softmax[1:]) | + 1e-6 MumfordShah(image, softmax) | + 0.1
                    entropy_loss(softmax, 4)  (..._pCE_TV_2D.py:113-114, ..._pCE_MumfordShah_Loss_2D.py:102-103,
                    ..._pCE_Entropy_Mini_2D.py:99-102), unet only
  'ustm'            train_weakly_supervised_ustm_2D.py:119-163: pCE + w(t) * uncertainty-masked consistency against the EMA
                    teacher (T = 8 stochastic teacher passes on the rot90'ed batch), unet only
  'mean_teacher'    SURVEY 8d config 4 (unet student + EMA teacher): pCE + 1e-2*tv_loss(softmax[1:]) (pCE_TV_2D.py:113-114)
                    + w(t)*mean((softmax(s)-softmax(teacher(x+noise)))^2) (train_mean_teacher_2D.py:147-171); teacher =
                    EMA of the student every step (train_weakly_supervised_ustm_2D.py:61-65,163), kept in train mode
Optimiser: SGD(lr, momentum 0.9, wd 1e-4) with the poly schedule applied one step late (ours_proposed.py:126-132).

Data parallel (DDP-equivalent semantics): one process per GPU, per-rank BatchNorm statistics and loss
normalisation, gradients averaged.  The flat gradient arena is all-reduced in two buckets on a side stream: the
decoders' slice as soon as the decoder backward has been enqueued (it overlaps the encoder backward), then the
encoder's slice.  Losses stay on the device; `losses()` is the only host sync.
"""
import ctypes as C
import math

import torch
import torch.distributed as dist

from . import _lib
from . import runtime as rt
from .networks.net_factory import net_factory


class TrainEngine:
    # pCE + weight * regulariser(softmax(outputs)) of the single-branch scripts: (weight, reference lines)
    REGULARISED = ("pce_tv", "pce_ms", "pce_entropy", "ce_dice")
    FUSED_REG = {"pce_tv": 1, "pce_ms": 2, "pce_entropy": 3}     # WSL_REG_* of wsl_head_reg_fwd_bwd (include/wsl_hip.h)
    REG_WEIGHT = {"pce_tv": 1e-2,        # train_weakly_supervised_pCE_TV_2D.py:113-114 (tv_loss on outputs_soft[1:])
                  "pce_ms": 1e-6,        # ..._pCE_MumfordShah_Loss_2D.py:102-103 (MumfordShah_Loss(image, softmax))
                  "pce_entropy": 0.1,    # ..._pCE_Entropy_Mini_2D.py:99-102 (entropy_loss(softmax, C=4))
                  "ce_dice": 0.5}        # train_fully_supervised_2D.py:100-102 and ..._pCE_random_walker_2D.py:99-101:
    #                                      0.5 * (CE(outputs, label) + DiceLoss(softmax, label.unsqueeze(1))) on dense labels

    def __init__(self, net_type="unet_cct", in_chns=1, class_num=4, base_lr=0.01, max_iterations=60000, momentum=0.9,
                 weight_decay=1e-4, loss="ours_proposed", w_pse=0.5, crf_radius=5, crf_weight=0.1,
                 crf_desc=None, ignore_index=4, model=None, force_dp=False, conv_precision="f32"):
        if loss not in ("ours_proposed", "pce", "pce_gatedcrf", "mean_teacher", "ustm") + self.REGULARISED:
            raise NotImplementedError(f"loss composition '{loss}'")
        # conv_precision: "f32" (default, the headline path) | "split_f16x3" (opt-in: networks/unet.py, include/wsl_hip.h)
        self.model = model if model is not None else net_factory(net_type, in_chns, class_num, conv_precision=conv_precision)
        if self.model is None:
            raise _lib.WslError(f"unknown net_type {net_type}")
        self.model.train()
        self.dual = self.model._n_dec == 2
        if loss == "ours_proposed" and not self.dual:
            raise _lib.WslError("'ours_proposed' needs the dual-branch unet_cct")
        if loss in self.REGULARISED and self.dual:
            raise _lib.WslError(f"'{loss}' is a single-branch (unet) composition")
        self.loss_kind, self.w_pse, self.ignore = loss, w_pse, ignore_index
        self.crf_radius, self.crf_weight = crf_radius, crf_weight
        self.crf_desc = crf_desc or {"weight": 1.0, "xy": 6.0, "rgb": 0.1}
        self.base_lr, self.max_it, self.mu, self.wd = base_lr, max_iterations, momentum, weight_decay
        self.lr, self.it = base_lr, 0
        dev = rt.device()
        self.n = self.model.n_param
        self.mom = torch.zeros(self.n + 64, dtype=torch.float32, device=dev)
        self.loss_out = torch.zeros(8, dtype=torch.float32, device=dev)
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        # force_dp: take the data-parallel route (split backward, bucketed all-reduce on the side stream) in a 1-rank group too
        self.dp = self.world > 1 or (force_dp and dist.is_available() and dist.is_initialized())
        self.comm = torch.cuda.Stream() if (self.dp and dev.type == "cuda") else None
        if self.world > 1:   # identical start on every rank
            dist.broadcast(self.model._param_arena, src=0)
            dist.broadcast(self.model._buf_arena, src=0)
        self._bufs = {}
        self._diag = None                                  # comm_diag(True): [(event, event)] around the waits for the comm stream
        self.teacher = None
        self._tstream, self._zt = None, None
        self.fused_heads = True                            # regulariser / mean-teacher loss heads through wsl_head_reg_fwd_bwd (False: the chain of calls)
        self.concurrent = True                             # teacher forward on a side stream; set False to serialise
        if loss in ("mean_teacher", "ustm"):
            if self.dual:
                raise _lib.WslError(f"'{loss}' is defined for the single-decoder unet")
            self.teacher = net_factory(net_type, in_chns, class_num, conv_precision=self.model.conv_precision)
            self.teacher.train()                          # the reference never puts the EMA model in eval()
            with torch.no_grad():
                self.teacher._param_arena.copy_(self.model._param_arena)
                self.teacher._buf_arena.copy_(self.model._buf_arena)
            self.tv_weight, self.cons_max, self.ema_decay = 1e-2, 0.1, 0.99

    # ------------------------------------------------------------------ helpers
    def _tensors(self, N, H, W):
        key = (N, H, W)
        if key not in self._bufs:
            dev, C_ = rt.device(), self.model.class_num
            mk = lambda: torch.empty((N, C_, H, W), dtype=torch.float32, device=dev)  # noqa: E731
            t = {"dz1": mk(), "dz2": mk() if self.dual else None}
            if self.loss_kind == "pce_gatedcrf":
                t["y"], t["msg"] = mk(), mk()
            if self.loss_kind == "mean_teacher":
                t["s"], t["ds"], t["dzx"] = mk(), mk(), mk()
            if self.loss_kind == "ustm":
                t["zr"], t["dzx"], t["pm"] = mk(), mk(), mk()
            if self.loss_kind in self.REGULARISED:
                t["s"], t["ds"], t["dzx"] = mk(), mk(), mk()
            self._bufs = {key: t}
        return self._bufs[key]

    def _allreduce(self, flat):
        if self.comm is None:
            dist.all_reduce(flat)
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.comm.wait_event(ev)
        with torch.cuda.stream(self.comm):
            if self._diag is not None:     # per-bucket all-reduce time: HIP events on the COMM stream (torch.cuda.Event times the stream it is recorded on)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(self.comm)
                dist.all_reduce(flat)
                b.record(self.comm)
                self._diag_buckets.append((int(flat.numel()), a, b))
            else:
                dist.all_reduce(flat)

    # ------------------------------------------------------------------ one optimiser step
    def step(self, x, label_u8, beta=0.5, noise=None):
        """One optimiser step: forward, loss, backward (+ gradient all-reduce), SGD (+EMA teacher), poly-LR update."""
        self.forward_backward(x, label_u8, beta, noise)
        self.optimizer_step()

    def _noisy(self, x, noise, reps=1):
        """x (repeated `reps` times along the batch) + clamp(randn * 0.1, +-0.2) in one library launch (ustm_2D.py:125-127,
        133-135 / train_mean_teacher_2D.py:147-149); `noise`: a tensor to add instead (parity tests replay the reference's)"""
        out = torch.empty((reps * x.shape[0],) + tuple(x.shape[1:]), dtype=torch.float32, device=x.device)
        if noise is not None:
            noise = rt.f32c(noise, "noise")
            if noise.numel() != out.numel():
                raise _lib.WslError(f"noise has {noise.numel()} elements, expected {out.numel()}")
        seed = 0 if noise is not None else int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
        rt.call("wsl_noisy_copy", rt.ptr(x), rt.ptr(noise), rt.ptr(out), x.numel(), reps, 0.1, 0.2, C.c_uint64(seed), rt.stream())
        return out

    def _teacher_forward(self, x, noise):
        with torch.no_grad():
            return self.teacher._run_forward(self._noisy(x, noise))[0]

    def _start_teacher(self, x, noise):
        """The teacher's forward is independent of the student's: on the GPU it runs on a side stream, forked here (before
        the student forward is enqueued) and joined where its logits are consumed."""
        if x.device.type != "cuda" or not self.concurrent:
            self._zt = None
            return
        if self._tstream is None:
            self._tstream = torch.cuda.Stream(device=x.device)
        cur = torch.cuda.current_stream()
        self._tstream.wait_stream(cur)
        with torch.cuda.stream(self._tstream):
            self._zt = self._teacher_forward(x, noise)
        self._zt.record_stream(cur)

    def _teacher_logits(self, x, noise):
        if self._zt is None:
            return self._teacher_forward(x, noise)
        torch.cuda.current_stream().wait_stream(self._tstream)
        zt, self._zt = self._zt, None
        return zt

    def _regularised_losses(self, x, label_u8, z, t):
        """pCE + weight * R(softmax(z)) with R = tv_loss([1:]) | MumfordShah(image, .) | entropy_loss(., C), and the dense-label
        0.5 * (CE + DiceLoss) of the fully-supervised / random-walker scripts."""
        N, H, W = x.shape[0], x.shape[2], x.shape[3]
        HW, C_ = H * W, self.model.class_num
        nl = rt.L().wsl_loss_ws_bytes(N, C_, HW)
        lws = rt.workspace("loss", nl)
        lo, w = self.loss_out, self.REG_WEIGHT[self.loss_kind]
        if self.loss_kind in self.FUSED_REG and self.fused_heads:
            # one call: the head's first pass keeps softmax(z), the regulariser turns it into its weighted gradient, the head's second
            # pass writes dz once (wsl_head_reg_fwd_bwd; the chain below is kept for ce_dice and as the reference of the fused form)
            rt.call("wsl_head_reg_fwd_bwd", rt.ptr(z), rt.ptr(label_u8), self.ignore, 1.0, self.FUSED_REG[self.loss_kind], w, rt.ptr(x),
                    None, 0.0, rt.ptr(lo), rt.ptr(t["dz1"]), rt.ptr(t["s"]), rt.ptr(t["ds"]), N, C_, H, W, rt.ptr(lws), nl, rt.stream())
            return
        w_ce = 0.5 if self.loss_kind == "ce_dice" else 1.0
        rt.call("wsl_head_fwd_bwd", rt.ptr(z), None, rt.ptr(label_u8), self.ignore, 0.0, 0.0, w_ce, rt.ptr(lo), None,
                rt.ptr(t["dz1"]), None, N, C_, HW, rt.ptr(lws), nl, rt.stream())
        rt.call("wsl_softmax_fwd", rt.ptr(z), rt.ptr(t["s"]), N, C_, HW, rt.stream())
        if self.loss_kind == "pce_tv":
            rt.call("wsl_tv_fwd_bwd", rt.ptr(t["s"]), 1, rt.ptr(lo[4:]), rt.ptr(t["ds"]), w, N, C_, H, W, rt.ptr(lws), nl, rt.stream())
        elif self.loss_kind == "pce_ms":
            rt.call("wsl_mumford_shah_fwd_bwd", rt.ptr(x), rt.ptr(t["s"]), rt.ptr(lo[4:]), rt.ptr(t["ds"]), w, N, C_, H, W,
                    rt.ptr(lws), nl, rt.stream())
        elif self.loss_kind == "ce_dice":
            if "dice" not in t:
                t["dice"] = (torch.empty(3 * C_, dtype=torch.float32, device=z.device),
                             torch.full((1,), w, dtype=torch.float32, device=z.device))
            sums, gout = t["dice"]
            rt.call("wsl_pdice_fwd", rt.ptr(t["s"]), rt.ptr(label_u8), 0, -1, rt.ptr(lo[4:]), rt.ptr(sums), N, C_, HW,
                    rt.ptr(lws), nl, rt.stream())
            rt.call("wsl_pdice_bwd", rt.ptr(t["s"]), rt.ptr(label_u8), 0, -1, rt.ptr(sums), rt.ptr(gout), rt.ptr(t["ds"]),
                    N, C_, HW, rt.stream())
        else:
            rt.call("wsl_entropy_fwd_bwd", rt.ptr(t["s"]), rt.ptr(lo[4:]), rt.ptr(t["ds"]), w, N, C_, HW, C_, rt.ptr(lws), nl, rt.stream())
        rt.call("wsl_softmax_bwd", rt.ptr(t["s"]), rt.ptr(t["ds"]), rt.ptr(t["dzx"]), N, C_, HW, rt.stream())
        rt.call("wsl_axpy", rt.ptr(t["dz1"]), rt.ptr(t["dzx"]), 1.0, N * C_ * HW, rt.stream())

    def _ustm_losses(self, x, label_u8, z, t, noise):
        """train_weakly_supervised_ustm_2D.py:119-157: pCE + w(t) * uncertainty-masked consistency.  `noise`: None (drawn
        like the script) or a list of T//2 + 1 tensors (teacher input noise, then the T//2 double-batch noises)."""
        m, N, H, W = self.model, x.shape[0], x.shape[2], x.shape[3]
        HW, C_, T_ = H * W, self.model.class_num, 8
        if H != W:
            raise _lib.WslError("ustm rotates the batch by multiples of 90 degrees: square inputs only")
        nl = rt.L().wsl_loss_ws_bytes(N, C_, HW)
        lws = rt.workspace("loss", nl)
        lo = self.loss_out
        rt.call("wsl_head_fwd_bwd", rt.ptr(z), None, rt.ptr(label_u8), self.ignore, 0.0, 0.0, 1.0, rt.ptr(lo), None,
                rt.ptr(t["dz1"]), None, N, C_, HW, rt.ptr(lws), nl, rt.stream())
        import random as _random
        k = _random.randrange(0, 4)                           # ustm_2D.py:121 (the script's only python-RNG draw per step)
        xr = torch.empty_like(x)
        rt.call("wsl_rot90", rt.ptr(x), rt.ptr(xr), N * x.shape[1], H, W, k, rt.stream())

        with torch.no_grad():
            zt = self.teacher._run_forward(self._noisy(xr, noise[0] if noise is not None else None))[0]
            for i in range(T_ // 2):                          # T stochastic passes, two per double batch (xr.repeat(2, 1, 1, 1))
                z2 = self.teacher._run_forward(self._noisy(xr, noise[1 + i] if noise is not None else None, reps=2))[0]
                for h in range(2):
                    rt.call("wsl_softmax_accum", rt.ptr(z2[h * N:]), rt.ptr(t["pm"]), 1.0 / T_, int(i == 0 and h == 0), N, C_,
                            HW, rt.stream())
        from .utils.ramps import sigmoid_rampup
        self._cons_w = 1.0 * sigmoid_rampup(self.it // 1000, 60)      # ustm_2D.py:56-58,146: get_current_consistency_weight(iter // 1000)
        thr = (0.75 + 0.25 * sigmoid_rampup(self.it, self.max_it)) * math.log(2.0)
        rt.call("wsl_rot90", rt.ptr(z), rt.ptr(t["zr"]), N * C_, H, W, k, rt.stream())
        rt.call("wsl_ustm_consistency_fwd_bwd", rt.ptr(t["zr"]), rt.ptr(zt), rt.ptr(t["pm"]), float(thr), rt.ptr(lo[4:]),
                rt.ptr(t["dzx"]), self._cons_w, N, C_, HW, rt.ptr(lws), nl, rt.stream())
        rt.call("wsl_rot90", rt.ptr(t["dzx"]), rt.ptr(t["zr"]), N * C_, H, W, 4 - k, rt.stream())    # gradient back through rot90
        rt.call("wsl_axpy", rt.ptr(t["dz1"]), rt.ptr(t["zr"]), 1.0, N * C_ * HW, rt.stream())

    def _mean_teacher_losses(self, x, label_u8, z, t, noise):
        """dz of pCE + tv + consistency for the student logits z; teacher logits from x + noise (no gradient)."""
        m, N, H, W = self.model, x.shape[0], x.shape[2], x.shape[3]
        HW, C_ = H * W, self.model.class_num
        zt = self._teacher_logits(x, noise)                # usually already in flight on the side stream
        nl = rt.L().wsl_loss_ws_bytes(N, C_, HW)
        lws = rt.workspace("loss", nl)
        lo = self.loss_out
        from .utils.ramps import sigmoid_rampup
        self._cons_w = self.cons_max * sigmoid_rampup(self.it // 300, 200.0)
        if self.fused_heads:      # pCE + tv_weight * TV(softmax[1:]) + w(t) * softmax-MSE(student, teacher): one call, dz written once
            rt.call("wsl_head_reg_fwd_bwd", rt.ptr(z), rt.ptr(label_u8), self.ignore, 1.0, 1, self.tv_weight, None, rt.ptr(zt),
                    self._cons_w, rt.ptr(lo), rt.ptr(t["dz1"]), rt.ptr(t["s"]), rt.ptr(t["ds"]), N, C_, H, W, rt.ptr(lws), nl, rt.stream())
            return
        rt.call("wsl_head_fwd_bwd", rt.ptr(z), None, rt.ptr(label_u8), self.ignore, 0.0, 0.0, 1.0, rt.ptr(lo), None,
                rt.ptr(t["dz1"]), None, N, C_, HW, rt.ptr(lws), nl, rt.stream())
        rt.call("wsl_softmax_fwd", rt.ptr(z), rt.ptr(t["s"]), N, C_, HW, rt.stream())
        rt.call("wsl_tv_fwd_bwd", rt.ptr(t["s"]), 1, rt.ptr(lo[4:]), rt.ptr(t["ds"]), self.tv_weight, N, C_, H, W, rt.ptr(lws),
                nl, rt.stream())
        rt.call("wsl_softmax_bwd", rt.ptr(t["s"]), rt.ptr(t["ds"]), rt.ptr(t["dzx"]), N, C_, HW, rt.stream())
        rt.call("wsl_axpy", rt.ptr(t["dz1"]), rt.ptr(t["dzx"]), 1.0, N * C_ * HW, rt.stream())
        rt.call("wsl_softmax_mse_fwd_bwd", rt.ptr(z), rt.ptr(zt), rt.ptr(lo[5:]), rt.ptr(t["dzx"]), self._cons_w, N, C_, HW,
                rt.ptr(lws), nl, rt.stream())
        rt.call("wsl_axpy", rt.ptr(t["dz1"]), rt.ptr(t["dzx"]), 1.0, N * C_ * HW, rt.stream())

    def forward_backward(self, x, label_u8, beta, noise=None):
        """Everything up to (and including) the gradient all-reduce; flat_grads() then holds the SUM over ranks."""
        m = self.model
        x = rt.f32c(x, "image batch")
        N, _, H, W = x.shape
        HW = H * W
        t = self._tensors(N, H, W)
        m.train()
        if self.loss_kind == "mean_teacher":
            self._start_teacher(x, noise)
        outs = m._run_forward(x, keep_for_backward=True)
        z1, z2 = outs[0], (outs[1] if self.dual else None)
        L = rt.L()
        nl = L.wsl_loss_ws_bytes(N, m.class_num, HW)
        lws = rt.workspace("loss", nl)
        if self.loss_kind == "mean_teacher":
            self._mean_teacher_losses(x, label_u8, z1, t, noise)
            self._finish_backward(x, t)
            return
        if self.loss_kind == "ustm":
            self._ustm_losses(x, label_u8, z1, t, noise)
            self._finish_backward(x, t)
            return
        if self.loss_kind in self.REGULARISED:
            self._regularised_losses(x, label_u8, z1, t)
            self._finish_backward(x, t)
            return
        if self.loss_kind == "pce_gatedcrf":              # pCE + crf_weight * GatedCRF(y): one fused entry point
            d = self.crf_desc
            rt.call("wsl_head_gatedcrf_fwd_bwd", rt.ptr(z1), rt.ptr(z2), rt.ptr(label_u8), self.ignore, float(beta), rt.ptr(x),
                    self.crf_radius, d["xy"], d["rgb"], d["weight"], self.crf_weight, rt.ptr(self.loss_out), rt.ptr(t["dz1"]),
                    rt.ptr(t["dz2"]), rt.ptr(t["y"]), rt.ptr(t["msg"]), N, m.class_num, H, W, rt.ptr(lws), nl, rt.stream())
            self._finish_backward(x, t)
            return
        w_pse = self.w_pse if self.loss_kind == "ours_proposed" else 0.0
        rt.call("wsl_head_fwd_bwd", rt.ptr(z1), rt.ptr(z2), rt.ptr(label_u8), self.ignore, float(beta), w_pse, 1.0,
                rt.ptr(self.loss_out), None, rt.ptr(t["dz1"]), rt.ptr(t["dz2"]), N, m.class_num, HW, rt.ptr(lws), nl,
                rt.stream())
        self._finish_backward(x, t)

    def _finish_backward(self, x, t):
        m = self.model
        g = [t["dz1"], t["dz2"]]
        flat_g = m.flat_grads()
        if self.dp:
            m._run_backward(x, g, phase=1)
            self._allreduce(flat_g[m.n_enc_param:])
            m._run_backward(x, g, phase=2)
            self._allreduce(flat_g[:m.n_enc_param])
            if self.comm is not None:
                cur = torch.cuda.current_stream()
                if self._diag is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(cur)
                    cur.wait_stream(self.comm)
                    e1.record(cur)
                    self._diag.append((e0, e1))
                else:
                    cur.wait_stream(self.comm)
        else:
            m._run_backward(x, g, phase=0)

    def comm_diag(self, on):
        """on=True: start timing how long the main stream sits blocked on the all-reduce stream at the end of every backward;
        on=False: stop, return the total in ms (host sync on the recorded events)."""
        if on:
            self._diag, self._diag_buckets = [], []
            return 0.0
        pairs, self._diag = self._diag or [], None
        if pairs:
            pairs[-1][1].synchronize()
        return float(sum(a.elapsed_time(b) for a, b in pairs))

    def bucket_diag(self):
        """after comm_diag(False): {elements of the bucket: mean microseconds of its all-reduce on the comm stream} of the timed steps
        (decoders' bucket first, then the encoder's) -- what explains a scaling line without a second run"""
        recs, self._diag_buckets = getattr(self, "_diag_buckets", []), []
        if not recs:
            return {}
        recs[-1][2].synchronize()
        out = {}
        for n, a, b in recs:
            out.setdefault(n, []).append(a.elapsed_time(b) * 1e3)
        return {str(n): round(sum(v) / len(v), 2) for n, v in out.items()}

    def optimizer_step(self):
        m = self.model
        ema, alpha = None, 0.0
        if self.teacher is not None:      # update_ema_variables(model, ema_model, 0.99, iter_num), iter before increment
            ema, alpha = self.teacher._param_arena, min(1.0 - 1.0 / (self.it + 1), self.ema_decay)
        rt.call("wsl_sgd_step", rt.ptr(m._param_arena), rt.ptr(m._grad_arena), rt.ptr(self.mom), self.n, float(self.lr),
                self.mu, self.wd, int(self.it == 0), 1.0 / self.world, rt.ptr(ema), alpha, rt.stream())
        self.lr = self.base_lr * (1.0 - self.it / self.max_it) ** 0.9      # takes effect at the NEXT step
        self.it += 1

    def losses(self):
        """{loss, ce, pse|crf, n_valid} of the last step (host sync)."""
        o = self.loss_out.tolist()
        if self.loss_kind == "pce_gatedcrf":
            return {"loss": o[1] + self.crf_weight * o[4], "ce": o[1], "crf": o[4], "n_valid": o[3]}
        if self.loss_kind == "ce_dice":
            return {"loss": 0.5 * (o[1] + o[4]), "ce": o[1], "dice": o[4], "n_valid": o[3]}
        if self.loss_kind in self.REGULARISED:  # reg is the raw (unweighted) regulariser
            return {"loss": o[1] + self.REG_WEIGHT[self.loss_kind] * o[4], "ce": o[1], "reg": o[4], "n_valid": o[3]}
        if self.loss_kind == "ustm":           # cons is the raw (unweighted) masked consistency; n_certain = sum(mask)
            return {"loss": o[1] + self._cons_w * o[4], "ce": o[1], "cons": o[4], "n_certain": o[5], "n_valid": o[3]}
        if self.loss_kind == "mean_teacher":   # tv / cons are the raw (unweighted) terms
            return {"loss": o[1] + self.tv_weight * o[4] + self._cons_w * o[5], "ce": o[1], "tv": o[4], "cons": o[5],
                    "n_valid": o[3]}
        return {"loss": o[0], "ce": o[1], "pse": o[2], "n_valid": o[3]}
