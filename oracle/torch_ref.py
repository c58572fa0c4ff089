"""ORACLE (test infrastructure, NOT product code).

A functional restatement, in stock PyTorch CPU ops, of the reference's hot path:
the UNet / UNet_CCT forward (code/networks/unet.py), the weak-supervision losses
(code/utils/losses.py, code/utils/gate_crf_loss.py, tv_loss in the trainers) and
the optimiser step of the `ours_proposed` trainer.  The reference itself is nothing
but torch calls, so this is what "the reference's CPU PyTorch path" reduces to.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module -- as the checker / the timed CPU baseline, never as the thing shipped.
The product (wsl4mis_amd/) never imports it and has no CPU fallback.

Pinned against the golden fixtures produced from the real reference
(tests/golden/make_golden.py) by tests/test_oracle_golden.py.

Differences of form (not of arithmetic) from the reference: no nn.Module classes --
parameters live in a flat {state_dict key: tensor} mapping; dropout masks are explicit
inputs (the reference draws them from torch's RNG inside the forward); GatedCRF is
written as a tap loop over shifted views instead of two F.unfold materialisations.
"""
import math

import torch
import torch.nn.functional as F

FT = (16, 32, 64, 128, 256)            # unet.py:291 'feature_chns'
DROP = (0.05, 0.1, 0.2, 0.3, 0.5)      # unet.py:292 'dropout'
BN_EPS, BN_MOM, LEAKY = 1e-5, 0.1, 0.01


# --------------------------------------------------------------------------- layout
def conv_block_keys(prefix, ci, co):
    """state_dict entries of one ConvBlock (unet.py:18-26), in registration order."""
    ks = []
    for idx, cin in (("0", ci), ("4", co)):
        bn = str(int(idx) + 1)
        ks += [(f"{prefix}.{idx}.weight", (co, cin, 3, 3)), (f"{prefix}.{idx}.bias", (co,)),
               (f"{prefix}.{bn}.weight", (co,)), (f"{prefix}.{bn}.bias", (co,)),
               (f"{prefix}.{bn}.running_mean", (co,)), (f"{prefix}.{bn}.running_var", (co,)),
               (f"{prefix}.{bn}.num_batches_tracked", ())]
    return ks


def state_layout(net="unet_cct", in_chns=1, class_num=4):
    """Ordered [(key, shape)] identical to the reference module's state_dict()
    (unet.py:71-135, 286-346): 202 entries for unet_cct, 138 for unet."""
    ks = conv_block_keys("encoder.in_conv.conv_conv", in_chns, FT[0])
    for i in range(1, 5):
        ks += conv_block_keys(f"encoder.down{i}.maxpool_conv.1.conv_conv", FT[i - 1], FT[i])
    decs = ("decoder",) if net == "unet" else ("main_decoder", "aux_decoder1")
    for d in decs:
        for i in range(1, 5):
            c1, c2 = FT[5 - i], FT[4 - i]
            ks += [(f"{d}.up{i}.conv1x1.weight", (c2, c1, 1, 1)), (f"{d}.up{i}.conv1x1.bias", (c2,))]
            ks += conv_block_keys(f"{d}.up{i}.conv.conv_conv", 2 * c2, c2)
        ks += [(f"{d}.out_conv.weight", (class_num, FT[0], 3, 3)), (f"{d}.out_conv.bias", (class_num,))]
    return ks


def is_param(key):
    return not key.endswith(("running_mean", "running_var", "num_batches_tracked"))


# --------------------------------------------------------------------------- network
def _bn_act(sd, pre, y, training):
    """BatchNorm2d (train: batch stats, running stats updated in place; eval: running
    stats) followed by LeakyReLU(0.01) -- unet.py:20-21,24-25."""
    w, b = sd[pre + ".weight"], sd[pre + ".bias"]
    rm, rv = sd[pre + ".running_mean"], sd[pre + ".running_var"]
    if training:
        sd[pre + ".num_batches_tracked"] += 1
    z = F.batch_norm(y, rm, rv, w, b, training, BN_MOM, BN_EPS)
    return F.leaky_relu(z, LEAKY)


def conv_block(sd, pre, x, p, emask, training):
    """ConvBlock forward (unet.py:18-29). emask: uint8 keep mask for nn.Dropout(p) or None."""
    y = F.conv2d(x, sd[pre + ".0.weight"], sd[pre + ".0.bias"], padding=1)
    a = _bn_act(sd, pre + ".1", y, training)
    if training and p > 0.0:
        scale = torch.tensor(1.0 / (1.0 - p), dtype=torch.float32).to(a.dtype)     # (fp64 "truth" runs: tests only)
        a = a * (emask.to(a.dtype) * scale)
    y = F.conv2d(a, sd[pre + ".4.weight"], sd[pre + ".4.bias"], padding=1)
    return _bn_act(sd, pre + ".5", y, training)


def encoder(sd, x, emasks, training):
    """Encoder.forward (unet.py:92-98) -> [x0..x4]."""
    feats = [conv_block(sd, "encoder.in_conv.conv_conv", x, DROP[0], emasks[0] if emasks else None, training)]
    for i in range(1, 5):
        pooled = F.max_pool2d(feats[-1], 2)
        feats.append(conv_block(sd, f"encoder.down{i}.maxpool_conv.1.conv_conv", pooled, DROP[i],
                                emasks[i] if emasks else None, training))
    return feats


def decoder(sd, d, feats, training):
    """Decoder.forward (unet.py:123-135) with the live bilinear UpBlock (unet.py:63-68)."""
    x = feats[4]
    for i in range(1, 5):
        u = F.conv2d(x, sd[f"{d}.up{i}.conv1x1.weight"], sd[f"{d}.up{i}.conv1x1.bias"])
        u = F.interpolate(u, scale_factor=2, mode="bilinear", align_corners=True)
        x = conv_block(sd, f"{d}.up{i}.conv.conv_conv", torch.cat([feats[4 - i], u], dim=1), 0.0, None, training)
    return F.conv2d(x, sd[f"{d}.out_conv.weight"], sd[f"{d}.out_conv.bias"], padding=1)


def upblock_t(sd, x1, x2, p, emask, training, pre=""):
    """UpBlock.forward with bilinear=False (unet.py:58-60, 63-68): ConvTranspose2d(k=2, s=2) -> cat([x2, x1]) -> ConvBlock.
    SURVEY 8f rank 4 (opt-in; the reference's Decoder never selects this branch)."""
    u = F.conv_transpose2d(x1, sd[pre + "up.weight"], sd[pre + "up.bias"], stride=2)
    return conv_block(sd, pre + "conv.conv_conv", torch.cat([x2, u], dim=1), p, emask, training)


def net_forward(sd, x, net="unet_cct", emasks=None, cmasks=None, training=True):
    """UNet.forward (unet.py:300-303) / UNet_CCT.forward (unet.py:341-346).
    emasks: 5 uint8 [N,C,H,W] keep masks (training only); cmasks: 5 float [N,C]
    channel multipliers (0 or 2) for the aux branch's always-on dropout2d (unet.py:254-256)."""
    feats = encoder(sd, x, emasks, training)
    if net == "unet":
        return decoder(sd, "decoder", feats, training)
    main = decoder(sd, "main_decoder", feats, training)
    aux_feats = [f * cm[:, :, None, None] for f, cm in zip(feats, cmasks)]
    aux = decoder(sd, "aux_decoder1", aux_feats, training)
    return main, aux


# --------------------------------------------------------------------------- losses
def ce_ignore(logits, label, ignore=4):
    """CrossEntropyLoss(ignore_index=4) (pCE_2D.py:81,100): mean over non-ignored pixels, NaN if none."""
    return F.cross_entropy(logits, label.long(), ignore_index=ignore)


def mix_argmax(s1, s2, beta):
    """ours_proposed.py:119-120: argmax_c(beta*s1 + (1-beta)*s2); two fp32 multiplies by the rounded
    python-double scalars then one fp32 add; first index on ties."""
    return torch.argmax(beta * s1 + (1.0 - beta) * s2, dim=1)


def pdice(s, target, n_classes=4, ignore=4):
    """pDLoss.forward (losses.py:219-232): squared-denominator Dice, averaged over classes.
    QUIRK restated faithfully: `_dice_loss` multiplies score[:, i] of shape [N,H,W] by ignore_mask of shape
    [N,1,H,W] (losses.py:209-213), which BROADCASTS to [N,N,H,W]; each sum is therefore
    sum_{h,w} (sum_b term[b,h,w]) * (sum_a mask[a,h,w]).  With pseudo labels (mask == 1) every sum is simply
    N times the plain one, which only rescales the 1e-5 smoothing.  target [N,1,H,W] integer."""
    msum = (target != ignore).to(s.dtype).sum(dim=0)[0]          # [H,W]
    loss = 0.0
    for i in range(n_classes):
        t = (target[:, 0] == i).to(s.dtype)
        si = s[:, i]
        inter = torch.sum((si * t).sum(0) * msum)
        y_sum = torch.sum((t * t).sum(0) * msum)
        z_sum = torch.sum((si * si).sum(0) * msum)
        loss = loss + (1 - (2 * inter + 1e-5) / (z_sum + y_sum + 1e-5))
    return loss / n_classes


def dice(s, target, n_classes=4):
    """DiceLoss.forward (losses.py:181-192), softmax=False, unit weights (no mask, no broadcast quirk)."""
    loss = 0.0
    for i in range(n_classes):
        t = (target[:, 0] == i).to(s.dtype)
        si = s[:, i]
        loss = loss + (1 - (2 * torch.sum(si * t) + 1e-5) / (torch.sum(si * si) + torch.sum(t * t) + 1e-5))
    return loss / n_classes


def gatedcrf(y, img, radius, sigma_xy=6.0, sigma_rgb=0.1, weight=1.0):
    """ModelLossSemsegGatedCRF.forward (gate_crf_loss.py:20-124) for one descriptor
    {'weight','xy','rgb'}, Potts compatibility, no masks, prediction at input resolution.
    Zero-padded unfold semantics (gate_crf_loss.py:184-188): an out-of-image tap sees
    feature vector 0 and y=0, so it adds to sum(K) but not to the product."""
    N, C, H, W = y.shape
    r = radius
    fx = (torch.arange(W, dtype=y.dtype) / sigma_xy).view(1, 1, 1, W).expand(N, 1, H, W)
    fy = (torch.arange(H, dtype=y.dtype) / sigma_xy).view(1, 1, H, 1).expand(N, 1, H, W)
    fi = img / sigma_rgb
    feats = torch.cat([fx, fy, fi], dim=1)                 # order xy then rgb (gate_crf_loss.py:142-156)
    fp = F.pad(feats, (r, r, r, r))
    yp = F.pad(y, (r, r, r, r))
    ksum = torch.zeros((), dtype=y.dtype)
    msg = torch.zeros_like(y)
    for dy in range(-r, r + 1):
        for dx in range(-r, r + 1):
            if dy == 0 and dx == 0:
                continue                                   # centre tap := 0 (gate_crf_loss.py:171)
            fq = fp[:, :, r + dy:r + dy + H, r + dx:r + dx + W]
            k = weight * torch.exp((-0.5 * (fq - feats) ** 2).sum(dim=1, keepdim=True))
            ksum = ksum + k.sum()
            msg = msg + k * yp[:, :, r + dy:r + dy + H, r + dx:r + dx + W]
    loss = (ksum - (msg * y).sum()) / (N * H * W)
    return loss, msg


def gatedcrf_general(y, sample, kernels_desc, radius):
    """ModelLossSemsegGatedCRF.forward for what its signature admits beyond the trainers' one descriptor (gate_crf_loss.py:20-188; Potts
    model, no masks): `sample` [N,1,Hs,Ws] is brought to the prediction's resolution by F.adaptive_avg_pool2d (:127-133); every kernel
    descriptor contributes weight * exp(-0.5 * |f_q - f_p|^2) with features f = cat(mesh / sigma if 'xy' is listed, sample / sigma for EVERY
    other modality it lists) in the descriptor's key order (:135-161), the kernels are summed, the centre tap is 0 (:171), out-of-image taps
    see the all-zero feature vector and y = 0 (zero-padded unfold, :184-188).  Returns the loss (a sum over descriptors: it is linear in the
    kernel)."""
    N, C, H, W = y.shape
    r = radius
    if tuple(sample.shape[-2:]) != (H, W):
        sample = F.adaptive_avg_pool2d(sample, (H, W))
    yp = F.pad(y, (r, r, r, r))
    ksum = torch.zeros((), dtype=y.dtype)
    prod = torch.zeros((), dtype=y.dtype)
    for desc in kernels_desc:
        feats = []
        for modality, sigma in desc.items():
            if modality == "weight":
                continue
            if modality == "xy":
                feats.append((torch.arange(W, dtype=y.dtype) / sigma).view(1, 1, 1, W).expand(N, 1, H, W))
                feats.append((torch.arange(H, dtype=y.dtype) / sigma).view(1, 1, H, 1).expand(N, 1, H, W))
            else:
                feats.append(sample.to(y.dtype) / sigma)
        f = torch.cat(feats, dim=1)
        fp = F.pad(f, (r, r, r, r))
        for dy in range(-r, r + 1):
            for dx in range(-r, r + 1):
                if dy == 0 and dx == 0:
                    continue
                fq = fp[:, :, r + dy:r + dy + H, r + dx:r + dx + W]
                k = desc["weight"] * torch.exp((-0.5 * (fq - f) ** 2).sum(dim=1, keepdim=True))
                ksum = ksum + k.sum()
                prod = prod + (k * yp[:, :, r + dy:r + dy + H, r + dx:r + dx + W] * y).sum()
    return (ksum - prod) / (N * H * W)


def tv_loss(p):
    """tv_loss (pCE_TV_2D.py:58-65): mean(relu(dilate3(erode3(p)) - erode3(p)))."""
    e = -F.max_pool2d(-p, (3, 3), 1, 1)
    return torch.mean(torch.abs(torch.relu(F.max_pool2d(e, (3, 3), 1, 1) - e)))


def mumford_shah(img, p):
    """MumfordShah_Loss.forward(image, prediction) (losses.py:275-309; call order of
    MumfordShah_Loss_2D.py:102): centroid divides by sum(image); l1 gradient penalty; a SUM."""
    level = 0.0
    for c in range(p.shape[1]):
        pc = p[:, c:c + 1]
        cen = torch.sum(pc * img, (2, 3), keepdim=True) / torch.sum(img, (2, 3), keepdim=True)
        level = level + torch.sum((pc - cen) ** 2 * img)
    dH = torch.abs(p[:, :, 1:, :] - p[:, :, :-1, :])
    dW = torch.abs(p[:, :, :, 1:] - p[:, :, :, :-1])
    return level + dH.sum() + dW.sum()


def softmax_mse(a, b):
    """losses.softmax_mse_loss (losses.py:65-82), sigmoid=False, elementwise map."""
    return (F.softmax(a, 1) - F.softmax(b, 1)) ** 2


def ours_proposed_loss(o1, o2, label_u8, beta):
    """ours_proposed.py:110-125."""
    s1, s2 = torch.softmax(o1, 1), torch.softmax(o2, 1)
    loss_ce = 0.5 * (ce_ignore(o1, label_u8) + ce_ignore(o2, label_u8))
    pseudo = mix_argmax(s1.detach(), s2.detach(), beta)
    loss_pse = 0.5 * (pdice(s1, pseudo.unsqueeze(1)) + pdice(s2, pseudo.unsqueeze(1)))
    return loss_ce + 0.5 * loss_pse, loss_ce, loss_pse, pseudo


# --------------------------------------------------------------------------- optimiser
def sgd_step(params, grads, bufs, lr, momentum=0.9, wd=1e-4, first=False):
    """torch.optim.SGD(momentum=0.9, weight_decay=1e-4) (ours_proposed.py:89-90), in place."""
    for p, g, b in zip(params, grads, bufs):
        g = g + wd * p
        if first:
            b.copy_(g)
        else:
            b.mul_(momentum).add_(g)
        p.sub_(lr * b)


def poly_lr(base_lr, it, max_it):
    """ours_proposed.py:130: lr for the step AFTER iteration `it` (0-based)."""
    return base_lr * (1.0 - it / max_it) ** 0.9


def ema_update(ema, params, alpha, step):
    """update_ema_variables (ustm_2D.py:61-65)."""
    a = min(1 - 1 / (step + 1), alpha)
    for e, p in zip(ema, params):
        e.mul_(a).add_(p, alpha=1 - a)


def sigmoid_rampup(current, rampup_length):
    """utils/ramps.py:19-26."""
    if rampup_length == 0:
        return 1.0
    cur = min(max(float(current), 0.0), float(rampup_length))
    return float(math.exp(-5.0 * (1.0 - cur / rampup_length) ** 2))


def mean_teacher_loss(z_s, z_t, label_u8, it):
    """Config 4 of BASELINE.json as defined in SURVEY 8d (a composition; no single reference script holds it):
    pCE(student) (pCE_2D.py:100) + 1e-2 * tv_loss(softmax(student)[1:]) (pCE_TV_2D.py:113-114) +
    w(it) * mean((softmax(student) - softmax(teacher))**2) (train_mean_teacher_2D.py:147-171,
    w = 0.1 * sigmoid_rampup(it // 300, 200), :73-75)."""
    s = torch.softmax(z_s, 1)
    ce = ce_ignore(z_s, label_u8)
    tv = tv_loss(s[1:])
    w = 0.1 * sigmoid_rampup(it // 300, 200.0)
    cons = torch.mean(softmax_mse(z_s, z_t.detach()))
    return ce + 1e-2 * tv + w * cons, ce, tv, cons


def ustm_loss(z_s, z_t_rot, preds_logits, label_u8, rot_k, it, max_it):
    """train_weakly_supervised_ustm_2D.py:119-157 given the forwards: z_s student logits, z_t_rot teacher logits on the
    rotated noisy batch, preds_logits = the T//2 teacher outputs on the doubled rotated batch (each [2N,C,H,W])."""
    ce = ce_ignore(z_s, label_u8)
    N, C = z_s.shape[0], z_s.shape[1]
    preds = torch.cat(list(preds_logits), 0)                                    # [stride*T, C, w, h], stride = N
    T = preds.shape[0] // N
    preds = torch.softmax(preds, 1).reshape(T, N, C, z_s.shape[2], z_s.shape[3]).mean(0)
    unc = -1.0 * torch.sum(preds * torch.log(preds + 1e-6), dim=1, keepdim=True)
    w = 1.0 * sigmoid_rampup(it // 1000, 60)                                     # ustm_2D.py:56-58,146
    dist = softmax_mse(torch.rot90(z_s, rot_k, [2, 3]), z_t_rot.detach())
    thr = (0.75 + 0.25 * sigmoid_rampup(it, max_it)) * math.log(2.0)
    mask = (unc < thr).float()
    cons = torch.sum(mask * dist) / (2 * torch.sum(mask) + 1e-16)
    return ce + w * cons, ce, cons, torch.sum(mask)


# --------------------------------------------------------------------------- whole step (CPU baseline)
class RefTrainer:
    """One process' training loop of ours_proposed (or pCE+GatedCRF) in stock torch CPU ops, used as the
    timed `cpu_baseline` ("port") and as the large-input checker on the GPU box."""

    def __init__(self, sd, net="unet_cct", base_lr=0.01, max_it=60000):
        self.net, self.base_lr, self.max_it, self.it = net, base_lr, max_it, 0
        self.sd = {k: v.clone() for k, v in sd.items()}
        self.pkeys = [k for k in self.sd if is_param(k)]
        for k in self.pkeys:
            self.sd[k].requires_grad_(True)
        self.bufs = [torch.zeros_like(self.sd[k]) for k in self.pkeys]
        self.lr = base_lr

    def step(self, x, label_u8, beta, emasks, cmasks, crf=None, kind=None):
        """kind: None = ours_proposed (unet_cct) / pCE (unet), + 0.1 GatedCRF when `crf` (a radius) is given;
        'pce' = the dual-branch 0.5 (ce1 + ce2) alone on unet_cct (BASELINE.json config 1)."""
        for k in self.pkeys:
            self.sd[k].grad = None
        out = net_forward(self.sd, x, self.net, emasks, cmasks, True)
        if self.net == "unet_cct" and kind == "pce":
            lce = 0.5 * (ce_ignore(out[0], label_u8) + ce_ignore(out[1], label_u8))
            loss, lpse = lce, torch.zeros(())
        elif self.net == "unet_cct" and crf is None:
            loss, lce, lpse, _ = ours_proposed_loss(out[0], out[1], label_u8, beta)
        elif self.net == "unet_cct":
            s1, s2 = torch.softmax(out[0], 1), torch.softmax(out[1], 1)
            lce = 0.5 * (ce_ignore(out[0], label_u8) + ce_ignore(out[1], label_u8))
            lcrf, _ = gatedcrf(beta * s1 + (1.0 - beta) * s2, x, crf)
            loss, lpse = lce + 0.1 * lcrf, lcrf
        else:
            lce = ce_ignore(out, label_u8)
            loss, lpse = lce, torch.zeros(())
            if crf is not None:
                lpse, _ = gatedcrf(torch.softmax(out, 1), x, crf)
                loss = lce + 0.1 * lpse
        loss.backward()
        with torch.no_grad():
            ps = [self.sd[k] for k in self.pkeys]
            sgd_step(ps, [p.grad for p in ps], self.bufs, self.lr, first=(self.it == 0))
        self.lr = poly_lr(self.base_lr, self.it, self.max_it)
        self.it += 1
        return float(loss.detach()), float(lce.detach()), float(lpse.detach())


class RefMeanTeacher:
    """BASELINE.json config 4 as SURVEY 8d defines it: student unet + EMA teacher (train mode, no_grad, noisy input);
    loss = mean_teacher_loss; SGD on the student; EMA update with the iteration count before its increment."""

    def __init__(self, sd, base_lr=0.01, max_it=60000):
        self.student = RefTrainer(sd, "unet", base_lr, max_it)
        self.teacher = {k: v.clone() for k, v in sd.items()}

    def step(self, x, label_u8, emasks_s, emasks_t, noise):
        st = self.student
        for k in st.pkeys:
            st.sd[k].grad = None
        with torch.no_grad():
            zt = net_forward(self.teacher, x + noise, "unet", emasks_t, None, True)
        zs = net_forward(st.sd, x, "unet", emasks_s, None, True)
        loss, lce, ltv, lcons = mean_teacher_loss(zs, zt, label_u8, st.it)
        loss.backward()
        with torch.no_grad():
            ps = [st.sd[k] for k in st.pkeys]
            sgd_step(ps, [p.grad for p in ps], st.bufs, st.lr, first=(st.it == 0))
            ema_update([self.teacher[k] for k in st.pkeys], ps, 0.99, st.it)
        st.lr = poly_lr(st.base_lr, st.it, st.max_it)
        st.it += 1
        return float(loss.detach()), float(lce.detach()), float(ltv.detach()), float(lcons.detach())
