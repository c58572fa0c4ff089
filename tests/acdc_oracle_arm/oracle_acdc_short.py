#!/usr/bin/env python3
"""(Checker-side script: it lives under tests/ because it imports the oracle, which only tests/, smoke() and the bench CPU leg may.)
Short-schedule ACDC training of the ORACLE (stock torch CPU ops, oracle/torch_ref.py) -- the reference-side arm of
profiles/r3_acdc_short_schedule.md.  Build container only: it reads the reference's data directory in
place and never travels to the GPU box.  The HIP arm is examples/train_acdc_scribble.py with the same flags
(--stop_iterations); both arms share file selection, batch order, augmentation draws (python `random` / numpy in the
reference's order), the poly schedule of the 60 000-iteration run and the validation protocol (code/val_2D.py:18-50: zoom,
eval forward, argmax, zoom back, Dice per class over the 20 fold-1 validation volumes).

    python tests/acdc_oracle_arm/oracle_acdc_short.py --root_path /root/reference/data/ACDC --loss pce_tv --seed 2022 \
        --stop_iterations 2000 --val_every 200 --curve_json profiles/r3_acdc_short_oracle_pce_tv_seed2022.json

Loss compositions (single-branch unet, as code/train_wss.sh runs them):
    pce     train_weakly_supervised_pCE_2D.py:99-101      CrossEntropyLoss(ignore_index=4)
    pce_tv  train_weakly_supervised_pCE_TV_2D.py:108-114  + 1e-2 * tv_loss(softmax(z)[1:])
"""
import argparse
import json
import math
import os
import random
import sys
import time

import numpy as np
import torch
from scipy.ndimage import zoom

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import data_ref, torch_ref as R  # noqa: E402
from wsl4mis_amd.dataloaders.dataset import BaseDataSets, draw_params  # noqa: E402  (file selection + draw order only: no kernels)


def default_init(net):
    """nn.Conv2d / nn.BatchNorm2d default initialisation in construction order (what net_factory does on the CPU)."""
    sd, fan_in = {}, 1
    for k, shape in R.state_layout(net, 1, 4):
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros((), dtype=torch.int64)
        elif k.endswith("running_var"):
            sd[k] = torch.ones(shape)
        elif k.endswith("running_mean"):
            sd[k] = torch.zeros(shape)
        elif len(shape) == 4:
            w = torch.empty(shape)
            torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))
            sd[k], fan_in = w, shape[1] * shape[2] * shape[3]
        else:
            is_bn = k.split(".")[-2] in ("1", "5")
            if is_bn:
                sd[k] = torch.ones(shape) if k.endswith("weight") else torch.zeros(shape)
            else:
                b = 1 / math.sqrt(fan_in)
                sd[k] = torch.empty(shape).uniform_(-b, b)
    return sd


def predict_volume(sd, image, patch, slices_per_forward=8):
    pred = np.zeros(image.shape, dtype=np.uint8)
    x, y = image.shape[1:]
    for i0 in range(0, image.shape[0], slices_per_forward):
        inp = np.stack([zoom(s, (patch[0] / x, patch[1] / y), order=0) for s in image[i0:i0 + slices_per_forward]])
        with torch.no_grad():
            z = R.net_forward(sd, torch.from_numpy(np.ascontiguousarray(inp, dtype=np.float32))[:, None], "unet", None, None, False)
        lab = torch.argmax(z, 1).numpy().astype(np.uint8)
        for k in range(lab.shape[0]):
            pred[i0 + k] = zoom(lab[k], (x / patch[0], y / patch[1]), order=0)
    return pred


def dice(pred, gt):            # code/val_2D.py:7-15 with medpy's dc
    pred, gt = pred.astype(bool), gt.astype(bool)
    if pred.sum() == 0:
        return 0.0
    return 2.0 * np.count_nonzero(pred & gt) / (np.count_nonzero(pred) + np.count_nonzero(gt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--root_path", default="/root/reference/data/ACDC")
    ap.add_argument("--fold", default="fold1")
    ap.add_argument("--loss", default="pce_tv", choices=["pce", "pce_tv"])
    ap.add_argument("--max_iterations", type=int, default=60000)
    ap.add_argument("--stop_iterations", type=int, default=2000)
    ap.add_argument("--batch_size", type=int, default=12)
    ap.add_argument("--base_lr", type=float, default=0.01)
    ap.add_argument("--seed", type=int, default=2022)
    ap.add_argument("--val_every", type=int, default=200)
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--curve_json", required=True)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    patch = (256, 256)
    random.seed(a.seed), np.random.seed(a.seed), torch.manual_seed(a.seed)
    train = BaseDataSets(base_dir=a.root_path, split="train", fold=a.fold, sup_type="scribble", labeled_type="all", cache=True)
    val = BaseDataSets(base_dir=a.root_path, split="val", fold=a.fold, cache=True)
    sd = default_init("unet")
    pkeys = [k for k in sd if R.is_param(k)]
    for k in pkeys:
        sd[k].requires_grad_(True)
    bufs = [torch.zeros_like(sd[k]) for k in pkeys]
    torch.manual_seed(a.seed)                      # dropout stream (the example trainer re-seeds here too)
    order = np.random.RandomState(a.seed)
    it, lr, best, log, t0 = 0, a.base_lr, 0.0, [], time.time()

    def flush():
        with open(a.curve_json, "w") as fh:
            json.dump({"arm": "oracle (torch CPU)", "args": vars(a), "train_slices": len(train), "val_volumes": len(val),
                       "best_mean_dice": best, "wall_seconds": round(time.time() - t0, 1), "curve": log}, fh)

    while it < a.stop_iterations:
        perm = order.permutation(len(train))
        for b in range(0, len(perm), a.batch_size):
            idx = perm[b:b + a.batch_size]
            if len(idx) < 2:
                continue
            imgs, labs = [], []
            for i in idx:
                s = train[int(i)]
                p = draw_params(np.asarray(s["label"]))
                im, lb = data_ref.apply(np.asarray(s["image"]), np.asarray(s["label"]), p, patch)
                imgs.append(im), labs.append(lb)
            x = torch.from_numpy(np.stack(imgs))
            lab = torch.from_numpy(np.stack(labs))
            n = x.shape[0]
            em = [(torch.rand((n, 16 << l, 256 >> l, 256 >> l)) >= R.DROP[l]).to(torch.uint8) for l in range(5)]
            random.random()                          # the example trainer draws a beta per step (unused by these losses)
            for k in pkeys:
                sd[k].grad = None
            z = R.net_forward(sd, x, "unet", em, None, True)
            ce = R.ce_ignore(z, lab)
            loss = ce
            if a.loss == "pce_tv":
                loss = ce + 1e-2 * R.tv_loss(torch.softmax(z, 1)[1:])
            loss.backward()
            with torch.no_grad():
                ps = [sd[k] for k in pkeys]
                R.sgd_step(ps, [p.grad for p in ps], bufs, lr, first=(it == 0))
            lr = R.poly_lr(a.base_lr, it, a.max_iterations)
            it += 1
            if it % 20 == 0 or it == 1:
                log.append({"iteration": it, "loss": float(loss.detach()), "loss_ce": float(ce.detach()), "lr": lr})
                print(f"iteration {it} : loss {float(loss.detach()):.4f} ce {float(ce.detach()):.4f} ({time.time() - t0:.0f} s)", flush=True)
            if it % a.val_every == 0:
                acc = np.zeros(3)
                for i in range(len(val)):
                    v = val[i]
                    image, label = np.asarray(v["image"], dtype=np.float32), np.asarray(v["label"])
                    pred = predict_volume(sd, image, patch)
                    acc += np.array([dice(pred == c, label == c) for c in (1, 2, 3)])
                dice_c = acc / len(val)
                md = float(dice_c.mean())
                best = max(best, md)
                log.append({"iteration": it, "mean_dice": md, "dice_per_class": [float(d) for d in dice_c]})
                print(f"iteration {it} : mean_dice {md:.4f} (best {best:.4f})", flush=True)
                flush()
            if it >= a.stop_iterations:
                break
    flush()


if __name__ == "__main__":
    main()
