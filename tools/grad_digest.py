#!/usr/bin/env python3
"""SHA-256 of the parameter gradients of ONE full-size step (unet_cct, 64 x 256 x 256, pCE + GatedCRF, fixed weights / batch / dropout masks)
in both precisions -- to show that a kernel change which must not alter the arithmetic (scheduling, addressing, DMA form) really leaves
every bit where it was: run with two builds and compare the lines.
   python tools/grad_digest.py [--lib tools/exp/libwslhip_<name>.so]"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wsl4mis_amd import _lib  # noqa: E402
if "--lib" in sys.argv:
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from wsl4mis_amd.engine import TrainEngine  # noqa: E402
from wsl4mis_amd.networks.net_factory import net_factory  # noqa: E402
from wsl4mis_amd.synthetic import batch  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
n, S = 64, 256
torch.manual_seed(2022)
sd0 = {k: v.detach().clone() for k, v in net_factory("unet_cct", 1, 4).state_dict().items()}
x, lab = batch(n, S, S, 2022, dev)
gen = torch.Generator().manual_seed(3)
DROP = (0.05, 0.1, 0.2, 0.3, 0.5)
em = [(torch.rand((n, 16 << l, S >> l, S >> l), generator=gen) >= DROP[l]).to(torch.uint8).to(dev) for l in range(5)]
cm = [((torch.rand((n, 16 << l), generator=gen) >= 0.5).float() * 2.0).to(dev) for l in range(5)]
for prec in ("f32", "split_f16x3"):
    m = net_factory("unet_cct", 1, 4, conv_precision=prec)
    m.load_state_dict(sd0)
    m.train()
    eng = TrainEngine("unet_cct", 1, 4, loss="pce_gatedcrf", crf_radius=5, model=m)
    m.set_dropout_masks(em, cm)
    eng.forward_backward(x, lab, 0.37)
    g = m.flat_grads().detach().cpu().contiguous()
    print(prec, hashlib.sha256(g.numpy().tobytes()).hexdigest(), f"|g| = {float(g.double().norm()):.9e}", flush=True)
