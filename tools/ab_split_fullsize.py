#!/usr/bin/env python3
"""Split-precision path against the f32 path on ONE full-size batch (unet_cct, 64 x 256 x 256, pCE + GatedCRF), GPU only: relative L2
distance of the two parameter gradients, whole and per sub-network.  Two correct fp32-class implementations differ by the LeakyReLU /
max-pool decisions round-off flips (~2e-3 at this size: tests/test_error_budget.py); a wrong kernel shows as 1e-2 and more.
   WSL_LIB=<other build of libwslhip.so> python tools/ab_split_fullsize.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wsl4mis_amd import _lib  # noqa: E402
if os.environ.get("WSL_LIB"):
    _lib.LIB_PATH = os.environ["WSL_LIB"]
from wsl4mis_amd.engine import TrainEngine  # noqa: E402
from wsl4mis_amd.networks.net_factory import net_factory  # noqa: E402
from wsl4mis_amd.synthetic import batch  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
n, S = 64, 256
torch.manual_seed(2022)
model = net_factory("unet_cct", 1, 4)
sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
x, lab = batch(n, S, S, 2022, dev)
gen = torch.Generator().manual_seed(3)
DROP = (0.05, 0.1, 0.2, 0.3, 0.5)
em = [(torch.rand((n, 16 << l, S >> l, S >> l), generator=gen) >= DROP[l]).to(torch.uint8).to(dev) for l in range(5)]
cm = [((torch.rand((n, 16 << l), generator=gen) >= 0.5).float() * 2.0).to(dev) for l in range(5)]
names = [k for k, _ in model.named_parameters()]
sizes = [p.numel() for _, p in model.named_parameters()]


def grads(prec):
    m = net_factory("unet_cct", 1, 4, conv_precision=prec)
    m.load_state_dict(sd0)
    m.train()
    eng = TrainEngine("unet_cct", 1, 4, loss="pce_gatedcrf", crf_radius=5, model=m)
    m.set_dropout_masks(em, cm)
    eng.forward_backward(x, lab, 0.37)
    return m.flat_grads().double().clone()


g32 = grads("f32")
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    gsp = grads("split_f16x3")
    parts, off = {}, 0
    for k, sz in zip(names, sizes):
        grp = k.split(".")[0]
        a, b = gsp[off:off + sz], g32[off:off + sz]
        num, den = parts.get(grp, (0.0, 0.0))
        parts[grp] = (num + float(((a - b) ** 2).sum()), den + float((b ** 2).sum()))
        off += sz
    tot = float((gsp - g32).norm() / g32.norm())
    print(f"[{os.environ.get('WSL_LIB', 'product')}] rep {rep}: split vs f32 whole-gradient L2 {tot:.2e}; " +
          ", ".join(f"{k} {(v[0] / v[1]) ** 0.5:.2e}" for k, v in parts.items()), flush=True)
