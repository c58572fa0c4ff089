#!/usr/bin/env python3
"""Where do two runs of the split-precision step differ?  (profiles/r3_sp_hunt.md cause 2; GPU only.)

Runs the SAME full-size training forward + backward (unet_cct, 64 x 256 x 256, pCE + GatedCRF, fixed weights / masks / batch) R times
with the split-precision conv path of the library named by WSL_LIB and compares every named region of the network workspace
(wsl_debug_net_ws_region: raw conv outputs, decoder tensors, gradients in the scratch sets) and the parameter gradient with run 0,
bit for bit.  For a region that differs it prints how many elements differ, the size of the deviation relative to the region's RMS,
and the spatial structure of the differing elements (which samples / channels / rows x columns), which is what tells a
whole-tile failure (a kernel's workgroup) from a scattered one.
    WSL_LIB=tools/exp/libwslhip_sp_compiler_chains.so python tools/diff_runs_split.py [reps] [serial]
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wsl4mis_amd import _lib, runtime as rt  # noqa: E402
if os.environ.get("WSL_LIB"):
    _lib.LIB_PATH = os.environ["WSL_LIB"]
from wsl4mis_amd.engine import TrainEngine  # noqa: E402
from wsl4mis_amd.networks.net_factory import net_factory  # noqa: E402
from wsl4mis_amd.synthetic import batch  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
serial = len(sys.argv) > 2 and sys.argv[2] == "serial"
prec = os.environ.get("WSL_PREC", "split_f16x3")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
n, S = 64, 256
torch.manual_seed(2022)
model = net_factory("unet_cct", 1, 4, conv_precision=prec)
model.train()
x, lab = batch(n, S, S, 2022, dev)
gen = torch.Generator().manual_seed(3)
DROP = (0.05, 0.1, 0.2, 0.3, 0.5)
em = [(torch.rand((n, 16 << l, S >> l, S >> l), generator=gen) >= DROP[l]).to(torch.uint8).to(dev) for l in range(5)]
cm = [((torch.rand((n, 16 << l), generator=gen) >= 0.5).float() * 2.0).to(dev) for l in range(5)]
eng = TrainEngine("unet_cct", 1, 4, loss="pce_gatedcrf", crf_radius=5, model=model)
if serial:
    eng.concurrent = False
    rt.L().wsl_net_concurrent(0)
model.set_dropout_masks(em, cm)

L = rt.L()
L.wsl_debug_net_ws_region.restype = C.c_int
L.wsl_debug_net_ws_region.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]


def regions(d):
    out, i = [], 0
    while True:
        nm = C.create_string_buffer(64)
        off, cnt = C.c_size_t(), C.c_size_t()
        if L.wsl_debug_net_ws_region(C.byref(d), i, nm, 64, C.byref(off), C.byref(cnt)) != 0:
            return out
        out.append((nm.value.decode(), off.value, cnt.value))
        i += 1


def shape_of(name, cnt):
    """[N, C, H, W] of an activation-shaped region (None otherwise)"""
    for c in (16, 32, 64, 128, 256, 512, 4, 1):
        for l in range(5):
            if cnt == n * c * (S >> l) * (S >> l):
                return (n, c, S >> l, S >> l)
    return None


def describe(name, a, b):
    d = a != b
    k = int(d.sum())
    rms = float(a.double().pow(2).mean().sqrt())
    dev_ = float((a.double() - b.double())[d].abs().max())
    msg = f"    {name:28s} {k:10d} / {a.numel()} differ; max |delta| {dev_:.3e} = {dev_ / max(rms, 1e-30):.2e} of the region's RMS"
    shp = shape_of(name, a.numel())
    if shp:
        dd = d.view(shp)
        ns = dd.flatten(1).any(1).nonzero().flatten().tolist()
        cs = dd.permute(1, 0, 2, 3).flatten(1).any(1).nonzero().flatten().tolist()
        msg += f"\n        shape {shp}; samples {ns[:12]}{'...' if len(ns) > 12 else ''} ({len(ns)}); channels {cs[:20]}{'...' if len(cs) > 20 else ''} ({len(cs)})"
        # bounding boxes per (sample, channel-block of 16) of the first few
        shown = 0
        for s_ in ns[:3]:
            m2 = dd[s_].any(0)
            rows = m2.any(1).nonzero().flatten()
            cols = m2.any(0).nonzero().flatten()
            msg += f"\n        sample {s_}: rows {int(rows[0])}..{int(rows[-1])} ({len(rows)}), cols {int(cols[0])}..{int(cols[-1])} ({len(cols)}), px {int(m2.sum())}"
            shown += 1
    print(msg, flush=True)


snap0 = None
for rep in range(reps):
    eng.forward_backward(x, lab, 0.37)
    torch.cuda.synchronize()
    d, ws, nws, _, _ = model._saved
    wsf = ws[: (nws // 4) * 4].view(torch.float32)
    regs = regions(d)
    snap = {nm: wsf[off:off + cnt].clone() for nm, off, cnt in regs if not nm.startswith("images")}
    snap["GRAD"] = model.flat_grads().clone()
    if snap0 is None:
        snap0 = snap
        print(f"[{os.environ.get('WSL_LIB', 'product')}] {prec} {'serial' if serial else 'two streams'}: run 0 recorded, {len(regs)} regions", flush=True)
        continue
    bad = [nm for nm in snap if not torch.equal(snap[nm].view(torch.int32), snap0[nm].view(torch.int32))]
    print(f"  run {rep}: {len(bad)} of {len(snap)} regions differ from run 0" + (": " + ", ".join(bad[:40]) if bad else ""), flush=True)
    order = [nm for nm, _, _ in regs if nm in bad][:8] + (["GRAD"] if "GRAD" in bad else [])
    for nm in order:
        describe(nm, snap0[nm], snap[nm])
