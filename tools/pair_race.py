#!/usr/bin/env python3
"""Does kernel A change kernel B's result when both share the chip?  (profiles/r4_sp_root_cause.md; GPU only.)

tools/diff_runs_split.py showed that with hipcc's own v_mfma_f32_16x16x32_f16 chains the FIRST tensor that differs between two
identical training forwards is the output of an f32 1x1 convolution of one decoder -- a kernel without any f16 MFMA -- while the
other decoder's split-precision convolutions run on the side stream.  This harness reproduces such a pair in isolation:
   victim    (stream 1): one launch of a library kernel, repeated R times, its output compared bit for bit with a quiet run;
   aggressor (stream 2): a library kernel launched back to back for the whole time.
Kernels: bil = bilinear x2 upsampling, sm = softmax (no matrix instructions), c3 = f32 direct 3x3 convolution, c1 = f32 1x1 convolution of a BatchNorm source (conv_mfma2l_kernel), sp = split-precision 3x3 forward convolution
(conv_sp_kernel, BatchNorm source), spd = the same on a plain source (data-gradient form), wino = f32 Winograd 3x3 forward.
   WSL_LIB=tools/exp/libwslhip_sp_compiler_chains.so python tools/pair_race.py VICTIM AGGRESSOR [reps]
   e.g.  c1:64,32,32,128,64  sp:64,32,32,256,128       (kind:N,H,W,Ci,Co)
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wsl4mis_amd import _lib  # noqa: E402
if os.environ.get("WSL_LIB"):
    _lib.LIB_PATH = os.environ["WSL_LIB"]
L = _lib.lib()
dev = torch.device("cuda:0")
torch.cuda.set_device(0)


class Case:
    def __init__(self, spec, seed):
        kind, dims = spec.split(":")
        self.kind = kind
        N, H, W, Ci, Co = (int(v) for v in dims.split(","))
        self.dims = (N, H, W, Ci, Co)
        g = torch.Generator().manual_seed(seed)
        self.x = torch.randn(N, Ci, H, W, generator=g).to(dev)
        ks = 1 if kind == "c1" else 3
        if kind == "bil":      # bilinear x2 upsampling of [N, Ci, H, W] (no matrix instruction, LDS-staged): Co ignored
            self.y = torch.zeros(N, Ci, 2 * H, 2 * W, device=dev)
            return
        if kind == "sm":       # softmax over Ci classes of [N, Ci, H * W] (plain vector code)
            self.y = torch.zeros(N, Ci, H, W, device=dev)
            return
        self.w = (torch.randn(Co, Ci, ks, ks, generator=g) * 0.07).to(dev)
        self.scale = (torch.rand(Ci, generator=g) + 0.5).to(dev)
        self.shift = (torch.randn(Ci, generator=g) * 0.3).to(dev)
        self.y = torch.zeros(N, Co, H, W, device=dev)
        s = _lib.WslSrc()
        s.x, s.bs, s.C, s.emask_scale = self.x.data_ptr(), Ci * H * W, Ci, 1.0
        if kind != "spd":
            s.scale, s.shift = self.scale.data_ptr(), self.shift.data_ptr()
        self.src = s
        st = torch.cuda.current_stream().cuda_stream
        if kind in ("sp", "spd"):
            self.img = torch.empty(10 * Ci * Co + 16, device=dev)
            self.wmax = torch.zeros(4, dtype=torch.int64, device=dev)
            _lib.check(L.wsl_sp_pack_weights(self.w.data_ptr(), self.img.data_ptr(), self.wmax.data_ptr(), Co, Ci, 0, st))
            self.amax = torch.full((64,), 0x41000000, dtype=torch.int32, device=dev)    # max |x| = 8.0 in every slot
        elif kind == "wino":
            self.img = torch.empty(16 * Ci * Co + 16, device=dev)
            _lib.check(L.wsl_conv2d_pack_weights(self.w.data_ptr(), self.img.data_ptr(), Co, Ci, 3, 2, st))
        else:
            self.img = torch.empty(ks * ks * Ci * Co + 16, device=dev)
            _lib.check(L.wsl_conv2d_pack_weights(self.w.data_ptr(), self.img.data_ptr(), Co, Ci, ks, 0, st))
        torch.cuda.synchronize()

    def launch(self, stream):
        N, H, W, Ci, Co = self.dims
        st = stream.cuda_stream
        if self.kind == "bil":
            _lib.check(L.wsl_bilinear_up2_fwd(self.x.data_ptr(), self.y.data_ptr(), Ci * 4 * H * W, N, Ci, H, W, st))
            return
        if self.kind == "sm":
            _lib.check(L.wsl_softmax_fwd(self.x.data_ptr(), self.y.data_ptr(), N, Ci, H * W, st))
            return
        if self.kind in ("sp", "spd"):
            _lib.check(L.wsl_sp_conv2d_fwd(C.byref(self.src), None, self.img.data_ptr(), self.wmax.data_ptr(),
                                           self.amax.data_ptr() if self.kind == "spd" else None, None, self.y.data_ptr(), Co * H * W, N, H, W, Co,
                                           None, None, st))
        elif self.kind == "wino":
            _lib.check(L.wsl_conv2d_fwd(C.byref(self.src), None, self.img.data_ptr(), None, self.y.data_ptr(), Co * H * W, N, H, W, Co, 3, 4,
                                        None, None, st))
        else:
            _lib.check(L.wsl_conv2d_fwd(C.byref(self.src), None, self.img.data_ptr(), None, self.y.data_ptr(), Co * H * W, N, H, W, Co,
                                        1 if self.kind == "c1" else 3, 2, None, None, st))


def main():
    victim, aggr = Case(sys.argv[1], 5), Case(sys.argv[2], 6)
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    victim.launch(s1)
    torch.cuda.synchronize()
    ref = victim.y.clone()
    aggr.launch(s2)
    torch.cuda.synchronize()
    aref = aggr.y.clone()
    # quiet repetitions first: is the victim reproducible on its own?
    quiet_bad = 0
    for _ in range(20):
        victim.y.zero_()
        victim.launch(s1)
        torch.cuda.synchronize()
        quiet_bad += int(not torch.equal(victim.y, ref))
    bad, abad, worst, where = 0, 0, 0.0, None
    for r in range(reps):
        with torch.cuda.stream(s1):
            victim.y.zero_()
        for _ in range(3):
            aggr.launch(s2)
        victim.launch(s1)
        for _ in range(3):
            aggr.launch(s2)
        torch.cuda.synchronize()
        if not torch.equal(victim.y, ref):
            bad += 1
            d = (victim.y - ref).abs()
            if float(d.max()) > worst:
                worst = float(d.max())
                idx = (d > 0).nonzero()
                where = (int(idx.shape[0]), torch.unique(idx[:, 0]).tolist()[:8], torch.unique(idx[:, 1]).tolist()[:8],
                         torch.unique(idx[:, 2]).tolist()[:12], torch.unique(idx[:, 3]).tolist()[:12])
        if not torch.equal(aggr.y, aref):
            abad += 1
    rms = float(ref.pow(2).mean().sqrt())
    print(f"[{os.environ.get('WSL_LIB', 'product')}] victim {sys.argv[1]} | aggressor {sys.argv[2]}: quiet {quiet_bad}/20 wrong; "
          f"under load victim {bad}/{reps} wrong (worst |delta| {worst:.3e} = {worst / rms:.2e} RMS), aggressor {abad}/{reps} wrong", flush=True)
    if where:
        print(f"    worst launch: {where[0]} elements; samples {where[1]} channels {where[2]} rows {where[3]} cols {where[4]}", flush=True)


if __name__ == "__main__":
    main()
