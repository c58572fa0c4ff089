"""Kernels of the two decoder streams share compute units: no kernel's result may depend on what runs beside it.

Round 3's split-precision path gave gradients that were wrong and different from run to run; round 4 found the cause
(profiles/r4_sp_root_cause.md): on gfx950 a packed-f32 FMA / multiply with `op_sel` on src1 / src2 -- the form hipcc built for the
BatchNorm transform of the f32 kernels' loaders -- returns a wrong low half while another wave of the SIMD issues
v_mfma_f32_16x16x32_f16, i.e. whenever the OTHER decoder's split-precision convolution shared the CU.  tests/test_abi.py scans the
library's disassembly for the instruction form; this file tests the behaviour on the device, per kernel pair, at the shapes of the
benchmark step: every victim launch beside an aggressor on a second stream must equal its quiet launch bit for bit, and so must the
aggressor.  (Both orders of each pair; the split kernels are themselves f16-MFMA aggressors for each other.)"""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
N = 64


class _Case:
    """one library kernel with fixed inputs: kind c1 = f32 1x1 conv, c3 = f32 direct 3x3 conv, wino = f32 Winograd 3x3 conv (all with a
    BatchNorm + LeakyReLU source), sp = split-precision 3x3 conv of a BatchNorm source, spd = of a plain (gradient) source,
    spw = split-precision weight gradient"""

    def __init__(self, kind, H, Ci, Co, seed):
        from wsl4mis_amd import _lib, runtime as rt
        self.kind, self.dims, self.rt, self.lib = kind, (N, H, H, Ci, Co), rt, _lib
        L, dev = rt.L(), rt.device()
        g = torch.Generator().manual_seed(seed)
        self.x = torch.randn(N, Ci, H, H, generator=g).to(dev)
        ks = 1 if kind == "c1" else 3
        self.w = (torch.randn(Co, Ci, ks, ks, generator=g) * 0.07).to(dev)
        self.scale, self.shift = (torch.rand(Ci, generator=g) + 0.5).to(dev), (torch.randn(Ci, generator=g) * 0.3).to(dev)
        self.y = torch.zeros(N, Co, H, H, device=dev)
        s = _lib.WslSrc()
        s.x, s.bs, s.C, s.emask_scale = self.x.data_ptr(), Ci * H * H, Ci, 1.0
        if kind != "spd":
            s.scale, s.shift = self.scale.data_ptr(), self.shift.data_ptr()
        self.src = s
        st = rt.stream()
        self.amax = torch.full((64,), 0x41000000, dtype=torch.int32, device=dev)        # max |x| = 8.0 in every slot
        if kind in ("sp", "spd"):
            self.img = torch.empty(10 * Ci * Co + 16, device=dev)
            self.wmax = torch.zeros(4, dtype=torch.int64, device=dev)
            rt.call("wsl_sp_pack_weights", self.w.data_ptr(), self.img.data_ptr(), self.wmax.data_ptr(), Co, Ci, 0, st)
        elif kind == "spw":
            self.dy = (torch.randn(N, Co, H, H, generator=g) * 0.5).to(dev)
            self.nws = L.wsl_sp_conv2d_wgrad_ws_bytes(N, H, H, Ci, Co)
            self.ws = torch.zeros(self.nws // 4 + 16, device=dev)
            self.dw, self.db = torch.zeros(Co, Ci, 3, 3, device=dev), torch.zeros(Co, device=dev)
            self.pending = _lib.WslWgradPending()
            self.y = self.ws                                                            # what is compared: the partials
        elif kind == "wino":
            self.img = torch.empty(16 * Ci * Co + 16, device=dev)
            rt.call("wsl_conv2d_pack_weights", self.w.data_ptr(), self.img.data_ptr(), Co, Ci, 3, 2, st)
        else:
            self.img = torch.empty(ks * ks * Ci * Co + 16, device=dev)
            rt.call("wsl_conv2d_pack_weights", self.w.data_ptr(), self.img.data_ptr(), Co, Ci, ks, 0, st)
        torch.cuda.synchronize()

    def launch(self, stream):
        n, H, W, Ci, Co = self.dims
        st, rt = stream.cuda_stream, self.rt
        if self.kind in ("sp", "spd"):
            rt.call("wsl_sp_conv2d_fwd", C.byref(self.src), None, self.img.data_ptr(), self.wmax.data_ptr(),
                    self.amax.data_ptr() if self.kind == "spd" else None, None, self.y.data_ptr(), Co * H * W, n, H, W, Co, None, None, st)
        elif self.kind == "spw":
            rt.call("wsl_sp_conv2d_wgrad_partial", C.byref(self.src), None, self.dy.data_ptr(), Co * H * W, self.amax.data_ptr(),
                    self.dw.data_ptr(), self.db.data_ptr(), n, H, W, Co, self.ws.data_ptr(), self.nws, C.byref(self.pending), st)
        else:
            ks, mode = (1, 2) if self.kind == "c1" else (3, 4 if self.kind == "wino" else 2)
            rt.call("wsl_conv2d_fwd", C.byref(self.src), None, self.img.data_ptr(), None, self.y.data_ptr(), Co * H * W, n, H, W, Co, ks,
                    mode, None, None, st)


# pairs met in the unet_cct step when the two decoders overlap (shape = level of the step); each is run in both roles
PAIRS = [
    (("c1", 32, 128, 64), ("sp", 32, 256, 128)),      # up2's 1x1 conv beside the other decoder's first block conv
    (("c1", 16, 256, 128), ("spw", 32, 128, 128)),    # ... beside a split weight gradient
    (("wino", 256, 16, 16), ("spd", 64, 64, 64)),     # the 16-channel f32 layers (their weight gradients stay Winograd) beside a split data gradient
    (("c3", 64, 64, 32), ("sp", 64, 128, 64)),        # the f32 direct 3x3 kernel (generic shapes) beside a split conv
    (("sp", 32, 128, 128), ("spw", 64, 64, 64)),      # split kernels beside each other
]


@pytest.mark.parametrize("pair", PAIRS, ids=lambda p: f"{p[0][0]}{p[0][1]}-{p[1][0]}{p[1][1]}")
def test_kernel_results_do_not_depend_on_the_other_stream(pair):
    from wsl4mis_amd import _lib, runtime
    _lib._reset_for_tests()
    runtime._ws_cache.clear()
    a, b = _Case(*pair[0], seed=5), _Case(*pair[1], seed=6)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    refs = []
    for c in (a, b):
        c.launch(s1)
        torch.cuda.synchronize()
        refs.append(c.y.clone())
    bad = [0, 0]
    for rep in range(12):
        for c in (a, b):
            with torch.cuda.stream(s1):
                c.y.zero_()
        torch.cuda.synchronize()
        for _ in range(2):                   # interleaved submissions: each kernel meets the other at its start, middle and tail
            b.launch(s2)
            a.launch(s1)
            b.launch(s2)
        torch.cuda.synchronize()
        bad[0] += int(not torch.equal(a.y, refs[0]))
        bad[1] += int(not torch.equal(b.y, refs[1]))
    assert bad == [0, 0], (pair, bad)
