"""Two ways to run the same C-ABI calls in the tests:
  EmulBackend -- tests/emul/libwslhip_emul.so, the kernel sources compiled for the lock-step host emulator
                 (logic check, CPU, tiny shapes).  Test infrastructure only.
  HipBackend  -- the product library on cuda:0 (tests marked gpu).
  HipExpBackend -- tools/exp/libwslhip_exp.so on cuda:0: the same sources built with -DWSL_EXPERIMENTS, the only hipcc build that has
                 routing overrides (wsl_debug_conv_plan / _conv_wino / _wgrad_workgroups).  Only the tests that FORCE a route use it
                 (kernel instantiations / tile walks at sizes the product's plan would not pick); every default-route test runs on the
                 product library."""
import ctypes as C
import os

import numpy as np

from wsl4mis_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMUL_PATH = os.path.join(ROOT, "tests", "emul", "libwslhip_emul.so")
EXP_PATH = os.path.join(ROOT, "tools", "exp", "libwslhip_exp.so")


class _Base:
    def src(self, x=None, C_=0, bs=None, scale=None, shift=None, emask=None, es=1.0, cmask=None):
        s = _lib.WslSrc()
        if x is None:
            return s
        s.x, s.C = self.ptr(x), C_
        s.bs = bs if bs is not None else int(np.prod(self.shape(x)[1:]))
        s.scale = self.ptr(scale) if scale is not None else None
        s.shift = self.ptr(shift) if shift is not None else None
        s.emask = self.ptr(emask) if emask is not None else None
        s.emask_scale = es
        s.cmask = self.ptr(cmask) if cmask is not None else None
        return s

    def call(self, name, *args):
        rc = getattr(self.lib, name)(*args)
        if rc != 0:
            raise _lib.WslError(f"{name} -> {rc}: {self.lib.wsl_last_error().decode()}")

    def ws(self, nbytes):
        return self.zeros(((max(int(nbytes), 16) + 3) // 4,), np.float32)


class EmulBackend(_Base):
    name = "emul"
    stream = None

    def __init__(self, path=EMUL_PATH):
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = C.CDLL(path)
        _lib.bind(self.lib, strict=False)

    def arr(self, a, dtype=None):
        return np.ascontiguousarray(np.asarray(a, dtype=dtype)).copy()

    def zeros(self, shape, dtype=np.float32):
        return np.zeros(shape, dtype=dtype)

    def np(self, a):
        return np.asarray(a)

    def ptr(self, a):
        return a.ctypes.data

    def shape(self, a):
        return a.shape

    def sync(self):
        pass


class HipBackend(_Base):
    name = "hip"

    def __init__(self):
        import torch
        self.torch = torch
        assert torch.cuda.is_available(), "HipBackend needs a GPU"
        self.lib = _lib.lib()
        self.dev = torch.device("cuda:0")

    @property
    def stream(self):
        return self.torch.cuda.current_stream().cuda_stream

    def arr(self, a, dtype=None):
        return self.torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=dtype))).to(self.dev)

    def zeros(self, shape, dtype=np.float32):
        td = {np.float32: self.torch.float32, np.uint8: self.torch.uint8, np.int64: self.torch.int64,
              np.float64: self.torch.float64}[dtype]
        return self.torch.zeros(tuple(shape), dtype=td, device=self.dev)

    def np(self, a):
        return a.detach().cpu().numpy()

    def ptr(self, a):
        return a.data_ptr()

    def shape(self, a):
        return tuple(a.shape)

    def sync(self):
        self.torch.cuda.synchronize()


class HipExpBackend(HipBackend):
    name = "hip"          # same device, tensors and tolerances as the product backend

    def __init__(self, path=EXP_PATH):
        import torch
        self.torch = torch
        assert torch.cuda.is_available(), "HipExpBackend needs a GPU"
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path}: run wsl4mis_amd/csrc/build.sh exp (__graft_entry__.build() does)")
        self.lib = C.CDLL(path)
        _lib.bind(self.lib)
        info = self.lib.wsl_build_info()
        assert b"EXPERIMENTS" in info, info
        assert _lib.library_sha256(self.lib) == _lib.source_sha256(), "tools/exp/libwslhip_exp.so was not built from this tree"
        self.dev = torch.device("cuda:0")
