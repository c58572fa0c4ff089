"""Device / stream plumbing shared by the host-side mirror of the reference interface.

PyTorch is used for device memory, streams and torch.distributed only; arithmetic goes through libwslhip.so.
There is no CPU path: `device()` raises without a GPU.  (The test-suite can inject the host-emulation build with
`_lib.use_library_for_tests`; that is the only way the modules will accept CPU tensors.)"""
import ctypes as C

import torch

from . import _lib


def device():
    if _lib.is_test_emulation():
        return torch.device("cpu")
    if not torch.cuda.is_available():
        raise _lib.WslError("wsl4mis_amd needs an AMD GPU (MI355X / gfx950); there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def stream():
    return None if _lib.is_test_emulation() else torch.cuda.current_stream().cuda_stream


def L():
    return _lib.lib()


def call(name, *args):
    rc = getattr(L(), name)(*args)
    if rc != 0:
        raise _lib.WslError(f"{name} failed ({rc}): {L().wsl_last_error().decode()}")


def ptr(t):
    return None if t is None else t.data_ptr()


def f32c(t, what="tensor"):
    """fp32, contiguous, on the engine's device -- or raise (never silently convert across devices)."""
    if t.device != device():
        raise _lib.WslError(f"{what} lives on {t.device}, expected {device()}")
    if t.dtype != torch.float32:
        raise _lib.WslError(f"{what} has dtype {t.dtype}, expected float32")
    return t if t.is_contiguous() else t.contiguous()


_ws_cache = {}


def workspace(key, nbytes):
    """Grow-only scratch tensor per (device, key)."""
    k = (str(device()), key)
    t = _ws_cache.get(k)
    if t is None or t.numel() < nbytes:
        t = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device())
        _ws_cache[k] = t
    return t


def ptr_array(tensors):
    if tensors is None:
        return None
    return (C.c_void_p * len(tensors))(*[ptr(t) for t in tensors])
