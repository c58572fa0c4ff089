"""The drop-in boundary: the hipcc-built C-ABI library loads on a machine without a GPU and exports every entry point
include/wsl_hip.h declares; the ctypes binding covers them; the product loader refuses the host-emulation build."""
import ctypes
import os
import re

import pytest
import torch  # noqa: F401  (the HIP runtime the library links against is the one torch ships)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "wsl_hip.h")
LIB = os.path.join(ROOT, "wsl4mis_amd", "csrc", "libwslhip.so")


def declared():
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    names = re.findall(r"^\s*(?:const\s+)?[A-Za-z_][\w\s\*]*?\b(wsl_\w+)\s*\(", src, flags=re.M)
    assert len(names) > 40, names
    return sorted(set(names))


def test_product_library_exports_every_declared_symbol():
    if not os.path.exists(LIB):
        pytest.fail("wsl4mis_amd/csrc/libwslhip.so is not built (run __graft_entry__.build())")
    lib = ctypes.CDLL(LIB)
    missing = [n for n in declared() if not hasattr(lib, n)]
    assert not missing, missing
    lib.wsl_build_info.restype = ctypes.c_char_p
    assert lib.wsl_build_info() == b"gfx950 hipcc"              # host-only calls: no kernel is launched here
    lib.wsl_version.restype = ctypes.c_int
    assert lib.wsl_version() >= 100


def test_ctypes_binding_covers_the_header():
    from wsl4mis_amd import _lib
    unbound = [n for n in declared() if n not in _lib._PROTOS]
    assert not unbound, unbound
    ghosts = [n for n in _lib._PROTOS if n not in declared()]
    assert not ghosts, ghosts


def test_product_loader_refuses_the_emulation_build():
    from conftest import get_backend
    emu = get_backend("emul").lib
    emu.wsl_build_info.restype = ctypes.c_char_p
    assert b"HOST-EMULATION" in emu.wsl_build_info()
    missing = [n for n in declared() if not hasattr(emu, n)]     # the emulator builds the very same sources
    assert not missing, missing
    from wsl4mis_amd import _lib
    _lib._reset_for_tests()
    real, _lib.LIB_PATH = _lib.LIB_PATH, os.path.join(ROOT, "tests", "emul", "libwslhip_emul.so")
    try:
        with pytest.raises(_lib.WslError, match="host-emulation"):
            _lib.lib()
        _lib.LIB_PATH = os.path.join(ROOT, "no_such_dir", "libwslhip.so")
        with pytest.raises(_lib.WslError, match="no CPU fallback"):
            _lib.lib()
    finally:
        _lib.LIB_PATH = real
        _lib._reset_for_tests()
