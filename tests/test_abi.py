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
    info = lib.wsl_build_info()                                    # host-only calls: no kernel is launched here
    assert info.startswith(b"gfx950 hipcc sha256:") and b"EXPERIMENTS" not in info and b"EMULATION" not in info, info
    lib.wsl_version.restype = ctypes.c_int
    assert lib.wsl_version() >= 100


def test_product_library_is_built_from_this_tree():
    """libwslhip.so is git-ignored and travels prebuilt -- wsl_build_info() carries the SHA-256 of the sources it was
    compiled from (csrc/build.sh puts it on wsl_api.hip's command line and rebuilds objects by content hash, not by modification time);
    it must equal the hash of the tree the tests run in.  bench.py checks the same and prints it as config.library_sha."""
    from wsl4mis_amd import _lib
    lib = ctypes.CDLL(LIB)
    lib.wsl_build_info.restype = ctypes.c_char_p
    assert _lib.library_sha256(lib) == _lib.source_sha256(), "libwslhip.so is stale: run wsl4mis_amd/csrc/build.sh"


def test_product_library_has_no_routing_hooks():
    """no process-global switch that changes which kernel a launch takes -- the routing overrides and the
    ablation template arms exist only with -DWSL_EXPERIMENTS (tools/exp/libwslhip_exp.so, the host emulator)."""
    lib = ctypes.CDLL(LIB)
    for hook in ("wsl_debug_conv_plan", "wsl_debug_conv_wino", "wsl_debug_wgrad_workgroups", "wsl_debug_mfma_stream", "wsl_debug_pk_probe"):
        assert not hasattr(lib, hook), hook
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import scan_vop3p
    nm = os.path.join(os.path.dirname(scan_vop3p.OBJDUMP), "llvm-nm")
    if os.path.exists(nm):
        syms = subprocess.run([nm, "-D", "--defined-only", LIB], check=True, capture_output=True, text=True).stdout
        assert "g_forced" not in syms and "g_wino" not in syms
        dbg = sorted(ln.split()[-1] for ln in syms.splitlines() if "wsl_debug_" in ln)
        assert dbg == ["wsl_debug_net_decisions", "wsl_debug_net_ws_region", "wsl_debug_sp_conv_residency"], dbg   # read-only queries


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


def test_no_packed_f32_forms_unsafe_next_to_f16_mfma():
    """gfx950: v_pk_fma_f32 / v_pk_mul_f32 with op_sel on src1 / src2 return a wrong low half while another wave of the SIMD issues
    v_mfma_f32_16x16x32_f16 (profiles/r4_sp_root_cause.md) -- the root cause of round 3's run-to-run different gradients of the
    split-precision path.  No kernel of the product library may contain such an instruction (the disassembly is scanned; no GPU needed)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import scan_vop3p
    assert scan_vop3p.unsafe("v_pk_fma_f32 v[76:77], v[2:3], v[74:75], v[74:75] op_sel:[0,0,1] op_sel_hi:[1,0,1]")
    assert scan_vop3p.unsafe("v_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1]")
    assert not scan_vop3p.unsafe("v_pk_fma_f32 v[88:89], v[46:47], v[78:79], v[80:81] op_sel_hi:[1,0,0]")
    assert not scan_vop3p.unsafe("v_pk_add_f32 v[0:1], v[2:3], v[2:3] op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]")
    if not os.path.exists(scan_vop3p.OBJDUMP):
        pytest.skip("llvm-objdump of the ROCm toolchain not found")
    total, bad = scan_vop3p.scan(LIB)
    assert total > 1000, total                                   # the scan sees the device code
    assert not bad, f"{len(bad)} unsafe packed-f32 instructions, e.g. {bad[:3]}"


def test_kernels_of_the_step_do_not_spill_beyond_the_known_few():
    """scratch (private segment) per kernel, read from the code objects' metadata.  Round 4 removed the
    scratch of the loss head (address-taken per-class arrays) and of the narrow-K convolution (register cap 3 -> 2: classifier data
    gradient 134 -> 97 us).  What is left is listed here with its size, so that a change that makes a hot kernel spill fails:
    a handful of registers in kernels that sit exactly at a register cap which was MEASURED faster with the cap (the 128-tile Winograd
    form: 256 registers), and the generic fallbacks no step of the networks launches.  The split-precision conv kernels must not
    spill at all: a spill reload inside their tile loop is a vector-memory load, and vmcnt being in-order it waits for every prefetch
    in flight (they were made spill-free by one epilogue per instantiation, per-item operand offsets and a scheduling fence in the
    64-wide form)."""
    import subprocess
    import sys
    import tempfile
    import shutil
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import scan_vop3p
    if not os.path.exists(scan_vop3p.OBJDUMP):
        pytest.skip("llvm-objdump of the ROCm toolchain not found")
    readelf = os.path.join(os.path.dirname(scan_vop3p.OBJDUMP), "llvm-readelf")
    allowed = {   # kernel-name fragment -> bytes of scratch tolerated
        "conv_sp_kernelILi8ELi16ELi16ELb0ELi2": 32,   # a shape no layer of the networks has (16-wide block of > 64 input channels)
        "conv_wino2_kernelILi8ELi64ELi1": 16, "conv_wino2r_kernelILi8ELi64ELi1": 16,
        "conv_mfma2_kernel": 64,            # generic direct fallback (odd shapes in tests; not launched by a 16-aligned network)
        "head_reduce_kernelILi0": 176,      # generic class count (C != 4): per-class arrays indexed at run time
    }
    bad = []
    with tempfile.TemporaryDirectory() as work:
        for co in scan_vop3p.code_objects(LIB, work):
            notes = subprocess.run([readelf, "--notes", co], check=True, capture_output=True, text=True).stdout
            for blk in notes.split("- .agpr_count")[1:]:
                name = re.search(r"\.name:\s+(\S+)", blk).group(1)
                scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1))
                if scratch == 0:
                    continue
                lim = max([v for k, v in allowed.items() if k in name] or [0])
                if scratch > lim:
                    bad.append((name, scratch, lim))
    assert not bad, bad
