"""ctypes binding of libwslhip.so (include/wsl_hip.h).

The product path has NO fallback: if the hipcc-built library is missing this module raises, and every op raises if
the library reports an error.  (tests/emul builds the same sources for a host emulator; that library is loaded by the
tests themselves through `bind()`, never by this loader.)"""
import ctypes as C
import os

# torch must be imported BEFORE libwslhip.so is dlopen'ed: the library needs libamdhip64.so.7, and the process must
# resolve that soname to the HIP runtime torch ships (torch/lib), not to a second copy from /opt/rocm -- with the system
# copy loaded first, torch and the library end up on a runtime that reports "no ROCm-capable device" on the GPU box.
import torch  # noqa: F401,E402

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libwslhip.so")

c_fp = C.c_void_p  # device pointers travel as integers


class WslSrc(C.Structure):
    _fields_ = [("x", c_fp), ("bs", C.c_int64), ("C", C.c_int32), ("_pad0", C.c_int32), ("scale", c_fp),
                ("shift", c_fp), ("emask", c_fp), ("emask_scale", C.c_float), ("_pad1", C.c_float), ("cmask", c_fp)]


WSL_PROF_FAMILIES = 20


class WslProfRow(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("calls", C.c_int64), ("ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double), ("issued_flops", C.c_double)]


class WslNetDesc(C.Structure):
    _fields_ = [("in_chns", C.c_int32), ("n_class", C.c_int32), ("n_dec", C.c_int32), ("N", C.c_int32),
                ("H", C.c_int32), ("W", C.c_int32), ("precision", C.c_int32), ("_pad", C.c_int32)]


class WslWgradPending(C.Structure):
    _fields_ = [("part_dw", c_fp), ("part_db", c_fp), ("dw", c_fp), ("db", c_fp), ("Co", C.c_int32), ("Ci", C.c_int32),
                ("KK", C.c_int32), ("nsplit", C.c_int32)]


class WslUpBlockDesc(C.Structure):
    _fields_ = [("C1", C.c_int32), ("C2", C.c_int32), ("Co", C.c_int32), ("N", C.c_int32), ("h", C.c_int32), ("w", C.c_int32),
                ("dropout_p", C.c_float)]


class WslNetEntry(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("kind", C.c_int32), ("ndim", C.c_int32), ("shape", C.c_int64 * 4),
                ("offset", C.c_int64)]


class WslAugSample(C.Structure):
    _fields_ = [("img", c_fp), ("lab", c_fp), ("h", C.c_int32), ("w", C.c_int32), ("op", C.c_int32), ("k", C.c_int32),
                ("axis", C.c_int32), ("lab_cval", C.c_int32), ("img_cval", C.c_float), ("m00", C.c_double),
                ("m01", C.c_double), ("m10", C.c_double), ("m11", C.c_double), ("off0", C.c_double), ("off1", C.c_double)]


i32, i64, f32, f64, sz = C.c_int, C.c_int64, C.c_float, C.c_double, C.c_size_t
PS, PD, PE = C.POINTER(WslSrc), C.POINTER(WslNetDesc), C.POINTER(WslNetEntry)
PP = C.POINTER(c_fp)

_PROTOS = {
    "wsl_version": (i32, []),
    "wsl_last_error": (C.c_char_p, []),
    "wsl_build_info": (C.c_char_p, []),
    "wsl_prof_enable": (i32, [i32]),
    "wsl_prof_report": (i32, [C.POINTER(WslProfRow), i32]),
    "wsl_net_concurrent": (i32, [i32]),
    "wsl_conv2d_fwd": (i32, [PS, PS, c_fp, c_fp, c_fp, i64, i32, i32, i32, i32, i32, i32, c_fp, c_fp, c_fp]),
    "wsl_conv2d_stat_blocks": (i32, [i32, i32, i32, i32, i32, i32]),
    "wsl_conv2d_pack_weights": (i32, [c_fp, c_fp, i32, i32, i32, i32, c_fp]),
    "wsl_conv2d_fast_ok": (i32, [PS, PS, c_fp, i64, i32]),
    "wsl_conv2d_wino_ok": (i32, [i32, i32, i32, i32, i32, i32, i32]),
    "wsl_conv2d_wgrad": (i32, [PS, PS, c_fp, i64, c_fp, c_fp, i32, i32, i32, i32, i32, c_fp, sz, c_fp]),
    "wsl_conv2d_wgrad_ws_bytes": (sz, [i32, i32, i32, i32, i32, i32]),
    "wsl_conv2d_wgrad_partial": (i32, [PS, PS, c_fp, i64, c_fp, c_fp, i32, i32, i32, i32, i32, c_fp, sz,
                                       C.POINTER(WslWgradPending), c_fp]),
    "wsl_wgrad_reduce_batch": (i32, [C.POINTER(WslWgradPending), i32, c_fp]),
    "wsl_bn_stats_finalize": (i32, [c_fp, c_fp, i32, i32, c_fp, c_fp, f32, f32, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp,
                                    c_fp, c_fp]),
    "wsl_bn_eval_affine": (i32, [c_fp, c_fp, c_fp, c_fp, f32, i32, c_fp, c_fp, c_fp]),
    "wsl_src_materialize": (i32, [PS, c_fp, i64, i32, i32, i32, c_fp]),
    "wsl_pool2_fwd": (i32, [PS, c_fp, i32, i32, i32, c_fp]),
    "wsl_feat_grad_combine": (i32, [PS, c_fp, i64, c_fp, i64, c_fp, c_fp, c_fp, i32, i32, i32, c_fp]),
    "wsl_bnact_bwd": (i32, [c_fp, i64, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, f32, c_fp, c_fp, c_fp, i32, i32, i32, i32,
                            c_fp, sz, c_fp]),
    "wsl_bnact_bwd_ws_bytes": (sz, [i32, i32, i32, i32]),
    "wsl_conv2d_dgrad_bn": (i32, [PS, c_fp, c_fp, i64, i32, i32, i32, i32, i32, i32, c_fp, c_fp, c_fp, f32, c_fp, C.POINTER(C.c_int), c_fp]),
    "wsl_conv2d_dgrad_bn_d": (i32, [PS, c_fp, c_fp, i64, i32, i32, i32, i32, i32, i32, c_fp, c_fp, c_fp, f32, c_fp, C.POINTER(C.c_int), c_fp]),
    "wsl_bnact_bwd_finish_d_amax": (i32, [c_fp, i64, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, i32, i32, i32, i32,
                                          c_fp, i32, i32, c_fp, sz, c_fp, c_fp]),
    "wsl_feat_grad_combine_blocks": (i32, [i32, i32, i32]),
    "wsl_feat_grad_combine_bn": (i32, [PS, c_fp, i64, c_fp, i64, c_fp, c_fp, c_fp, i32, i32, i32, c_fp, c_fp, c_fp, c_fp]),
    "wsl_bnact_bwd_finish": (i32, [c_fp, i64, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, f32, c_fp, c_fp, c_fp, i32, i32, i32, i32,
                                   c_fp, i32, i32, c_fp, sz, c_fp]),
    # split-precision conv path (opt-in)
    "wsl_sp_conv2d_ok": (i32, [PS, PS, c_fp, i64, i32, i32, i32, i32, i32]),
    "wsl_sp_weight_image_bytes": (sz, [i32, i32]),
    "wsl_sp_pack_weights": (i32, [c_fp, c_fp, c_fp, i32, i32, i32, c_fp]),
    "wsl_sp_conv2d_fwd": (i32, [PS, PS, c_fp, c_fp, c_fp, c_fp, c_fp, i64, i32, i32, i32, i32, c_fp, c_fp, c_fp]),
    "wsl_sp_conv2d_stat_blocks": (i32, [i32, i32, i32, i32, i32]),
    "wsl_sp_conv2d_dgrad_bn": (i32, [PS, c_fp, c_fp, c_fp, c_fp, i64, i32, i32, i32, i32, c_fp, c_fp, c_fp, f32, c_fp,
                                     C.POINTER(C.c_int), c_fp]),
    "wsl_sp_conv2d_wgrad_ws_bytes": (sz, [i32, i32, i32, i32, i32]),
    "wsl_sp_conv2d_wgrad_partial": (i32, [PS, PS, c_fp, i64, c_fp, c_fp, c_fp, i32, i32, i32, i32, c_fp, sz,
                                          C.POINTER(WslWgradPending), c_fp]),
    "wsl_sp_conv2d_wgrad_partial_amax": (i32, [PS, PS, c_fp, i64, c_fp, c_fp, c_fp, c_fp, i32, i32, i32, i32, c_fp, sz,
                                               C.POINTER(WslWgradPending), c_fp]),
    "wsl_bilinear_up2_fwd_amax_ws_bytes": (sz, [i32, i32, i32, i32]),
    "wsl_bilinear_up2_fwd_amax": (i32, [c_fp, c_fp, i64, i32, i32, i32, i32, c_fp, sz, c_fp, c_fp]),
    "wsl_bnact_bwd_finish_ws_bytes": (sz, [i32, i32, i32, i32, i32]),
    "wsl_bnact_bwd_amax": (i32, [c_fp, i64, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, f32, c_fp, c_fp, c_fp, i32, i32, i32, i32,
                                 c_fp, sz, c_fp, c_fp]),
    "wsl_bnact_bwd_finish_amax": (i32, [c_fp, i64, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, f32, c_fp, c_fp, c_fp, i32, i32, i32, i32,
                                        c_fp, i32, i32, c_fp, sz, c_fp, c_fp]),
    "wsl_bilinear_up2_fwd": (i32, [c_fp, c_fp, i64, i32, i32, i32, i32, c_fp]),
    "wsl_bilinear_up2_bwd": (i32, [c_fp, i64, c_fp, i32, i32, i32, i32, c_fp]),
    "wsl_softmax_fwd": (i32, [c_fp, c_fp, i32, i32, i32, c_fp]),
    "wsl_softmax_bwd": (i32, [c_fp, c_fp, c_fp, i32, i32, i32, c_fp]),
    "wsl_ce_fwd_bwd": (i32, [c_fp, c_fp, i32, i32, c_fp, c_fp, f32, i32, i32, i32, c_fp, sz, c_fp]),
    "wsl_mix_argmax": (i32, [c_fp, c_fp, f64, c_fp, i32, i32, i32, c_fp]),
    "wsl_pdice_fwd": (i32, [c_fp, c_fp, i32, i32, c_fp, c_fp, i32, i32, i32, c_fp, sz, c_fp]),
    "wsl_pdice_bwd": (i32, [c_fp, c_fp, i32, i32, c_fp, c_fp, c_fp, i32, i32, i32, c_fp]),
    "wsl_head_fwd_bwd": (i32, [c_fp, c_fp, c_fp, i32, f64, f32, f32, c_fp, c_fp, c_fp, c_fp, i32, i32, i32, c_fp, sz,
                               c_fp]),
    "wsl_loss_ws_bytes": (sz, [i32, i32, i32]),
    "wsl_mixprob_fwd": (i32, [c_fp, c_fp, f64, c_fp, i32, i32, i32, c_fp]),
    "wsl_mixprob_bwd": (i32, [c_fp, c_fp, f64, c_fp, f32, c_fp, c_fp, i32, i32, i32, i32, c_fp]),
    "wsl_gatedcrf_fwd": (i32, [c_fp, c_fp, c_fp, c_fp, i32, i32, i32, i32, i32, f32, f32, f32, c_fp, sz, c_fp]),
    "wsl_gatedcrf_bwd": (i32, [c_fp, c_fp, f32, c_fp, i32, i32, i32, i32, c_fp]),
    "wsl_head_reg_fwd_bwd": (i32, [c_fp, c_fp, i32, f32, i32, f32, c_fp, c_fp, f32, c_fp, c_fp, c_fp, c_fp, i32, i32, i32, i32, c_fp, sz, c_fp]),
    "wsl_tv_fwd_bwd": (i32, [c_fp, i32, c_fp, c_fp, f32, i32, i32, i32, i32, c_fp, sz, c_fp]),
    "wsl_mumford_shah_fwd_bwd": (i32, [c_fp, c_fp, c_fp, c_fp, f32, i32, i32, i32, i32, c_fp, sz, c_fp]),
    "wsl_softmax_mse_fwd_bwd": (i32, [c_fp, c_fp, c_fp, c_fp, f32, i32, i32, i32, c_fp, sz, c_fp]),
    "wsl_rot90": (i32, [c_fp, c_fp, i32, i32, i32, i32, c_fp]),
    "wsl_softmax_accum": (i32, [c_fp, c_fp, f32, i32, i32, i32, i32, c_fp]),
    "wsl_ustm_consistency_fwd_bwd": (i32, [c_fp, c_fp, c_fp, f32, c_fp, c_fp, f32, i32, i32, i32, c_fp, sz, c_fp]),
    "wsl_entropy_fwd_bwd": (i32, [c_fp, c_fp, c_fp, f32, i32, i32, i32, i32, c_fp, sz, c_fp]),
    "wsl_axpy": (i32, [c_fp, c_fp, f32, i64, c_fp]),
    "wsl_sgd_step": (i32, [c_fp, c_fp, c_fp, i64, f32, f32, f32, i32, f32, c_fp, f32, c_fp]),
    "wsl_surface_u8": (i32, [c_fp, c_fp, i32, i32, i32, c_fp]),
    "wsl_nearest_dist2": (i32, [c_fp, i32, c_fp, i32, c_fp, c_fp]),
    "wsl_augment_batch": (i32, [C.POINTER(WslAugSample), i32, c_fp, c_fp, i32, i32, c_fp]),
    "wsl_noisy_copy": (i32, [c_fp, c_fp, c_fp, i64, i32, f32, f32, C.c_uint64, c_fp]),
    "wsl_head_gatedcrf_fwd_bwd": (i32, [c_fp, c_fp, c_fp, i32, f64, c_fp, i32, f32, f32, f32, f32, c_fp, c_fp, c_fp, c_fp, c_fp,
                                        i32, i32, i32, i32, c_fp, sz, c_fp]),
    "wsl_draw_masks": (i32, [i32, PP, C.POINTER(C.c_int64), C.POINTER(C.c_float), C.POINTER(C.c_float),
                             C.POINTER(C.c_int), C.c_uint64, c_fp]),
    "wsl_convt2x2_fwd": (i32, [c_fp, c_fp, c_fp, c_fp, i32, i32, i32, i32, i32, c_fp]),
    "wsl_convt2x2_dgrad": (i32, [c_fp, i64, c_fp, c_fp, i32, i32, i32, i32, i32, c_fp]),
    "wsl_convt2x2_wgrad_ws_bytes": (sz, [i32, i32, i32]),
    "wsl_convt2x2_wgrad": (i32, [c_fp, c_fp, i64, c_fp, c_fp, i32, i32, i32, i32, i32, c_fp, sz, c_fp]),
    "wsl_upblock_t_param_count": (i64, [C.POINTER(WslUpBlockDesc)]),
    "wsl_upblock_t_ws_bytes": (sz, [C.POINTER(WslUpBlockDesc)]),
    "wsl_upblock_t_forward": (i32, [C.POINTER(WslUpBlockDesc), c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, i32, c_fp, c_fp, sz, c_fp]),
    "wsl_upblock_t_backward": (i32, [C.POINTER(WslUpBlockDesc), c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, sz, c_fp]),
    "wsl_net_num_entries": (i32, [PD]),
    "wsl_net_entry": (i32, [PD, i32, PE]),
    "wsl_net_param_count": (i64, [PD]),
    "wsl_net_encoder_param_count": (i64, [PD]),
    "wsl_net_buffer_count": (i64, [PD]),
    "wsl_net_ws_bytes": (sz, [PD]),
    "wsl_net_forward": (i32, [PD, c_fp, c_fp, c_fp, c_fp, PP, PP, i32, c_fp, c_fp, c_fp, sz, c_fp]),
    "wsl_net_backward": (i32, [PD, c_fp, c_fp, PP, PP, c_fp, c_fp, c_fp, c_fp, sz, i32, c_fp]),
}


# private hooks (wsl4mis_amd/csrc/wsl_debug.h): bound when the loaded library has them; tests and tools only
_DEBUG_PROTOS = {
    "wsl_debug_conv_plan": (i32, [i32, i32, i32]),        # (experiments build / emulator only)
    "wsl_debug_conv_wino": (i32, [i32]),
    "wsl_debug_wgrad_workgroups": (i32, [i32]),
    "wsl_debug_net_decisions": (i32, [PD, c_fp, sz, i32, i32, c_fp, c_fp]),
    "wsl_debug_mfma4_probe": (i32, [c_fp, c_fp, c_fp, c_fp]),
    "wsl_debug_lds_dma_probe": (i32, [c_fp, c_fp, c_fp]),
    "wsl_debug_pk_probe": (i32, [c_fp, c_fp, c_fp]),
    "wsl_debug_mfma_stream": (i32, [i32, i32, i32, c_fp, c_fp]),
}


class WslError(RuntimeError):
    pass


def bind(cdll, strict=True):
    """Attach prototypes of include/wsl_hip.h to a loaded library; returns the list of missing symbols."""
    missing = []
    for name, (res, args) in _PROTOS.items():
        try:
            fn = getattr(cdll, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype, fn.argtypes = res, args
    for name, (res, args) in _DEBUG_PROTOS.items():
        if hasattr(cdll, name):
            fn = getattr(cdll, name)
            fn.restype, fn.argtypes = res, args
    if missing and strict:
        raise WslError(f"library lacks symbols declared in include/wsl_hip.h: {missing}")
    return missing


_lib = None


def lib():
    """The product library.  Raises (never falls back) if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise WslError(f"{LIB_PATH} not found: run wsl4mis_amd/csrc/build.sh (hipcc, gfx950). "
                           "There is no CPU fallback.")
        cdll = C.CDLL(LIB_PATH)
        # (another build handed in for A/B timing -- bench.py --lib, tools/explib.py -- may predate entry points of this tree: it is bound
        #  as far as it goes; the in-tree product library must export every declared symbol)
        bind(cdll, strict=os.path.abspath(LIB_PATH) == os.path.join(_HERE, "csrc", "libwslhip.so"))
        if b"HOST-EMULATION" in cdll.wsl_build_info():
            raise WslError("refusing to use a host-emulation build as the product library")
        _lib = cdll
    return _lib


_test_emul = False


def use_library_for_tests(cdll):
    """TEST HOOK ONLY: make the Python layer talk to the host-emulation build (tests/emul) with CPU tensors so the
    host logic can be exercised without a GPU.  Never called by the package or __graft_entry__; bench.py calls it only under its
    --selftest-emulator flag (the launch / rendezvous logic of `bench.py --gpus N` checked on the CPU by tests/test_dp.py: value null)."""
    global _lib, _test_emul
    bind(cdll, strict=False)
    if b"HOST-EMULATION" not in cdll.wsl_build_info():
        raise WslError("use_library_for_tests expects the emulation build")
    _lib, _test_emul = cdll, True


def _reset_for_tests():
    global _lib, _test_emul
    _lib, _test_emul = None, False


def is_test_emulation():
    return _test_emul


def check(rc, cdll=None):
    if rc != 0:
        l = cdll or lib()
        raise WslError(f"wsl error {rc}: {l.wsl_last_error().decode()}")


def source_sha256():
    """SHA-256 over the library's sources as csrc/build.sh hashes them (csrc/*.hip and csrc/*.h sorted by name, then include/wsl_hip.h):
    what wsl_build_info() of a library built from THIS tree must end in."""
    import hashlib
    cs = os.path.join(_HERE, "csrc")
    names = sorted(f for f in os.listdir(cs) if f.endswith((".hip", ".h")))
    h = hashlib.sha256()
    for f in [os.path.join(cs, n) for n in names] + [os.path.join(os.path.dirname(_HERE), "include", "wsl_hip.h")]:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def library_sha256(cdll=None):
    """the source hash compiled into a loaded library (None when it carries none)"""
    info = (cdll or lib()).wsl_build_info().decode()
    return info.rsplit("sha256:", 1)[1].strip() if "sha256:" in info else None


def declared_symbols():
    return list(_PROTOS)
