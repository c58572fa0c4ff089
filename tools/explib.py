"""Tuning tools only: point the ctypes loader at tools/exp/libwslhip_exp.so (`wsl4mis_amd/csrc/build.sh exp`: the same
sources with -DWSL_EXPERIMENTS, i.e. with the WSL_* environment knobs, ablation switches and machine probes that the product
library does not contain).  Never imported by wsl4mis_amd/, bench.py or the tests."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# WSL_EXP_LIB=old: tools/exp/libwslhip_exp_old.so, a build of another source revision kept next to it for A/B timing
_sel = os.environ.get("WSL_EXP_LIB")   # "old" or any other tag: tools/exp/libwslhip_exp_<tag>.so
EXP = os.path.join(ROOT, "tools", "exp", f"libwslhip_exp_{_sel}.so" if _sel else "libwslhip_exp.so")


def use():
    from wsl4mis_amd import _lib
    if not os.path.exists(EXP):
        raise SystemExit(f"{EXP} not found: run wsl4mis_amd/csrc/build.sh exp")
    _lib.LIB_PATH = EXP
    return _lib
