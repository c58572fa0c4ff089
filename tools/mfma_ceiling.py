#!/usr/bin/env python3
"""Practical f32 MFMA ceiling: a pure MFMA stream (no memory) per shape and occupancy, timed with events."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wsl4mis_amd import _lib  # noqa: E402

if os.environ.get("WSL_TOOLS_EXP", "1") != "0":   # the experiments build carries the knobs / probes these tools drive
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import explib  # noqa: E402
    explib.use()

L = _lib.lib()
out = torch.zeros(4, device="cuda")
st = torch.cuda.current_stream().cuda_stream
FL = {0: 2 * 16 * 16 * 4, 1: 2 * 32 * 32 * 2, 2: 2 * 4 * 4 * 1 * 16}
NAME = {0: "16x16x4", 1: "32x32x2", 2: "4x4x1(16 blocks)"}
for shape in (0, 1, 2):
    for wg_per_cu in (1, 2, 3, 4):
        blocks, iters = 256 * wg_per_cu, 20000
        for _ in range(2):
            _lib.check(L.wsl_debug_mfma_stream(shape, blocks, iters, out.data_ptr(), st))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(L.wsl_debug_mfma_stream(shape, blocks, iters, out.data_ptr(), st))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        fl = blocks * 4 * iters * 16 * FL[shape]
        print(f"{NAME[shape]:18s} {wg_per_cu} waves/SIMD: {ms:7.2f} ms  {fl / ms / 1e9:7.1f} TFLOP/s")

# f32 VALU work next to the MFMA stream (16x16x4): K v_fma_f32 per MFMA in the same wave (100 + K) or in the SIMD's
# second wave (200 + K, 8-wave workgroups: waves 0-3 MFMA, waves 4-7 VALU).  MFMA TFLOP/s only (the FMAs are not counted).
for shape in (102, 104, 108, 116, 202, 204, 208, 216, 301, 302, 304, 308):
    blocks, iters = 256, 20000
    for _ in range(2):
        _lib.check(L.wsl_debug_mfma_stream(shape, blocks, iters, out.data_ptr(), st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(L.wsl_debug_mfma_stream(shape, blocks, iters, out.data_ptr(), st))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    fl = blocks * 4 * iters * 16 * FL[0]
    what = "v_pk_add_f32 (op_sel)" if shape >= 300 else "v_fma_f32"
    print(f"16x16x4 + {shape % 100:2d} {what} per MFMA in the {'partner' if 200 <= shape < 300 else 'same'} wave: {ms:7.2f} ms  "
          f"{fl / ms / 1e9:7.1f} TFLOP/s (MFMA flops)")
