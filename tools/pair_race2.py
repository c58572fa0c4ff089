#!/usr/bin/env python3
"""pair_race.py with a SYNTHETIC aggressor: which form of v_mfma_f32_16x16x32_f16 stream (tools/aggr_mfma.hip) disturbs a co-resident
f32 MFMA kernel of the product library?  (profiles/r4_sp_root_cause.md; GPU only.)
   python tools/pair_race2.py VICTIM [reps]        VICTIM as in pair_race.py, e.g. c1:64,32,32,128,64"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pair_race import Case, dev  # noqa: E402

A = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "exp", "libaggr_mfma.so"))
victim = Case(sys.argv[1], 5)
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
sink = torch.zeros(4, device=dev)
victim.launch(s1)
torch.cuda.synchronize()
ref = victim.y.clone()
rms = float(ref.pow(2).mean().sqrt())
names = sys.argv[3].split(",") if len(sys.argv) > 3 else ("inplace", "inplace_chain", "renamed", "renamed_nops", "dst_on_a", "dst_on_b", "inplace_lds", "renamed_lds", "f32mfma", "valu", "sparse")
blocks = int(sys.argv[4]) if len(sys.argv) > 4 else 512
for name in names:
    fn = getattr(A, "launch_aggr_" + name)
    fn.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    bad, worst = 0, 0.0
    for r in range(reps):
        with torch.cuda.stream(s1):
            victim.y.zero_()
        rc = fn(blocks, 1500, sink.data_ptr(), s2.cuda_stream)
        assert rc == 0, rc
        victim.launch(s1)
        victim.launch(s1)
        torch.cuda.synchronize()
        if not torch.equal(victim.y, ref):
            bad += 1
            worst = max(worst, float((victim.y - ref).abs().max()))
    print(f"victim {sys.argv[1]} | aggressor {name:14s} x{blocks}: {bad}/{reps} wrong (worst |delta| {worst:.3e} = {worst / rms:.2e} RMS)", flush=True)
