#!/usr/bin/env python3
"""Where does global_load_lds_dwordx4 put each lane's 16 bytes?  Expected: M0 base + lane * 16 (lane-contiguous)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wsl4mis_amd import _lib  # noqa: E402

L = _lib.lib()
g = torch.arange(1024, dtype=torch.float32, device="cuda")
out = torch.zeros(2048, device="cuda")
_lib.check(L.wsl_debug_lds_dma_probe(g.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
o = out.cpu()
exp = torch.full((2048,), -1.0)
for w in range(4):
    for l in range(64):
        if l == 5:
            continue
        t = w * 64 + l
        exp[w * 256 + 8 + l * 4: w * 256 + 8 + l * 4 + 4] = torch.arange(t * 4, t * 4 + 4, dtype=torch.float32)
print("lane-contiguous destination (base + lane*16 B), masked lane untouched:", bool(torch.equal(o, exp)))
if not torch.equal(o, exp):
    bad = (o != exp).nonzero().flatten()[:16].tolist()
    print("first mismatches:", [(i, float(o[i]), float(exp[i])) for i in bad])
