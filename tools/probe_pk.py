#!/usr/bin/env python3
"""The packed Winograd transforms of wsl_rt.h (inline-assembly v_pk_add_f32 blocks with op_sel / neg_hi) on the GPU vs their
definitions: V = B^T d B of a 4 x 4 patch, Z = A dY A^T of a 2 x 2 tile with the sign folding of wgrad_wino_kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import explib  # noqa: E402

_lib = explib.use()
L = _lib.lib()
x = torch.randn(64, 20)
out = torch.zeros(64, 32, device="cuda")
_lib.check(L.wsl_debug_pk_probe(x.cuda().data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream))
o = out.cpu()
d = x[:, :16].view(64, 4, 4)
rt = torch.stack([d[:, 0] - d[:, 2], d[:, 1] + d[:, 2], d[:, 2] - d[:, 1], d[:, 1] - d[:, 3]], 1)          # rows
V = torch.stack([rt[:, :, 0] - rt[:, :, 2], rt[:, :, 1] + rt[:, :, 2], rt[:, :, 2] - rt[:, :, 1], rt[:, :, 1] - rt[:, :, 3]], 2)
r0, r1 = x[:, 16:18], x[:, 18:20]
q = torch.stack([r0, r0 + r1, r0 - r1, r1], 1)                                                               # [64, 4, 2]
Z = torch.stack([q[:, :, 0], q[:, :, 0] + q[:, :, 1], q[:, :, 0] - q[:, :, 1], q[:, :, 1]], 2)
print("V = B^T d B bit-exact:", bool(torch.equal(o[:, :16], V.reshape(64, 16))))
print("Z = A dY A^T (signs folded) bit-exact:", bool(torch.equal(o[:, 16:], Z.reshape(64, 16))))
