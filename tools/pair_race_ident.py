#!/usr/bin/env python3
"""What does the corrupted f32 1x1 convolution actually compute?  (profiles/r4_sp_root_cause.md; GPU only.)
Victim = the product's 1x1 conv (BatchNorm source) with IDENTITY weights, so output element (n, c, y, x) IS the staged input
leaky(x * scale[c] + shift[c]); aggressor = the synthetic f16-MFMA kernel on a second stream.  For a wrong launch, every wrong element is
matched against candidates: the raw input, the transform with another channel's coefficients, zero, a neighbouring pixel."""
import ctypes as C
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pair_race import Case, dev, L, _lib  # noqa: E402

A = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "exp", "libaggr_mfma.so"))
Cn = int(sys.argv[1]) if len(sys.argv) > 1 else 64
v = Case(f"c1:64,32,32,{Cn},{Cn}", 5)
v.w = torch.eye(Cn).view(Cn, Cn, 1, 1).contiguous().to(dev)
st = torch.cuda.current_stream().cuda_stream
_lib.check(L.wsl_conv2d_pack_weights(v.w.data_ptr(), v.img.data_ptr(), Cn, Cn, 1, 0, st))
torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
sink = torch.zeros(4, device=dev)
v.launch(s1)
torch.cuda.synchronize()
ref = v.y.clone()
want = F.leaky_relu(v.x * v.scale[None, :, None, None] + v.shift[None, :, None, None], 0.01)
print("quiet run vs the transform computed by torch: max |delta|", float((ref - want).abs().max()))
fn = A.launch_aggr_inplace
fn.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
for rep in range(6):
    with torch.cuda.stream(s1):
        v.y.zero_()
    fn(512, 1500, sink.data_ptr(), s2.cuda_stream)
    v.launch(s1)
    torch.cuda.synchronize()
    bad = (v.y != ref)
    if not bad.any():
        print(f"rep {rep}: right")
        continue
    idx = bad.nonzero()
    n_, c_, y_, x_ = idx.unbind(1)
    got, exp = v.y[bad], ref[bad]
    print(f"rep {rep}: {idx.shape[0]} wrong elements; samples {torch.unique(n_).numel()}, channels {torch.unique(c_).tolist()[:32]}, "
          f"rows {torch.unique(y_).tolist()[:16]}, cols {torch.unique(x_).tolist()[:16]}")
    raw = v.x[bad]
    print("   == 0:", int((got == 0).sum()), " == raw x:", int((got == raw).sum()), " == leaky(raw):", int((got == F.leaky_relu(raw, 0.01)).sum()))
    # another channel's coefficients?
    best = torch.zeros_like(got, dtype=torch.int64) - 1
    for c2 in range(Cn):
        cand = F.leaky_relu(raw * v.scale[c2] + v.shift[c2], 0.01)
        best[(cand == got) & (best < 0)] = c2
    hit = best >= 0
    print("   == transform with ANOTHER channel's coefficients:", int(hit.sum()), "of", got.numel())
    if hit.any():
        d = (best[hit] - c_[hit])
        vals, cnts = torch.unique(d, return_counts=True)
        print("      channel offset (used - own): count", list(zip(vals.tolist(), cnts.tolist()))[:16])
    # the right value of another element of the same sample / channel plane?
    for name, dy, dx in (("pixel x+4", 0, 4), ("pixel x-4", 0, -4), ("row y+1", 1, 0), ("row y-1", -1, 0)):
        yy, xx = (y_ + dy).clamp(0, 31), (x_ + dx).clamp(0, 31)
        print(f"   == right value of {name}:", int((got == ref[n_, c_, yy, xx]).sum()))
    for name, dc in (("channel c+1", 1), ("channel c-1", -1), ("channel c+8", 8), ("channel c-8", -8), ("channel c+4", 4), ("channel c-4", -4)):
        cc = (c_ + dc).clamp(0, Cn - 1)
        print(f"   == right value of {name} at the same pixel:", int((got == ref[n_, cc, y_, x_]).sum()))
    k = min(8, got.numel())
    print("   first wrong elements (n, c, y, x): got / expected / raw")
    for i in range(k):
        print("     ", idx[i].tolist(), float(got[i]), float(exp[i]), float(raw[i]))
    break
