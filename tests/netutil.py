"""Helpers shared by the network-level tests: arenas from the C ABI's entry table."""
import ctypes as C

import numpy as np

from detinit import det_state, sample_index
from wsl4mis_amd import _lib


def net_desc(net, N, H, W, in_chns=1, n_class=4, precision=0):
    """precision 1: the eligible 3x3 layers on the split-precision path (f16 hi / lo operands, three MFMA passes)"""
    return _lib.WslNetDesc(in_chns, n_class, 2 if net == "unet_cct" else 1, N, H, W, precision, 0)


def entries(lib, d):
    out = []
    for i in range(lib.wsl_net_num_entries(C.byref(d))):
        e = _lib.WslNetEntry()
        assert lib.wsl_net_entry(C.byref(d), i, C.byref(e)) == 0
        out.append((e.name.decode(), e.kind, tuple(e.shape[k] for k in range(e.ndim)), e.offset))
    return out


def det_arenas(lib, d, seed):
    """(params, buffers, nbt, entries) filled with the deterministic state used by the golden generator."""
    ents = entries(lib, d)
    vals = det_state({n: s for n, _, s, _ in ents}, seed)
    params = np.zeros(lib.wsl_net_param_count(C.byref(d)), np.float32)
    bufs = np.zeros(lib.wsl_net_buffer_count(C.byref(d)), np.float32)
    nbt = np.zeros(sum(1 for e in ents if e[1] == 2), np.int64)
    for n, kind, shape, off in ents:
        v = np.asarray(vals[n])
        if kind == 0:
            params[off:off + v.size] = v.ravel()
        elif kind == 1:
            bufs[off:off + v.size] = v.ravel()
        else:
            nbt[off] = int(v)
    return params, bufs, nbt, ents


def ptr_array(be, arrs):
    """const T* const* from a list of device arrays (or None)."""
    if arrs is None:
        return None
    a = (C.c_void_p * len(arrs))(*[be.ptr(x) if x is not None else None for x in arrs])
    return a


def check_grads(g, grads, ents, grad_tol, prefix="g."):
    bad = []
    for n, kind, shape, off in ents:
        if kind != 0:
            continue
        size = int(np.prod(shape)) if shape else 1
        got = grads[off:off + size]
        ref = g[f"{prefix}{n}"]
        err = float(np.max(np.abs(got[sample_index(size)] - ref)))
        if err > grad_tol(n, ref):
            bad.append((n, err, float(np.max(np.abs(ref)))))
        nrm = g[f"{prefix.replace('g.', 'gn.')}{n}"] if f"{prefix.replace('g.', 'gn.')}{n}" in g else None
        if nrm is not None and nrm[0] > 1e-4:
            got_n = np.sqrt((got.astype(np.float64) ** 2).sum())
            if abs(got_n - nrm[0]) > 2e-4 * nrm[0] + 1e-6:
                bad.append((n + " (norm)", got_n, nrm[0]))
    return bad


def grad_l2(g, grads, ents, prefix="g."):
    """Relative L2 error over the sampled entries of every parameter-gradient tensor."""
    num = den = 0.0
    for n, kind, shape, off in ents:
        if kind != 0:
            continue
        size = int(np.prod(shape)) if shape else 1
        got = grads[off:off + size][sample_index(size)].astype(np.float64)
        ref = g[f"{prefix}{n}"].astype(np.float64)
        num += float(((got - ref) ** 2).sum())
        den += float((ref ** 2).sum())
    return (num / den) ** 0.5


def strict_grads(g):
    """True when the fixture's forward stays clear of gradient discontinuities (LeakyReLU kink, max-pool ties) by more
    than cross-implementation fp32 noise -- see KinkMargins in tests/golden/make_golden.py."""
    m = g["margins"]
    return bool(m[0] >= 4e-6 and m[1] >= 1e-6)


class KinkMargins:
    """Context manager around ORACLE forwards: smallest |BatchNorm output| (distance of a pre-activation from the LeakyReLU
    kink) and smallest gap between the two largest values of a 2x2 pooling window.  Element-wise gradient parity between
    two fp32 implementations is only defined when both are clear of fp32 noise."""

    def __enter__(self):
        import torch
        import torch.nn.functional as F
        self._F, self._bn, self._mp = F, F.batch_norm, F.max_pool2d
        self.leaky, self.pool = float("inf"), float("inf")

        def bn(y, *a, **kw):
            z = self._bn(y, *a, **kw)
            self.leaky = min(self.leaky, float(z.abs().min()))
            return z

        def mp(x, k, *a, **kw):
            if k == 2:
                N, C, H, W = x.shape
                v = x.reshape(N, C, H // 2, 2, W // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(N, C, H // 2, W // 2, 4)
                s = torch.sort(v, dim=-1, descending=True)[0]
                self.pool = min(self.pool, float((s[..., 0] - s[..., 1]).min()))
            return self._mp(x, k, *a, **kw)
        F.batch_norm, F.max_pool2d = bn, mp
        return self

    def __exit__(self, *exc):
        self._F.batch_norm, self._F.max_pool2d = self._bn, self._mp


class DecisionReplay:
    """Context manager around ORACLE forwards (oracle.torch_ref calls F.leaky_relu once per BatchNorm layer and F.max_pool2d(x, 2)
    once per encoder level, in state_dict order).  record=True: keeps the oracle's own decisions (`signs`: bool [N,C,H,W] per
    BatchNorm layer, `args`: uint8 [N,C,H/2,W/2] first-maximum position per pooling).  Otherwise the given decisions REPLACE the
    oracle's: LeakyReLU becomes  where(sign, z, 0.01 z)  and the pooling a gather -- the forward then is the piecewise-linear
    function another implementation evaluated, and its autograd gradient is what that implementation's backward must reproduce
    element by element."""

    def __init__(self, signs=None, args=None, record=False):
        self.signs, self.args, self.record = (signs or []), (args or []), record
        self._i = self._j = 0

    @staticmethod
    def _windows(x):
        N, C, H, W = x.shape
        return x.reshape(N, C, H // 2, 2, W // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(N, C, H // 2, W // 2, 4)

    def __enter__(self):
        import torch
        import torch.nn.functional as F
        self._F, self._lr, self._mp = F, F.leaky_relu, F.max_pool2d

        def lr(z, slope=0.01, *a, **kw):
            if self.record:
                self.signs.append(z > 0)
                return self._lr(z, slope, *a, **kw)
            m = self.signs[self._i]
            self._i += 1
            return torch.where(m.to(torch.bool), z, z * slope)

        def mp(x, k, *a, **kw):
            if k != 2:
                return self._mp(x, k, *a, **kw)
            v = self._windows(x)
            if self.record:
                self.args.append(torch.argmax(v, dim=-1).to(torch.uint8))     # first index on ties, like the strict '>' scan
                return self._mp(x, k, *a, **kw)
            idx = self.args[self._j]
            self._j += 1
            return torch.gather(v, -1, idx.long().unsqueeze(-1)).squeeze(-1)
        F.leaky_relu, F.max_pool2d = lr, mp
        return self

    def __exit__(self, *exc):
        self._F.leaky_relu, self._F.max_pool2d = self._lr, self._mp
        return False
