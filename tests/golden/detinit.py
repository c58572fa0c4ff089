"""Deterministic, torch-free parameter initialiser shared by the golden-vector
generator (tests/golden/make_golden.py, runs only where /root/reference exists)
and by the parity tests (which rebuild the same parameters from the seed instead
of shipping multi-megabyte state_dicts in the fixtures).

Values depend only on (key name, shape, seed) through numpy's PCG64 `random()`
stream, which is stable across numpy versions.
"""
import zlib
import numpy as np


def _rng_for(name: str, seed: int) -> np.random.Generator:
    return np.random.default_rng([seed, zlib.crc32(name.encode())])


def det_tensor(name: str, shape, seed: int) -> np.ndarray:
    """Value for one state_dict entry, chosen by the suffix of its key."""
    rng = _rng_for(name, seed)
    shape = tuple(int(s) for s in shape)
    if name.endswith("num_batches_tracked"):
        return np.array(3, dtype=np.int64)
    u = rng.random(size=shape, dtype=np.float64)
    if name.endswith("running_mean"):
        return ((u - 0.5) * 0.2).astype(np.float32)
    if name.endswith("running_var"):
        return (0.5 + u).astype(np.float32)
    if len(shape) == 4:                       # conv weight, U(-b, b), b = 1/sqrt(fan_in)
        fan_in = shape[1] * shape[2] * shape[3]
        b = 1.0 / np.sqrt(fan_in)
        return ((2.0 * u - 1.0) * b * 1.7).astype(np.float32)
    # 1-D: conv bias / BN weight / BN bias. BN keys are '<blk>.1.*' / '<blk>.5.*'
    parts = name.split(".")
    is_bn = len(parts) >= 2 and parts[-2] in ("1", "5")
    if is_bn and name.endswith("weight"):
        return (0.5 + u).astype(np.float32)
    if is_bn and name.endswith("bias"):
        return ((u - 0.5) * 0.4).astype(np.float32)
    return ((u - 0.5) * 0.2).astype(np.float32)   # conv bias


def det_state(shapes: dict, seed: int) -> dict:
    """shapes: ordered {key: shape}. Returns {key: ndarray}."""
    return {k: det_tensor(k, s, seed) for k, s in shapes.items()}


def sample_index(numel: int, max_samples: int = 2048) -> np.ndarray:
    """Deterministic flat indices used to subsample large gradient tensors."""
    if numel <= max_samples:
        return np.arange(numel, dtype=np.int64)
    step = numel / float(max_samples)
    return np.unique((np.arange(max_samples) * step).astype(np.int64))


def scribble_labels(N, H, W, seed, n_classes=4, ignore=4):
    """Synthetic scribble annotation: everything `ignore` except one short
    random-walk polyline per class per slice (~1 % labelled), uint8 [N,H,W]."""
    rng = np.random.default_rng([seed, 77])
    lab = np.full((N, H, W), ignore, dtype=np.uint8)
    steps = max(4, int(0.0026 * H * W))
    for n in range(N):
        for c in range(n_classes):
            y, x = int(rng.integers(0, H)), int(rng.integers(0, W))
            for _ in range(steps):
                lab[n, y, x] = c
                d = int(rng.integers(0, 4))
                y = min(H - 1, max(0, y + (d == 0) - (d == 1)))
                x = min(W - 1, max(0, x + (d == 2) - (d == 3)))
    return lab
