"""Synthetic 256x256 scribble-supervised slices for benchmarks: image U[0,1) fp32 [N,1,H,W]; label uint8
[N,H,W] filled with the ignore index 4 except one random-walk scribble per class, sized so that about 1 % of the
pixels are labelled (the share measured on the shipped ACDC scribbles)."""
import numpy as np
import torch


def scribble_labels(N, H, W, seed, n_classes=4, ignore=4, share=0.0106):
    rng = np.random.default_rng([seed, 77])
    lab = np.full((N, H, W), ignore, dtype=np.uint8)
    steps = max(4, int(share * H * W / n_classes * 1.6))      # walks revisit pixels: ~60 % of steps are new
    for n in range(N):
        for c in range(n_classes):
            y, x = int(rng.integers(0, H)), int(rng.integers(0, W))
            d = rng.integers(0, 4, size=steps)
            for k in range(steps):
                lab[n, y, x] = c
                y = min(H - 1, max(0, y + int(d[k] == 0) - int(d[k] == 1)))
                x = min(W - 1, max(0, x + int(d[k] == 2) - int(d[k] == 3)))
    return lab


def batch(N, H, W, seed, device):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand((N, 1, H, W), generator=g, dtype=torch.float32)
    lab = torch.from_numpy(scribble_labels(N, H, W, seed))
    return x.to(device), lab.to(device)
