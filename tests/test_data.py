"""Data path: the batched device augmentation equals the reference's numpy/scipy pipeline bit for bit."""
import random

import numpy as np
import pytest
import torch

from conftest import get_backend
from oracle import data_ref, metrics_ref


@pytest.fixture(params=[pytest.param("emul"), pytest.param("hip", marks=pytest.mark.gpu)])
def mode(request):
    from wsl4mis_amd import _lib, runtime
    _lib._reset_for_tests()
    runtime._ws_cache.clear()
    if request.param == "emul":
        _lib.use_library_for_tests(get_backend("emul").lib)
    yield request.param
    _lib._reset_for_tests()
    runtime._ws_cache.clear()


def make_samples(rng, n, sizes=None):
    out = []
    for i in range(n):
        h, w = sizes[i] if sizes else (int(rng.integers(40, 120)), int(rng.integers(40, 120)))
        img = rng.random((h, w), dtype=np.float32)
        lab = np.full((h, w), 4, np.uint8)
        lab[rng.random((h, w)) < 0.05] = rng.integers(0, 4)
        if i % 3 == 2:
            lab = rng.integers(0, 4, (h, w)).astype(np.uint8)          # dense label: cval 0 branch
        out.append({"image": img, "label": lab})
    return out


def test_every_op_matches_numpy_scipy(mode):
    from wsl4mis_amd.dataloaders import dataset
    rng = np.random.default_rng(3)
    samples = make_samples(rng, 14, sizes=[(64, 64), (50, 70), (71, 45), (97, 33)] * 3 + [(256, 216), (34, 80)])
    params = [{"op": 0}] + [{"op": 1, "k": k, "axis": a} for k in range(4) for a in range(2)] + \
             [{"op": 2, "angle": a, "lab_cval": c} for a, c in ((-20, 4), (-7, 0), (0, 4), (13, 4), (19, 0))]
    img, lab = dataset.augment_batch([s["image"] for s in samples], [s["label"] for s in samples], params, (48, 56))
    for i, (s, p) in enumerate(zip(samples, params)):
        ri, rl = data_ref.apply(s["image"], s["label"], p, (48, 56))
        assert np.array_equal(img[i].cpu().numpy(), ri), (i, p)
        assert np.array_equal(lab[i].cpu().numpy(), rl), (i, p)


def test_random_generator_draw_order_and_pixels(mode):
    """same seeds -> same decisions and the same pixels as the reference's per-sample loop"""
    from wsl4mis_amd.dataloaders import dataset
    rng = np.random.default_rng(11)
    samples = make_samples(rng, 24)
    random.seed(5), np.random.seed(6)
    gen = dataset.BatchRandomGenerator((64, 64))
    img, lab = gen(samples)
    r1, r2 = random.Random(5), np.random.RandomState(6)
    ops = set()
    for i, s in enumerate(samples):
        (ri, rl), p = data_ref.random_generator(s, (64, 64), r1, r2)
        ops.add(p["op"])
        assert np.array_equal(img[i].cpu().numpy(), ri) and np.array_equal(lab[i].cpu().numpy(), rl), (i, p)
    assert ops == {0, 1, 2}
    one = dataset.RandomGenerator((32, 32))({"image": samples[0]["image"], "label": samples[0]["label"]})
    assert tuple(one["image"].shape) == (1, 32, 32) and one["label"].dtype == torch.uint8


def test_bad_input_raises(mode):
    from wsl4mis_amd.dataloaders import dataset
    with pytest.raises(ValueError):
        dataset.augment_batch([np.zeros((4, 5), np.float32)], [np.zeros((5, 4), np.uint8)], [{"op": 0}], (8, 8))


def blobs(rng, shape, k):
    z, y, x = np.meshgrid(*[np.arange(n) for n in shape], indexing="ij")
    v = np.zeros(shape, bool)
    for _ in range(k):
        c = [rng.uniform(0, n) for n in shape]
        r = rng.uniform(2, 0.35 * min(shape[1:]))
        v |= ((z - c[0]) * 2.5) ** 2 + (y - c[1]) ** 2 + (x - c[2]) ** 2 < r * r
    return v


def test_validation_metrics_match_the_medpy_algorithm(mode):
    """Dice and HD95 of the validation loop (val_2D.py:7-15) against the scipy restatement of medpy's algorithm"""
    from wsl4mis_amd import val_2D
    rng = np.random.default_rng(17)
    checked = 0
    for shape in ((6, 40, 48), (9, 64, 56), (1, 33, 47), (4, 30, 30)):
        gt, pred = blobs(rng, shape, 3), blobs(rng, shape, 3)
        if shape[0] == 4:
            pred = np.ones(shape, bool)                           # object touching every array face
        if not (gt.any() and pred.any()):
            continue
        d_ref, h_ref = metrics_ref.calculate_metric_percase(pred, gt)
        d, h = val_2D.metric_percase(pred, gt)
        assert d == d_ref
        assert abs(h - h_ref) <= 1e-12 * max(1.0, h_ref), (shape, h, h_ref)
        checked += 1
    assert checked >= 3
    assert val_2D.metric_percase(np.zeros((3, 8, 8), bool), np.ones((3, 8, 8), bool)) == (0, 0)
    with pytest.raises(RuntimeError, match="second supplied array"):
        val_2D.metric_percase(np.ones((2, 8, 8), bool), np.zeros((2, 8, 8), bool))
