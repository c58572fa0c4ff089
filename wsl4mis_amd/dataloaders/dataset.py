"""Device-side mirror of the reference's per-slice augmentation.

Reference: code/dataloaders/dataset_semi.py:128-171 (`random_rot_flip`, `random_rotate`, `RandomGenerator`), which runs
numpy / scipy.ndimage per sample on the CPU inside DataLoader workers.  Here the random parameters are drawn on the host
in exactly the reference's order from the same generators (`random`, `numpy.random`), and the pixels are produced by ONE
gather kernel for a whole batch of slices of different native sizes (`wsl_augment_batch`, csrc/wsl_data.hip): every
step of the reference is a nearest-neighbour index map, so the composition is exact -- results equal numpy/scipy bit for
bit (tests/test_data.py).  No CPU fallback: without the HIP library these calls raise.

`BaseDataSets` mirrors the fold-aware dataset of `dataset_semi.py:17-127` (same constructor, same file selection, same
sample dicts) on top of `h5lite` (h5py is not in the image); `TwoStreamBatchSampler` the labelled/unlabelled batch
composer of `dataset_semi.py:174-229`."""
import os
import random

import numpy as np
import torch
from scipy import special

from torch.utils.data import Dataset
from torch.utils.data.sampler import Sampler

from .. import _lib
from .. import runtime as rt
from . import h5lite


class BaseDataSets(Dataset):
    """ACDC slices (train) / volumes (val) of one cross-validation fold (ref: dataset_semi.py:17-127).

    Folds are blocks of 20 patients (fold k tests patients 20(k-1)+1 .. 20k); the "labeled" patients are
    patient010, 020, ..., 100 among the fold's training patients, the rest are "unlabeled"; train samples are the slice
    files whose name starts with a selected patient id, read as {'image', 'label' = file[sup_type]} and passed through
    `transform`; val samples are whole volumes {'image', 'label'} as numpy arrays.  'idx' carries the patient id."""

    FOLDS = ("fold1", "fold2", "fold3", "fold4", "fold5")

    def __init__(self, base_dir=None, num=4, labeled_type="labeled", split="train", transform=None, fold="fold1",
                 sup_type="label", cache=False):
        if fold not in self.FOLDS:
            raise ValueError(f"unknown fold {fold!r} (the reference returns 'ERROR KEY' here and fails later)")
        self._base_dir, self.split, self.sup_type, self.transform = base_dir, split, sup_type, transform
        self.num, self.labeled_type = num, labeled_type
        # cache=True (not in the reference): decoded arrays are kept in host memory after the first read -- h5lite inflates in
        # Python, and a 60000-iteration run touches each of the ~1500 slice files ~500 times
        self._cache = {} if cache else None
        k = self.FOLDS.index(fold)
        test_ids = ["patient{:0>3}".format(i) for i in range(20 * k + 1, 20 * k + 21)]
        train_ids = [p for p in ("patient{:0>3}".format(i) for i in range(1, 101)) if p not in test_ids]
        labeled = [p for p in ("patient{:0>3}".format(10 * i) for i in range(1, 11)) if p in train_ids]
        if split == "train":
            files = os.listdir(os.path.join(base_dir, "ACDC_training_slices"))
            # labeled_type: 'labeled' | anything else = the unlabeled remainder, like the reference; 'all' (extension) = every
            # training patient of the fold -- what the weakly-supervised scripts of upstream WSL4MIS train on (every slice
            # has a scribble), which the fold-aware class shipped in this reference cannot express
            chosen = labeled if labeled_type == "labeled" else train_ids if labeled_type == "all" else \
                [p for p in train_ids if p not in labeled]
        elif split == "val":
            files = os.listdir(os.path.join(base_dir, "ACDC_training_volumes"))
            chosen = test_ids
        else:
            files, chosen = [], []
        self.sample_list = [f for pid in chosen for f in files if f.startswith(pid)]     # per patient, listing order

    def __len__(self):
        return len(self.sample_list)

    def __getitem__(self, idx):
        case = self.sample_list[idx]
        if self._cache is not None and case in self._cache:
            image, label = self._cache[case]
        else:
            sub = "ACDC_training_slices" if self.split == "train" else "ACDC_training_volumes"
            with h5lite.File(os.path.join(self._base_dir, sub, case)) as f:
                image = f["image"][:]
                label = f[self.sup_type if self.split == "train" else "label"][:]
            if self._cache is not None:
                self._cache[case] = (image, label)
        sample = {"image": image, "label": label}
        if self.split == "train" and self.transform is not None:
            sample = self.transform(sample)
        sample["idx"] = case.split("_")[0]
        sample["case"] = case                                  # the file name ...
        # ... and WHICH dataset it came from: what BatchRandomGenerator(device_cache=True) keys on.  Two datasets that share file names
        # but not content (another sup_type, fold or root) must not alias in one generator's device cache.
        sample["source"] = (os.path.abspath(self._base_dir), self.split, self.sup_type if self.split == "train" else "label")
        return sample


def draw_params(label_np, has_ignore=None):
    """The reference's random decisions for ONE sample, same draw order and generators (dataset_semi.py:155-161).
    has_ignore: precomputed `4 in np.unique(label)` (then label_np is not read)."""
    p = {"op": 0}
    if random.random() > 0.5:
        p["op"], p["k"] = 1, int(np.random.randint(0, 4))
        p["axis"] = int(np.random.randint(0, 2))
    elif random.random() > 0.5:
        p["op"], p["angle"] = 2, int(np.random.randint(-20, 20))
        p["lab_cval"] = 4 if (has_ignore if has_ignore is not None else 4 in np.unique(label_np)) else 0
    return p


def _rotate_matrix(angle, shape):
    """scipy.ndimage.rotate(reshape=False) for a 2-D array: matrix from cosdg/sindg, offset about the centres (n-1)/2."""
    c, s = special.cosdg(angle), special.sindg(angle)
    m = np.array([[c, s], [-s, c]], dtype=np.float64)
    ctr = (np.array(shape, dtype=np.float64) - 1) / 2
    off = ctr - m @ ctr
    return m, off


def augment_batch(images, labels, params, output_size):
    """images: list of [h,w] float32 tensors, labels: list of [h,w] uint8 tensors (any device; moved to the GPU),
    params: list of dicts from draw_params.  Returns (image [N,1,Ho,Wo] float32, label [N,Ho,Wo] uint8) on the device."""
    L = _lib.lib()
    dev = rt.device()
    n = len(images)
    Ho, Wo = int(output_size[0]), int(output_size[1])
    keep = []
    arr = (_lib.WslAugSample * n)()
    for i, (im, lb, p) in enumerate(zip(images, labels, params)):
        im = torch.as_tensor(im, dtype=torch.float32).to(dev).contiguous()
        lb = torch.as_tensor(lb, dtype=torch.uint8).to(dev).contiguous()
        if im.dim() != 2 or im.shape != lb.shape:
            raise ValueError(f"augment_batch: sample {i}: image {tuple(im.shape)} / label {tuple(lb.shape)} must be equal 2-D shapes")
        keep += [im, lb]
        s = arr[i]
        s.img, s.lab, s.h, s.w = rt.ptr(im), rt.ptr(lb), im.shape[0], im.shape[1]
        s.op, s.k, s.axis = p["op"], p.get("k", 0), p.get("axis", 0)
        s.lab_cval, s.img_cval = p.get("lab_cval", 0), 0.0
        if p["op"] == 2:
            m, off = _rotate_matrix(p["angle"], im.shape)
            s.m00, s.m01, s.m10, s.m11, s.off0, s.off1 = m[0, 0], m[0, 1], m[1, 0], m[1, 1], off[0], off[1]
    out_img = torch.empty((n, 1, Ho, Wo), dtype=torch.float32, device=dev)
    out_lab = torch.empty((n, Ho, Wo), dtype=torch.uint8, device=dev)
    _lib.check(L.wsl_augment_batch(arr, n, rt.ptr(out_img), rt.ptr(out_lab), Ho, Wo, rt.stream()))
    # no host sync: the descriptors were copied into the launch's kernel arguments, and the staged inputs are released through
    # torch's stream-ordered allocator on this same stream
    return out_img, out_lab


class RandomGenerator(object):
    """Same call contract as the reference class: sample dict {'image': [h,w], 'label': [h,w]} -> {'image': [1,H,W]
    float32, 'label': [H,W] uint8}, with the tensors living on the device."""

    def __init__(self, output_size):
        self.output_size = output_size

    def __call__(self, sample):
        image, label = np.asarray(sample["image"]), np.asarray(sample["label"])
        img, lab = augment_batch([image], [label], [draw_params(label)], self.output_size)
        return {"image": img[0], "label": lab[0]}


class BatchRandomGenerator(object):
    """The batched form the engine wants: a list of samples in, one device batch out (one kernel launch)."""

    def __init__(self, output_size, device_cache=False, max_cached=16384):
        """device_cache: keep each distinct source slice on the device after its first use, so a step stages nothing over PCIe.
        Keyed by the sample's ('source', 'case') (dataset identity -- root, split, label kind -- and file name, which BaseDataSets puts
        into every sample); samples without a 'case' are keyed by the
        identity of their arrays, which only a caching dataset keeps stable -- a source that hands out fresh arrays on every
        access would grow the cache without bound, so more than `max_cached` entries raise instead."""
        self.output_size = output_size
        self._dev = {} if device_cache else None
        self._max = max_cached

    def _staged(self, s):
        case = s.get("case") if isinstance(s, dict) else None
        key = ("case", s.get("source"), case) if case is not None else (id(s["image"]), id(s["label"]))
        hit = self._dev.get(key)
        if hit is None:
            if len(self._dev) >= self._max:
                raise _lib.WslError(f"BatchRandomGenerator(device_cache=True): {len(self._dev)} distinct sources staged -- the dataset "
                                    "hands out new arrays on every access (use BaseDataSets(cache=True) or samples with a 'case' key), "
                                    "or raise max_cached")
            lab = np.asarray(s["label"])
            hit = (torch.as_tensor(np.asarray(s["image"]), dtype=torch.float32).to(rt.device()).contiguous(),
                   torch.as_tensor(lab, dtype=torch.uint8).to(rt.device()).contiguous(), bool(4 in np.unique(lab)),
                   s["image"], s["label"])                    # (the arrays themselves: keeps identity keys alive)
            self._dev[key] = hit
        return hit

    def __call__(self, samples):
        if self._dev is None:
            params = [draw_params(np.asarray(s["label"])) for s in samples]     # the reference's per-sample draw order
            return augment_batch([s["image"] for s in samples], [s["label"] for s in samples], params, self.output_size)
        st = [self._staged(s) for s in samples]
        params = [draw_params(None, has_ignore=t[2]) for t in st]
        return augment_batch([t[0] for t in st], [t[1] for t in st], params, self.output_size)


class TwoStreamBatchSampler(Sampler):
    """Batches of `batch_size - secondary_batch_size` primary indices followed by `secondary_batch_size` secondary ones
    (ref: dataset_semi.py:174-229).  One epoch = one random pass over the primary indices (a short tail is dropped);
    the secondary indices are drawn from an endless sequence of fresh permutations.  Uses `numpy.random` like the
    reference: one permutation of the primary set when the iterator is created, then one of the secondary set whenever
    the previous one is exhausted."""

    def __init__(self, primary_indices, secondary_indices, batch_size, secondary_batch_size):
        self.primary_indices, self.secondary_indices = primary_indices, secondary_indices
        self.secondary_batch_size = secondary_batch_size
        self.primary_batch_size = batch_size - secondary_batch_size
        assert len(self.primary_indices) >= self.primary_batch_size > 0
        assert len(self.secondary_indices) >= self.secondary_batch_size > 0

    def __iter__(self):
        primary = np.random.permutation(self.primary_indices)
        pool = []
        for b in range(len(self)):
            head = tuple(primary[b * self.primary_batch_size:(b + 1) * self.primary_batch_size])
            while len(pool) < self.secondary_batch_size:
                pool.extend(np.random.permutation(self.secondary_indices))
            tail, pool = tuple(pool[:self.secondary_batch_size]), pool[self.secondary_batch_size:]
            yield head + tail

    def __len__(self):
        return len(self.primary_indices) // self.primary_batch_size
