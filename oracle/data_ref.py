"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the reference's per-slice augmentation with the
same third-party calls the reference makes (numpy, scipy.ndimage).

Follows code/dataloaders/dataset_semi.py:128-135 (random_rot_flip), :138-143 (random_rotate), :146-171
(RandomGenerator.__call__), with the random decisions passed in explicitly (the product draws them with
wsl4mis_amd.dataloaders.dataset.draw_params in the reference's order)."""
import numpy as np
from scipy import ndimage
from scipy.ndimage import zoom


def apply(image, label, p, output_size):
    if p["op"] == 1:                                         # dataset_semi.py:128-135
        image, label = np.rot90(image, p["k"]), np.rot90(label, p["k"])
        image, label = np.flip(image, axis=p["axis"]).copy(), np.flip(label, axis=p["axis"]).copy()
    elif p["op"] == 2:                                       # dataset_semi.py:138-143
        image = ndimage.rotate(image, p["angle"], order=0, reshape=False)
        label = ndimage.rotate(label, p["angle"], order=0, reshape=False, mode="constant", cval=p["lab_cval"])
    x, y = image.shape                                       # dataset_semi.py:162-169
    image = zoom(image, (output_size[0] / x, output_size[1] / y), order=0)
    label = zoom(label, (output_size[0] / x, output_size[1] / y), order=0)
    return image.astype(np.float32)[None], label.astype(np.uint8)


def random_generator(sample, output_size, rng_random, rng_numpy):
    """The reference's full __call__ with explicit generators (python `random`-like and numpy RandomState-like)."""
    image, label = sample["image"], sample["label"]
    p = {"op": 0}
    if rng_random.random() > 0.5:
        p = {"op": 1, "k": int(rng_numpy.randint(0, 4))}
        p["axis"] = int(rng_numpy.randint(0, 2))
    elif rng_random.random() > 0.5:
        p = {"op": 2, "angle": int(rng_numpy.randint(-20, 20)), "lab_cval": 4 if 4 in np.unique(label) else 0}
    return apply(image, label, p, output_size), p


def two_stream_batches(primary, secondary, batch_size, secondary_batch_size, rng_numpy):
    """dataset_semi.py:174-229 restated: zip of fixed-length groups of one primary permutation and of the endless
    chain of secondary permutations (generator semantics: the next secondary permutation is only drawn when needed)."""
    import itertools
    pb = batch_size - secondary_batch_size
    prim = iter(rng_numpy.permutation(primary))

    def forever():
        while True:
            yield rng_numpy.permutation(secondary)
    sec = itertools.chain.from_iterable(forever())
    return [a + b for a, b in zip(zip(*[prim] * pb), zip(*[sec] * secondary_batch_size))]
