"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the metrics the reference's validation takes
from medpy (code/val_2D.py:3,7-15: `metric.binary.dc`, `metric.binary.hd95`).  medpy is a third-party dependency that is
absent from /root/reference and from this image (unpinned there; the algorithm below is medpy 0.4.0
`medpy/metric/binary.py`: dc, hd95, __surface_distances), restated with the scipy calls medpy itself makes."""
import numpy as np
from scipy.ndimage import binary_erosion, distance_transform_edt, generate_binary_structure


def dc(result, reference):
    result, reference = np.atleast_1d(result.astype(bool)), np.atleast_1d(reference.astype(bool))
    inter = np.count_nonzero(result & reference)
    s = np.count_nonzero(result) + np.count_nonzero(reference)
    try:
        return 2.0 * inter / float(s)
    except ZeroDivisionError:
        return 0.0


def _surface_distances(result, reference, voxelspacing=None, connectivity=1):
    result, reference = np.atleast_1d(result.astype(bool)), np.atleast_1d(reference.astype(bool))
    footprint = generate_binary_structure(result.ndim, connectivity)
    if 0 == np.count_nonzero(result):
        raise RuntimeError("The first supplied array does not contain any binary object.")
    if 0 == np.count_nonzero(reference):
        raise RuntimeError("The second supplied array does not contain any binary object.")
    result_border = result ^ binary_erosion(result, structure=footprint, iterations=1)
    reference_border = reference ^ binary_erosion(reference, structure=footprint, iterations=1)
    dt = distance_transform_edt(~reference_border, sampling=voxelspacing)
    return dt[result_border]


def hd95(result, reference):
    hd1 = _surface_distances(result, reference)
    hd2 = _surface_distances(reference, result)
    return np.percentile(np.hstack((hd1, hd2)), 95)


def calculate_metric_percase(pred, gt):        # code/val_2D.py:7-15
    pred, gt = (np.asarray(pred) > 0).astype(np.uint8), (np.asarray(gt) > 0).astype(np.uint8)
    if pred.sum() > 0:
        return dc(pred, gt), hd95(pred, gt)
    return 0, 0
