"""In-training validation of the reference (ref: code/val_2D.py:18-50 test_single_volume, :90-124
test_single_volume_cct): per-slice zoom to the patch size -> net in eval mode -> argmax over softmax -> zoom back, then
per-class metrics over the volume.  The forward runs on the HIP path; the zoom (scipy, order 0, like the reference) is
host-side glue.  medpy is not available offline: Dice is computed here, and HD95 follows medpy 0.4.0
`metric.binary.hd95` (surface = object minus its 6-neighbourhood erosion; both directed sets of nearest-surface
distances; 95th percentile with numpy's linear rule) with the two heavy pieces on the device (`wsl_surface_u8`,
`wsl_nearest_dist2`: exact integer squared distances) and `torch.nonzero` / `torch.sort` as plumbing."""
import numpy as np
import torch
from scipy.ndimage import zoom

from . import runtime as rt


def dice_percase(pred, gt):
    """binary Dice of one class, medpy.metric.binary.dc semantics (ref: val_2D.py:7-15: 0 when nothing is predicted)."""
    pred, gt = pred.astype(bool), gt.astype(bool)
    if pred.sum() == 0:
        return 0.0
    inter = np.count_nonzero(pred & gt)
    denom = np.count_nonzero(pred) + np.count_nonzero(gt)
    return 2.0 * inter / denom if denom else 0.0


def _surface_points(vol_bool):
    v = torch.as_tensor(np.ascontiguousarray(vol_bool, dtype=np.uint8)).to(rt.device())
    if v.dim() not in (2, 3):
        raise NotImplementedError(f"hd95 of a {v.dim()}-D array is not built (2-D masks and [D,H,W] volumes are)")
    depth = 0 if v.dim() == 2 else v.shape[0]          # D = 0: a 2-D array -> 4-neighbourhood erosion, like medpy
    if v.dim() == 2:
        v = v[None]
    border = torch.empty_like(v)
    rt.call("wsl_surface_u8", rt.ptr(v), rt.ptr(border), depth, v.shape[1], v.shape[2], rt.stream())
    return torch.nonzero(border).contiguous()          # [n, 3] int64 (z, y, x)


def hd95_percase(pred, gt, voxelspacing=None):
    """medpy.metric.binary.hd95(result, reference) for isotropic unit voxels (what val_2D.py:12 passes)."""
    if voxelspacing is not None:
        raise NotImplementedError("voxelspacing is not built (the reference's validation never passes it)")
    pred, gt = np.asarray(pred).astype(bool), np.asarray(gt).astype(bool)
    if not pred.any():
        raise RuntimeError("The first supplied array does not contain any binary object.")
    if not gt.any():
        raise RuntimeError("The second supplied array does not contain any binary object.")
    a, b = _surface_points(pred), _surface_points(gt)
    d = []
    for p, q in ((a, b), (b, a)):
        out = torch.empty((p.shape[0],), dtype=torch.int64, device=p.device)
        rt.call("wsl_nearest_dist2", rt.ptr(p), p.shape[0], rt.ptr(q), q.shape[0], rt.ptr(out), rt.stream())
        d.append(out)
    dist, _ = torch.sort(torch.sqrt(torch.cat(d).double()))
    n = dist.numel()
    pos = 0.95 * (n - 1)                                # numpy.percentile(..., 95), method 'linear'
    lo = int(np.floor(pos))
    hi = min(lo + 1, n - 1)
    t = pos - lo
    lo_v, hi_v = float(dist[lo]), float(dist[hi])
    diff = hi_v - lo_v
    return hi_v - diff * (1 - t) if t >= 0.5 else lo_v + diff * t


def metric_percase(pred, gt, with_hd95=True):
    """ref: val_2D.py:7-15 calculate_metric_percase -> (dice, hd95), (0, 0) when nothing is predicted."""
    pred, gt = np.asarray(pred) > 0, np.asarray(gt) > 0
    if pred.sum() > 0:
        return dice_percase(pred, gt), (hd95_percase(pred, gt) if with_hd95 else 0.0)
    return 0, 0


def _predict_volume(image, net, patch_size, first_output, slices_per_forward=16):
    """The reference feeds one slice per forward (val_2D.py:22-37, 94-111); the eval forward is per-sample (BatchNorm uses its
    running statistics -- bit-equality of a batch and its parts is tested), so the zoomed slices go through in batches."""
    image = np.asarray(image, dtype=np.float32)
    prediction = np.zeros(image.shape, dtype=np.uint8)
    net.eval()
    x, y = image.shape[1:]
    for i0 in range(0, image.shape[0], slices_per_forward):
        inp = np.stack([zoom(slc, (patch_size[0] / x, patch_size[1] / y), order=0) for slc in image[i0:i0 + slices_per_forward]])
        t = torch.from_numpy(np.ascontiguousarray(inp, dtype=np.float32))[:, None].to(rt.device())
        with torch.no_grad():
            out = net(t)
            if first_output and isinstance(out, (tuple, list)):
                out = out[0]                      # val_2D.py:104: only the main branch is evaluated
            lab = torch.argmax(out, dim=1).cpu().numpy().astype(np.uint8)     # argmax(softmax(z)) == argmax(z)
        for k in range(lab.shape[0]):
            prediction[i0 + k] = zoom(lab[k], (x / patch_size[0], y / patch_size[1]), order=0)
    return prediction


def _squeeze(v):
    a = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    return a[0] if a.ndim == 4 else a            # DataLoader batch dimension of the reference (val_2D.py:19-20)


def test_single_volume(image, label, net, classes, patch_size=(256, 256), with_hd95=True):
    image, label = _squeeze(image), _squeeze(label)
    if image.ndim != 3:
        raise NotImplementedError("2-D slices of a [D,H,W] volume are the built path (the reference's else-branch "
                                  "feeds a single image)")
    prediction = _predict_volume(image, net, patch_size, first_output=False)
    return [metric_percase(prediction == i, label == i, with_hd95) for i in range(1, classes)]


def test_single_volume_cct(image, label, net, classes, patch_size=(256, 256), with_hd95=True):
    image, label = _squeeze(image), _squeeze(label)
    if image.ndim != 3:
        raise NotImplementedError("2-D slices of a [D,H,W] volume are the built path (val_2D.py:116 unpacks four "
                                  "outputs in its else-branch, which no 2-D net returns)")
    prediction = _predict_volume(image, net, patch_size, first_output=True)
    return [metric_percase(prediction == i, label == i, with_hd95) for i in range(1, classes)]


test_single_volume.__test__ = False        # (names kept from the reference; not pytest cases)
test_single_volume_cct.__test__ = False
