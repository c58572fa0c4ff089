"""In-training validation of the reference (ref: code/val_2D.py:18-50 test_single_volume, :90-124
test_single_volume_cct): per-slice zoom to the patch size -> net in eval mode -> argmax over softmax -> zoom back, then
per-class metrics over the volume.  The forward runs on the HIP path; the zoom (scipy, order 0, like the reference) and
the metric are host-side glue.  medpy is not available offline, so Dice is computed here and HD95 is reported as NaN
("hd95/medpy out of scope")."""
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


def _predict_volume(image, net, patch_size, first_output):
    image = np.asarray(image, dtype=np.float32)
    prediction = np.zeros(image.shape, dtype=np.uint8)
    net.eval()
    for ind in range(image.shape[0]):
        slc = image[ind]
        x, y = slc.shape
        inp = zoom(slc, (patch_size[0] / x, patch_size[1] / y), order=0)
        t = torch.from_numpy(np.ascontiguousarray(inp, dtype=np.float32))[None, None].to(rt.device())
        with torch.no_grad():
            out = net(t)
            if first_output and isinstance(out, (tuple, list)):
                out = out[0]                      # val_2D.py:104: only the main branch is evaluated
            lab = torch.argmax(out, dim=1)[0]     # argmax(softmax(z)) == argmax(z)
        pred = zoom(lab.cpu().numpy().astype(np.uint8), (x / patch_size[0], y / patch_size[1]), order=0)
        prediction[ind] = pred
    return prediction


def _squeeze(v):
    a = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    return a[0] if a.ndim == 4 else a            # DataLoader batch dimension of the reference (val_2D.py:19-20)


def test_single_volume(image, label, net, classes, patch_size=(256, 256)):
    image, label = _squeeze(image), _squeeze(label)
    if image.ndim != 3:
        raise NotImplementedError("2-D slices of a [D,H,W] volume are the built path (the reference's else-branch "
                                  "feeds a single image)")
    prediction = _predict_volume(image, net, patch_size, first_output=False)
    return [(dice_percase(prediction == i, label == i), float("nan")) for i in range(1, classes)]


def test_single_volume_cct(image, label, net, classes, patch_size=(256, 256)):
    image, label = _squeeze(image), _squeeze(label)
    if image.ndim != 3:
        raise NotImplementedError("2-D slices of a [D,H,W] volume are the built path (val_2D.py:116 unpacks four "
                                  "outputs in its else-branch, which no 2-D net returns)")
    prediction = _predict_volume(image, net, patch_size, first_output=True)
    return [(dice_percase(prediction == i, label == i), float("nan")) for i in range(1, classes)]


test_single_volume.__test__ = False        # (names kept from the reference; not pytest cases)
test_single_volume_cct.__test__ = False
