"""Host-side mirror of the reference's utils/bbox.py call surface (same names, argument meaning,
return dtypes and exceptions), executed by the gfx950 library.

    iou_n(n_bbox_1, n_bbox_2)                      reference utils/bbox.py:28-41
    match_bbox(cls, bbox, default_box, thresh)     reference utils/bbox.py:44-91
    apply_anchor_box(origin_bbox, default_box)     reference utils/bbox.py:94-101

These numpy-in / numpy-out functions exist so that code written against the reference keeps
working unchanged; each call uploads its arguments, runs the HIP kernels and downloads the result.
The training pipeline does not go through them -- it uses the batched device path
(ops.match_encode), which is the same kernel without the per-image host round trip.
"""
import numpy as np
import torch

from .. import ops

_cache = {}


def _prior_set(default_box):
    """Device copy of a prior array (+ derived tables), cached by content: the reference passes the
    same 8732x4 array on every call."""
    arr = np.ascontiguousarray(np.asarray(default_box, dtype=np.float64))
    key = (arr.shape, hash(arr.tobytes()))
    ps = _cache.get(key)
    if ps is None:
        if len(_cache) > 8:
            _cache.clear()
        ps = ops.prior_set_from(torch.from_numpy(arr).cuda())
        _cache[key] = ps
    return ps


def iou_n(n_bbox_1, n_bbox_2):
    """IoU of paired (cx,cy,w,h) rows; float32 box_1 against float64 box_2 -> float64 [P]."""
    b1 = np.ascontiguousarray(np.asarray(n_bbox_1, dtype=np.float32))
    b2 = np.ascontiguousarray(np.asarray(n_bbox_2, dtype=np.float64))
    return ops.iou_n(torch.from_numpy(b1).cuda(), torch.from_numpy(b2).cuda()).cpu().numpy()


def match_bbox(cls, bbox, default_box, thresh=0.5):
    """Two-phase anchor matching of one image.  Returns (labeled_cls int32[A], labeled_boxes
    float32[A,4], mask bool[A]) exactly as the reference does (unmatched rows are 0 / False).

    Raises AssertionError when n_targets > n_defaults or thresh <= 0 (reference :50-51).
    Contract: finite boxes with w,h >= 0; gt arrays are used at float32 precision as the
    reference's loaders emit them."""
    target_cls = np.asarray(cls, dtype=np.float32).reshape(-1)
    target_box = np.asarray(bbox, dtype=np.float32).reshape(-1, 4)
    pset = _prior_set(default_box)
    n_t = target_box.shape[0]
    assert n_t <= pset.A, "number of default boxes should greater than the number of targets"
    assert thresh > 0.0, "thresh should greater than zero"
    gt_box, gt_cls, gt_off, total, max_nt = ops.pack_gt([target_box], [target_cls])
    owner = torch.empty((1, pset.A), dtype=torch.int32, device="cuda")
    o_cls, _, o_mask = ops.match_encode(gt_box, gt_cls, gt_off, total, max_nt, pset, thresh, owner=owner)
    own = owner[0].cpu().numpy()
    mask = o_mask[0].cpu().numpy().astype(bool)
    labeled_cls = o_cls[0].cpu().numpy()
    labeled_boxes = np.zeros((pset.A, 4), dtype=np.float32)
    labeled_boxes[mask] = target_box[own[mask]]
    return labeled_cls, labeled_boxes, mask


def apply_anchor_box(origin_bbox, default_box):
    """Encode boxes against priors: float64 [n,4] = ((g_xy-d_xy)/d_wh, log(max(g_wh,1e-5)/max(d_wh,1e-5)))."""
    assert np.shape(origin_bbox) == np.shape(default_box)
    box = np.ascontiguousarray(np.asarray(origin_bbox, dtype=np.float32).reshape(-1, 4))
    pri = np.ascontiguousarray(np.asarray(default_box, dtype=np.float64).reshape(-1, 4))
    return ops.apply_anchor_box(torch.from_numpy(box).cuda(), torch.from_numpy(pri).cuda()).cpu().numpy()
