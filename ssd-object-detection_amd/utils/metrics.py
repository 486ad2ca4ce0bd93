"""COCO-style mean average precision.

The reference fetches a validation split and drops it (models/ssd_model.py:291) and has no evaluation; this module is
build-defined, like the NMS it gives a consumer to.  It follows the published COCO protocol for boxes without crowd
regions: per class and per IoU threshold t in {0.50, 0.55, ..., 0.95}, detections are taken in order of decreasing score,
each claims the still-unmatched ground truth of its image with the highest IoU >= t, precision is made monotone from the
right and sampled at the 101 recall points 0, 0.01, ..., 1; AP = mean of the samples; mAP = mean over thresholds and over
the classes that have ground truth.  Boxes are (cx, cy, w, h) in any common unit; at most `max_dets` detections per image."""
import numpy as np

IOU_THRESHOLDS = np.linspace(0.5, 0.95, 10)
RECALL_POINTS = np.linspace(0.0, 1.0, 101)


def _corners(b):
    b = np.asarray(b, np.float64).reshape(-1, 4)
    return np.stack([b[:, 0] - b[:, 2] / 2, b[:, 1] - b[:, 3] / 2, b[:, 0] + b[:, 2] / 2, b[:, 1] + b[:, 3] / 2], 1)


def iou_matrix(a, b):
    """IoU of every box of a [n,4] with every box of b [m,4] (cx, cy, w, h) -> [n, m] float64."""
    ca, cb = _corners(a), _corners(b)
    lt = np.maximum(ca[:, None, :2], cb[None, :, :2])
    rb = np.minimum(ca[:, None, 2:], cb[None, :, 2:])
    wh = np.clip(rb - lt, 0.0, None)
    inter = wh[..., 0] * wh[..., 1]
    area_a = (ca[:, 2] - ca[:, 0]) * (ca[:, 3] - ca[:, 1])
    area_b = (cb[:, 2] - cb[:, 0]) * (cb[:, 3] - cb[:, 1])
    union = area_a[:, None] + area_b[None, :] - inter
    return np.where(union > 0, inter / np.maximum(union, 1e-300), 0.0)


def average_precision(scores, matched, n_gt):
    """AP (101-point) of one class at one threshold: scores [k], matched bool [k] (true positive flags), n_gt > 0."""
    if len(scores) == 0:
        return 0.0
    order = np.argsort(-np.asarray(scores, np.float64), kind="mergesort")
    tp = np.asarray(matched, bool)[order]
    ctp = np.cumsum(tp)
    cfp = np.cumsum(~tp)
    recall = ctp / float(n_gt)
    precision = ctp / np.maximum(ctp + cfp, 1)
    for i in range(len(precision) - 2, -1, -1):               # monotone envelope from the right
        precision[i] = max(precision[i], precision[i + 1])
    idx = np.searchsorted(recall, RECALL_POINTS, side="left")
    sampled = np.where(idx < len(precision), precision[np.minimum(idx, len(precision) - 1)], 0.0)
    return float(sampled.mean())


def coco_map(detections, ground_truths, max_dets=100):
    """detections: per image (score [k], cls [k], box [k,4]); ground_truths: per image (cls [n], box [n,4]).
    Returns dict(mAP=..., AP50=..., AP75=..., per_class={cls: AP@[.5:.95]})."""
    per_class_dets, per_class_ngt = {}, {}
    for img, (gcls, gbox) in enumerate(ground_truths):
        for c in np.asarray(gcls).astype(int):
            per_class_ngt[c] = per_class_ngt.get(c, 0) + 1
    ap = {}                                                   # (cls, threshold index) -> AP
    classes = sorted(per_class_ngt)
    flags = {c: [[] for _ in IOU_THRESHOLDS] for c in classes}
    scores = {c: [] for c in classes}
    for img, (dscore, dcls, dbox) in enumerate(detections):
        dscore = np.asarray(dscore, np.float64)
        dcls = np.asarray(dcls).astype(int)
        dbox = np.asarray(dbox, np.float64).reshape(-1, 4)
        keep = np.argsort(-dscore, kind="mergesort")[:max_dets]
        dscore, dcls, dbox = dscore[keep], dcls[keep], dbox[keep]
        gcls = np.asarray(ground_truths[img][0]).astype(int)
        gbox = np.asarray(ground_truths[img][1], np.float64).reshape(-1, 4)
        for c in np.unique(dcls):
            if c not in per_class_ngt:
                continue                                      # class without ground truth anywhere: not part of the mean
            d_idx = np.nonzero(dcls == c)[0]
            g_idx = np.nonzero(gcls == c)[0]
            iou = iou_matrix(dbox[d_idx], gbox[g_idx]) if len(g_idx) else np.zeros((len(d_idx), 0))
            scores[c].extend(dscore[d_idx].tolist())
            for ti, thr in enumerate(IOU_THRESHOLDS):
                taken = np.zeros(len(g_idx), bool)
                for k in range(len(d_idx)):                   # already in decreasing score order
                    best, best_j = thr, -1
                    for j in range(len(g_idx)):
                        if not taken[j] and iou[k, j] >= best:
                            best, best_j = iou[k, j], j
                    if best_j >= 0:
                        taken[best_j] = True
                    flags[c][ti].append(best_j >= 0)
    per_class = {}
    ap50, ap75 = [], []
    for c in classes:
        aps = [average_precision(scores[c], flags[c][ti], per_class_ngt[c]) for ti in range(len(IOU_THRESHOLDS))]
        per_class[c] = float(np.mean(aps))
        ap50.append(aps[0])
        ap75.append(aps[5])
    if not classes:
        return dict(mAP=0.0, AP50=0.0, AP75=0.0, per_class={})
    return dict(mAP=float(np.mean(list(per_class.values()))), AP50=float(np.mean(ap50)), AP75=float(np.mean(ap75)),
                per_class=per_class)
