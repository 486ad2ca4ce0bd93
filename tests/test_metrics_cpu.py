"""N2: COCO-style mAP of utils/metrics.py against hand-computed answers (build-defined: the reference has
no evaluation and pycocotools is not installable)."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("ssd_metrics", os.path.join(ROOT, "ssd-object-detection_amd", "utils", "metrics.py"))
M = importlib.util.module_from_spec(spec)
spec.loader.exec_module(M)


def test_iou_matrix():
    a = np.array([[5, 5, 10, 10]], float)
    b = np.array([[5, 5, 10, 10], [10, 5, 10, 10], [50, 50, 2, 2]], float)
    assert np.allclose(M.iou_matrix(a, b), [[1.0, 1.0 / 3.0, 0.0]])


def test_perfect_and_empty():
    gt = [(np.array([1, 2]), np.array([[10, 10, 4, 4], [30, 30, 6, 6]], float))]
    det = [(np.array([0.9, 0.8]), np.array([1, 2]), np.array([[10, 10, 4, 4], [30, 30, 6, 6]], float))]
    r = M.coco_map(det, gt)
    assert r["mAP"] == 1.0 and r["AP50"] == 1.0 and r["AP75"] == 1.0
    assert M.coco_map([(np.zeros(0), np.zeros(0), np.zeros((0, 4)))], gt)["mAP"] == 0.0


def test_average_precision_known_answer():
    # 2 ground truths; detections by score: TP, FP, TP -> precision 1, 1/2, 2/3 -> envelope 1, 2/3, 2/3; recall .5, .5, 1
    # 101-point: recall points 0..0.5 (51 points) -> 1.0, 0.51..1.0 (50 points) -> 2/3
    ap = M.average_precision([0.9, 0.8, 0.7], [True, False, True], 2)
    assert abs(ap - (51 * 1.0 + 50 * (2.0 / 3.0)) / 101.0) < 1e-12


def test_threshold_sweep_and_duplicates():
    # one ground truth; a detection with IoU 0.6 counts for thresholds .50, .55, .60 only -> mAP = 3/10
    gt = [(np.array([0]), np.array([[10, 10, 10, 10]], float))]
    # shift so that IoU = 0.6: boxes 10x10, overlap w: iou = w*10 / (200 - w*10) = 0.6 -> w = 7.5
    det = [(np.array([0.9, 0.5]), np.array([0, 0]), np.array([[12.5, 10, 10, 10], [12.5, 10, 10, 10]], float))]
    r = M.coco_map(det, gt)
    assert abs(M.iou_matrix(det[0][2][:1], gt[0][1])[0, 0] - 0.6) < 1e-12
    assert abs(r["mAP"] - 0.3) < 1e-12 and r["AP50"] == 1.0 and r["AP75"] == 0.0   # the duplicate is a false positive after the match
    # a class without ground truth is ignored; wrong-class detections do not help
    det2 = [(np.array([0.9]), np.array([3]), np.array([[10, 10, 10, 10]], float))]
    assert M.coco_map(det2, gt)["mAP"] == 0.0
