"""ctypes loader of oracle/libssd_oracle.so (TEST INFRASTRUCTURE; see ssd_oracle.c)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(HERE, "libssd_oracle.so")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(HERE, "ssd_oracle.c")):
            subprocess.check_call(["make", "-s", "-C", HERE])
        _lib = ctypes.CDLL(so)
        _lib.oracle_match_literal.restype = ctypes.c_int
        _lib.oracle_nms.restype = ctypes.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def match_encode(gt_cls, gt_box, priors, thresh=0.5):
    """Literal match + encode of one image.  Returns (cls i32[A], box f32[A,4], mask bool[A], enc f32[A,4], owner)."""
    L = lib()
    gt_cls = np.ascontiguousarray(gt_cls, np.float32)
    gt_box = np.ascontiguousarray(gt_box, np.float32).reshape(-1, 4)
    priors = np.ascontiguousarray(priors, np.float64)
    A = priors.shape[0]
    cls = np.empty(A, np.int32); box = np.empty((A, 4), np.float32); mask = np.empty(A, np.uint8)
    owner = np.empty(A, np.int32); enc = np.empty((A, 4), np.float32)
    r = L.oracle_match_literal(_p(gt_cls), _p(gt_box), ctypes.c_int(len(gt_cls)), _p(priors), ctypes.c_int(A),
                               ctypes.c_double(thresh), _p(cls), _p(box), _p(mask), _p(owner))
    assert r >= 0, "reference assert (utils/bbox.py:50-51)"
    L.oracle_encode(_p(box), _p(priors), ctypes.c_int(A), _p(enc))
    return cls, box, mask.astype(bool), enc, owner


def iou_n(b1, b2):
    L = lib()
    b1 = np.ascontiguousarray(b1, np.float32); b2 = np.ascontiguousarray(b2, np.float64)
    out = np.empty(len(b1), np.float64)
    L.oracle_iou_n(_p(b1), _p(b2), ctypes.c_int(len(b1)), _p(out))
    return out


def nms(score, cls, box, cand, iou_thresh=0.45, max_cand=1 << 30):
    L = lib()
    score = np.ascontiguousarray(score, np.float32); cls = np.ascontiguousarray(cls, np.int32)
    box = np.ascontiguousarray(box, np.float32); cand = np.ascontiguousarray(cand, np.uint8)
    keep = np.empty(len(score), np.uint8)
    L.oracle_nms(_p(score), _p(cls), _p(box), _p(cand), ctypes.c_int(len(score)), ctypes.c_float(iou_thresh),
                 ctypes.c_int(min(int(max_cand), 1 << 30)), _p(keep))
    return keep.astype(bool)
