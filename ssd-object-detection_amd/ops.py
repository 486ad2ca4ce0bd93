"""Device-tensor wrappers over the C ABI (include/ssd_hip.h).  torch supplies device memory and the
current HIP stream only; every computation runs in libssd_hip.so."""
import ctypes

import numpy as np
import torch

from . import _lib

# SSD300 geometry of the reference (models/ssd_model.py:153,164,176-177)
SSD300_GRIDS = ((38, 38), (19, 19), (10, 10), (5, 5), (3, 3), (1, 1))
SSD300_S_REF = (21, 45, 99, 153, 207, 261, 315)
SSD300_RATIOS = ((2,), (2, 3), (2, 3), (2, 3), (2,), (2,))


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t, dtype):
    assert t.is_cuda and t.dtype == dtype and t.is_contiguous(), (t.device, t.dtype, t.is_contiguous())
    return t


class PriorSet:
    """Default boxes on the device + everything derived from them once (unmatched-row encodings,
    the geometry hint for the matcher)."""

    def __init__(self, priors, enc_zero, grid=None):
        self.priors = priors            # f64 [A,4] device
        self.enc_zero = enc_zero        # f32 [A,4] device
        self.grid = grid                # _lib.PriorGrid or None
        self.A = priors.shape[0]


def make_grid(grids, ratios):
    g = _lib.PriorGrid()
    g.levels = len(grids)
    for i, ((h, w), r) in enumerate(zip(grids, ratios)):
        g.grid_h[i], g.grid_w[i], g.per_cell[i] = h, w, 2 + 2 * len(r)
    return g


def build_priors(grids=SSD300_GRIDS, s_ref=SSD300_S_REF, ratios=SSD300_RATIOS, in_size=300, device="cuda"):
    """ssd_priors + ssd_encode_zero (replaces _build_prior_box, models/ssd_model.py:173-194)."""
    L = _lib.lib()
    levels = len(grids)
    hw = (ctypes.c_int * (2 * levels))(*[v for g in grids for v in g])
    sref = (ctypes.c_double * (levels + 1))(*[float(s) for s in s_ref])
    flat = [r for rr in ratios for r in rr]
    rat = (ctypes.c_int * max(len(flat), 1))(*flat)
    off = [0]
    for rr in ratios:
        off.append(off[-1] + len(rr))
    roff = (ctypes.c_int * (levels + 1))(*off)
    A = L.ssd_priors_count(hw, levels, roff)
    if A <= 0:
        _lib.check(A if A < 0 else _lib.SSD_ERR_VALUE)
    pri = torch.empty((A, 4), dtype=torch.float64, device=device)
    _lib.check(L.ssd_priors(hw, levels, sref, rat, roff, float(in_size), _ptr(pri), _stream()))
    return prior_set_from(pri, make_grid(grids, ratios))


def prior_set_from(priors, grid=None):
    """Wrap an existing device f64 [A,4] prior array (any geometry)."""
    L = _lib.lib()
    priors = _dev(priors, torch.float64)
    A = priors.shape[0]
    enc0 = torch.empty((A, 4), dtype=torch.float32, device=priors.device)
    _lib.check(L.ssd_encode_zero(_ptr(priors), A, _ptr(enc0), _stream()))
    return PriorSet(priors, enc0, grid)


class MatchWorkspace:
    """Caller-owned scratch for ssd_match_encode, grown on demand."""

    def __init__(self):
        self.buf = None

    def get(self, nbytes, device):
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            self.buf = torch.empty((max(nbytes, 256),), dtype=torch.uint8, device=device)
        return self.buf


_default_ws = MatchWorkspace()


def match_encode(gt_box, gt_cls, gt_off, total_gt, max_nt, pset, thresh=0.5, out=None, ws=None, owner=None):
    """Batched match_bbox + apply_anchor_box (utils/bbox.py:44-101) on the device.

    gt_box f32[total_gt,4], gt_cls f32[total_gt], gt_off i32[B+1] device tensors; total_gt / max_nt are
    host ints.  `owner` (optional i32[B,A]) receives the matched gt row per anchor (-1 = none).
    Returns (cls i32[B,A], loc f32[B,A,4], mask u8[B,A])."""
    L = _lib.lib()
    B = gt_off.numel() - 1
    A = pset.A
    dev = pset.priors.device
    _dev(gt_off, torch.int32)
    if total_gt > 0:
        _dev(gt_box, torch.float32)
        _dev(gt_cls, torch.float32)
    if out is None:
        out = (torch.empty((B, A), dtype=torch.int32, device=dev),
               torch.empty((B, A, 4), dtype=torch.float32, device=dev),
               torch.empty((B, A), dtype=torch.uint8, device=dev))
    o_cls, o_loc, o_mask = out
    nbytes = L.ssd_match_encode_workspace_bytes(B, A, total_gt)
    wbuf = (ws or _default_ws).get(nbytes, dev)
    grid = ctypes.byref(pset.grid) if pset.grid is not None else None
    _lib.check(L.ssd_match_encode(_ptr(gt_box), _ptr(gt_cls), _ptr(gt_off), B, int(total_gt), int(max_nt),
                                  _ptr(pset.priors), _ptr(pset.enc_zero), A, grid, float(thresh),
                                  _ptr(o_cls), _ptr(o_loc), _ptr(o_mask), _ptr(owner), _ptr(wbuf), wbuf.numel(),
                                  _stream()))
    return o_cls, o_loc, o_mask


def pack_gt(boxes_list, cls_list, device="cuda"):
    """Concatenate per-image numpy gts into the (gt_box, gt_cls, gt_off, total, max_nt) batch form."""
    counts = [int(np.shape(b)[0]) for b in boxes_list]
    off = np.zeros(len(counts) + 1, np.int32)
    off[1:] = np.cumsum(counts)
    total = int(off[-1])
    if total:
        box = np.concatenate([np.asarray(b, np.float32).reshape(-1, 4) for b in boxes_list], 0)
        cls = np.concatenate([np.asarray(c, np.float32).reshape(-1) for c in cls_list], 0)
    else:
        box, cls = np.zeros((0, 4), np.float32), np.zeros((0,), np.float32)
    return (torch.from_numpy(box).to(device), torch.from_numpy(cls).to(device),
            torch.from_numpy(off).to(device), total, max(counts) if counts else 0)


def iou_n(b1, b2):
    """ssd_iou_n: f32 [n,4] vs f64 [n,4] device tensors -> f64 [n]."""
    L = _lib.lib()
    _dev(b1, torch.float32)
    _dev(b2, torch.float64)
    n = b1.shape[0]
    out = torch.empty((n,), dtype=torch.float64, device=b1.device)
    _lib.check(L.ssd_iou_n(_ptr(b1), _ptr(b2), n, _ptr(out), _stream()))
    return out


def apply_anchor_box(box, priors):
    """ssd_apply_anchor_box: f32 [n,4] vs f64 [n,4] device tensors -> f64 [n,4]."""
    L = _lib.lib()
    _dev(box, torch.float32)
    _dev(priors, torch.float64)
    n = box.shape[0]
    out = torch.empty((n, 4), dtype=torch.float64, device=box.device)
    _lib.check(L.ssd_apply_anchor_box(_ptr(box), _ptr(priors), n, _ptr(out), _stream()))
    return out


_loss_ws = MatchWorkspace()


def ssd_loss(conf, loc, gt_cls, gt_loc, gt_mask, grad_scale=1.0, ws=None):
    """ssd_loss_fwd_bwd (replaces _ssd_loss, models/ssd_model.py:341-396, and its gradient).

    conf [B,A,C], loc [B,A,4] (both f32 or both bf16); gt_* from match_encode.
    Returns (out8 f32[8] device tensor, dconf, dloc).  out8 = loc, pos, neg, total, P, N, tau, status."""
    L = _lib.lib()
    B, A, C = conf.shape
    assert conf.dtype == loc.dtype and conf.dtype in (torch.float32, torch.bfloat16)
    assert conf.is_cuda and conf.is_contiguous() and loc.is_contiguous()
    assert loc.shape == (B, A, 4) and gt_loc.shape == (B, A, 4)          # models/ssd_model.py:347-351
    assert gt_cls.shape == (B, A) and gt_mask.shape == (B, A)
    _dev(gt_cls, torch.int32); _dev(gt_loc, torch.float32); _dev(gt_mask, torch.uint8)
    dtype = 0 if conf.dtype == torch.float32 else 1
    out = torch.empty((8,), dtype=torch.float32, device=conf.device)
    dconf = torch.empty_like(conf)
    dloc = torch.empty_like(loc)
    nbytes = L.ssd_loss_workspace_bytes(B, A, C)
    wbuf = (ws or _loss_ws).get(nbytes, conf.device)
    _lib.check(L.ssd_loss_fwd_bwd(_ptr(conf), _ptr(loc), dtype, _ptr(gt_cls), _ptr(gt_loc), _ptr(gt_mask), B, A, C,
                                  float(grad_scale), _ptr(out), _ptr(dconf), _ptr(dloc), _ptr(wbuf), wbuf.numel(),
                                  _stream()))
    return out, dconf, dloc


def score_decode(conf, loc, pset, score_thresh=0.5, in_size=300.0):
    """ssd_score_decode (replaces visualize's scoring, models/ssd_model.py:479-488, + decode :466-467).
    Returns (score f32[B,A], cls i32[B,A], box f32[B,A,4] pixels, cand u8[B,A])."""
    L = _lib.lib()
    B, A, C = conf.shape
    assert conf.dtype == loc.dtype and conf.dtype in (torch.float32, torch.bfloat16)
    assert conf.is_cuda and conf.is_contiguous() and loc.is_contiguous() and A == pset.A
    dev = conf.device
    score = torch.empty((B, A), dtype=torch.float32, device=dev)
    cls = torch.empty((B, A), dtype=torch.int32, device=dev)
    box = torch.empty((B, A, 4), dtype=torch.float32, device=dev)
    cand = torch.empty((B, A), dtype=torch.uint8, device=dev)
    _lib.check(L.ssd_score_decode(_ptr(conf), _ptr(loc), 0 if conf.dtype == torch.float32 else 1, _ptr(pset.priors),
                                  B, A, C, float(score_thresh), float(in_size), _ptr(score), _ptr(cls), _ptr(box),
                                  _ptr(cand), _stream()))
    return score, cls, box, cand


def nms(score, cls, box, cand, iou_thresh=0.45, max_cand=400, want_count=False):
    """ssd_nms: per-image per-class greedy NMS (build-defined; the reference has none).
    Returns keep u8[B,A] (and keep_count i32[B] if want_count)."""
    L = _lib.lib()
    B, A = score.shape
    _dev(score, torch.float32); _dev(cls, torch.int32); _dev(box, torch.float32); _dev(cand, torch.uint8)
    keep = torch.empty((B, A), dtype=torch.uint8, device=score.device)
    count = torch.empty((B,), dtype=torch.int32, device=score.device) if want_count else None
    _lib.check(L.ssd_nms(_ptr(score), _ptr(cls), _ptr(box), _ptr(cand), B, A, float(iou_thresh), int(max_cand),
                         _ptr(keep), _ptr(count), _stream()))
    return (keep, count) if want_count else keep
