"""CPU oracle for the SSD hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (ssd-object-detection_amd/) never imports it and has no CPU fallback.

Every function restates, in plain numpy, what the reference does on the path named by
BASELINE.json:north_star; the reference file:line each one follows is cited in its docstring
(paths relative to /root/reference).

Pinning status:
  * priors, iou_n, match (literal + closed form), encode: PINNED bit-exactly against golden
    vectors captured from the reference itself (tests/golden/*.npz, made by
    tests/golden/gen_golden.py) and against the reference's own asserted test cases
    (tests/utils/test_bbox.py:35-44).  See tests/test_oracle_golden.py.
  * scalar iou: pinned by the 8 known-answer cases of tests/utils/test_bbox.py:10-17.
  * loss, score/decode, clip+Adam: PARITY UNPINNED -- they execute TensorFlow ops in the
    reference (tensorflow>=2.4.0, unpinned, requirements.txt:1; absent in this image), so they
    follow models/ssd_model.py and the published TF op semantics only.
  * NMS: no reference counterpart at all; this file *defines* it.
"""
import math

import numpy as np

# --------------------------------------------------------------------------------------
# A2  default boxes
# --------------------------------------------------------------------------------------
SSD300_GRIDS = ((38, 38), (19, 19), (10, 10), (5, 5), (3, 3), (1, 1))
SSD300_S_REF = (21, 45, 99, 153, 207, 261, 315)
SSD300_RATIOS = ((2,), (2, 3), (2, 3), (2, 3), (2,), (2,))
SSD300_IN_SIZE = 300


def priors(grids=SSD300_GRIDS, s_ref=SSD300_S_REF, ratios=SSD300_RATIOS, in_size=SSD300_IN_SIZE):
    """models/ssd_model.py:173-194 (_build_prior_box).  float64 [A,4] (cx,cy,w,h), unclipped.
    Order per level: y outer, x inner; per cell: s_k square, s' square, then for each ratio r
    the (s_k*sqrt r, s_k/sqrt r) box followed by its transpose."""
    rows = []
    for lvl, (gh, gw) in enumerate(grids):
        s_k = s_ref[lvl] / in_size
        s_next = s_ref[lvl + 1] / in_size
        s_prime = math.sqrt(s_k * s_next)
        for y in range(gh):
            cy = (y + 0.5) / gh
            for x in range(gw):
                cx = (x + 0.5) / gw
                rows.append((cx, cy, s_k, s_k))
                rows.append((cx, cy, s_prime, s_prime))
                for r in ratios[lvl]:
                    q = math.sqrt(r)
                    rows.append((cx, cy, s_k * q, s_k / q))
                    rows.append((cx, cy, s_k / q, s_k * q))
    return np.asarray(rows, dtype=np.float64)


# --------------------------------------------------------------------------------------
# A4  IoU
# --------------------------------------------------------------------------------------
def iou_n(b1, b2):
    """utils/bbox.py:28-41.  Element-wise IoU of paired (cx,cy,w,h) rows; each intersection side is
    clamped at 1e-10 (NOT 0) and 1e-10 is added to the union.  Arithmetic runs in whatever
    dtypes numpy promotion yields -- as called from match (f32 gt, f64 priors): gt corners and
    gt area in f32, prior corners/area in f64, everything after the first max/min in f64."""
    x1, y1, w1, h1 = b1[..., 0], b1[..., 1], b1[..., 2], b1[..., 3]
    x2, y2, w2, h2 = b2[..., 0], b2[..., 1], b2[..., 2], b2[..., 3]
    a1 = w1 * h1
    a2 = w2 * h2
    lo_x = np.maximum(x1 - w1 / 2, x2 - w2 / 2)
    lo_y = np.maximum(y1 - h1 / 2, y2 - h2 / 2)
    hi_x = np.minimum(x1 + w1 / 2, x2 + w2 / 2)
    hi_y = np.minimum(y1 + h1 / 2, y2 + h2 / 2)
    inter = np.maximum(1e-10, hi_x - lo_x) * np.maximum(1e-10, hi_y - lo_y)
    return inter / (a1 + a2 - inter + 1e-10)


def iou_scalar(b1, b2):
    """utils/bbox.py:6-25 (scalar `iou`, sides clamped at 0.0).  Evaluated in float32, the dtype
    TF gives python-float inputs.  The build's NMS uses exactly this arithmetic."""
    f = np.float32
    x1, y1, w1, h1 = (f(v) for v in b1)
    x2, y2, w2, h2 = (f(v) for v in b2)
    a1 = w1 * h1
    a2 = w2 * h2
    lo_x = max(x1 - w1 / f(2), x2 - w2 / f(2))
    lo_y = max(y1 - h1 / f(2), y2 - h2 / f(2))
    hi_x = min(x1 + w1 / f(2), x2 + w2 / f(2))
    hi_y = min(y1 + h1 / f(2), y2 + h2 / f(2))
    inter = max(f(0.0), hi_x - lo_x) * max(f(0.0), hi_y - lo_y)
    return inter / (a1 + a2 - inter + f(1e-10))


# --------------------------------------------------------------------------------------
# A3  matching
# --------------------------------------------------------------------------------------
def iou_matrix(gt_box, default_box):
    """utils/bbox.py:53-58: all n_t x A pairs, gt-major.  (Broadcast instead of np.repeat:
    same element-wise arithmetic and dtypes.)"""
    gt_box = np.asarray(gt_box)
    default_box = np.asarray(default_box)
    return np.ascontiguousarray(iou_n(gt_box[:, None, :], default_box[None, :, :]))


def match_literal(gt_cls, gt_box, default_box, thresh=0.5):
    """utils/bbox.py:44-91 (match_bbox), literal two-phase algorithm including one full-matrix
    argmax per recorded pair.  Returns (cls int32[A], boxes float32[A,4], mask bool[A])."""
    gt_cls = np.asarray(gt_cls)
    gt_box = np.asarray(gt_box)
    default_box = np.asarray(default_box)
    n_t, n_d = gt_box.shape[0], default_box.shape[0]
    assert n_t <= n_d, "number of default boxes should greater than the number of targets"   # :50
    assert thresh > 0.0, "thresh should greater than zero"                                  # :51
    live = iou_matrix(gt_box, default_box)
    work = live.copy()
    picks = []
    for _ in range(n_t):                                    # phase 1, :62-68
        r, c = np.unravel_index(np.argmax(work), work.shape)
        work[r, :] = 0.0
        work[:, c] = 0.0
        live[:, c] = 0.0
        picks.append((r, c))
    while True:                                             # phase 2, :71-79
        r, c = np.unravel_index(np.argmax(live), live.shape)
        if live[r, c] <= thresh:
            break
        picks.append((r, c))
        live[:, c] = 0.0
    mask = np.zeros((n_d,), dtype=bool)                     # scatter, :84-90
    boxes = np.zeros((n_d, 4), dtype=np.float32)
    cls = np.zeros((n_d,), dtype=np.int32)
    for r, c in picks:
        mask[c] = True
        boxes[c, :] = gt_box[r, :]
        cls[c] = int(gt_cls[r])
    return cls, boxes, mask


def match_closed_form(gt_cls, gt_box, default_box, thresh=0.5):
    """Same result as match_literal for valid boxes (w,h >= 0 => every IoU > 0), computed the way
    the HIP kernel does:
      phase 1 -- n_t rounds; each round takes, over still-free rows and columns, the largest IoU,
                 ties to the lowest row then lowest column (row-major argmax);
      phase 2 -- every column not taken in phase 1 whose column-max exceeds thresh goes to the
                 lowest row attaining that max."""
    gt_cls = np.asarray(gt_cls)
    gt_box = np.asarray(gt_box)
    default_box = np.asarray(default_box)
    n_t, n_d = gt_box.shape[0], default_box.shape[0]
    assert n_t <= n_d and thresh > 0.0
    m = iou_matrix(gt_box, default_box)
    owner = np.full((n_d,), -1, dtype=np.int64)
    row_free = np.ones((n_t,), dtype=bool)
    col_free = np.ones((n_d,), dtype=bool)
    for _ in range(n_t):
        sub = np.where(row_free[:, None] & col_free[None, :], m, -np.inf)
        r, c = np.unravel_index(np.argmax(sub), sub.shape)
        owner[c] = r
        row_free[r] = False
        col_free[c] = False
    col_best_r = np.argmax(m, axis=0)                       # lowest r attaining the max
    col_best_v = m[col_best_r, np.arange(n_d)]
    take = col_free & (col_best_v > thresh)
    owner[take] = col_best_r[take]
    mask = owner >= 0
    boxes = np.zeros((n_d, 4), dtype=np.float32)
    cls = np.zeros((n_d,), dtype=np.int32)
    boxes[mask] = gt_box[owner[mask]]
    cls[mask] = gt_cls[owner[mask]].astype(np.int32)        # int(float) truncation, :90
    return cls, boxes, mask


# --------------------------------------------------------------------------------------
# A5  box encoding
# --------------------------------------------------------------------------------------
def encode(matched_box, default_box):
    """utils/bbox.py:94-101 (apply_anchor_box).  f32 boxes vs f64 priors -> f64 result; the
    caller's TensorSpec (models/ssd_model.py:222) casts to f32.  np.maximum(f32, 1e-5) keeps
    f32 (python scalar is weak), so the gt-side clamp constant is float32(1e-5)."""
    matched_box = np.asarray(matched_box)
    default_box = np.asarray(default_box)
    assert matched_box.shape == default_box.shape                                            # :95
    t_xy = (matched_box[:, :2] - default_box[:, :2]) / default_box[:, 2:]
    t_wh = np.log(np.maximum(matched_box[:, 2:], 1e-5) / np.maximum(default_box[:, 2:], 1e-5))
    return np.concatenate([t_xy, t_wh], axis=-1)


def match_encode(gt_cls, gt_box, default_box, thresh=0.5, literal=True):
    """models/ssd_model.py:212-213 + the TensorSpec cast at :222: what one image contributes to a
    training batch.  Returns (cls i32[A], loc f32[A,4], mask bool[A])."""
    fn = match_literal if literal else match_closed_form
    cls, boxes, mask = fn(gt_cls, gt_box, default_box, thresh)
    return cls, encode(boxes, default_box).astype(np.float32), mask


# --------------------------------------------------------------------------------------
# A6  loss (PARITY UNPINNED: TF ops)
# --------------------------------------------------------------------------------------
def _log_softmax(z):
    z = np.asarray(z, dtype=np.float64)
    zmax = z.max(axis=-1, keepdims=True)
    e = z - zmax
    return e - np.log(np.exp(e).sum(axis=-1, keepdims=True))


def ssd_loss(gt_cls, gt_box, gt_mask, pred_box, pred_cls, want_grad=False):
    """models/ssd_model.py:341-396 (_ssd_loss), evaluated in float64 from the given inputs.
    CE = logsumexp(z) - z[label] (tf.nn.sparse_softmax_cross_entropy_with_logits).
    Returns dict(loc, pos, neg, total, num_pos, num_neg, tau[, dbox, dcls])."""
    gt_cls = np.asarray(gt_cls)
    gt_mask = np.asarray(gt_mask).astype(bool)
    gt_box = np.asarray(gt_box, dtype=np.float64)
    pred_box = np.asarray(pred_box, dtype=np.float64)
    pred_cls = np.asarray(pred_cls, dtype=np.float64)
    B, A, C = pred_cls.shape
    assert gt_cls.shape == (B, A) and gt_mask.shape == (B, A)                                # :347-351
    assert gt_box.shape == (B, A, 4) and pred_box.shape == (B, A, 4)
    logp = _log_softmax(pred_cls)
    pos = gt_mask
    P = int(pos.sum())
    ce_gt = -np.take_along_axis(logp, gt_cls[..., None].astype(np.int64), axis=-1)[..., 0]
    l_pos = (ce_gt * pos).sum() / P                                                          # :356-358
    ce_bg = -logp[..., C - 1] * (~pos)                                                       # :362-367
    k = 3 * P                                                                                # :368
    flat = ce_bg.reshape(-1)
    if k > flat.size:
        raise ValueError("top_k: k exceeds the number of anchors in the (micro)batch")
    tau = np.partition(flat, flat.size - k)[flat.size - k]                                   # k-th largest, :369
    neg = ce_bg >= tau                                                                       # :372
    assert not (neg & pos).any()                                                             # :375
    N = int(neg.sum())
    l_neg = (ce_bg * neg).sum() / N                                                          # :378-380
    diff = pred_box - gt_box
    l_loc = (np.abs(diff).sum(axis=-1) * pos).sum() / P                                      # :383-386
    out = dict(loc=l_loc, pos=l_pos, neg=l_neg, total=l_loc + l_pos + l_neg,
               num_pos=P, num_neg=N, tau=tau, neg_mask=neg)
    if want_grad:
        p = np.exp(logp)
        onehot_gt = np.zeros_like(p)
        np.put_along_axis(onehot_gt, gt_cls[..., None].astype(np.int64), 1.0, axis=-1)
        onehot_bg = np.zeros_like(p)
        onehot_bg[..., C - 1] = 1.0
        dcls = (p - onehot_gt) * (pos[..., None] / P) + (p - onehot_bg) * (neg[..., None] / N)
        dbox = np.sign(diff) * (pos[..., None] / P)
        out.update(dcls=dcls, dbox=dbox)
    return out


# --------------------------------------------------------------------------------------
# A9  scoring + decode (PARITY UNPINNED: TF ops), A9' NMS (build-defined)
# --------------------------------------------------------------------------------------
def score(pred_cls, thresh=0.5):
    """models/ssd_model.py:479-488 (visualize, mask=None branch): softmax; best non-background
    probability; candidate = score > thresh and not (p_bg > thresh); class = argmax (equals the
    best foreground class wherever candidate is True).  float64 evaluation."""
    p = np.exp(_log_softmax(pred_cls))
    fg = p[..., :-1]
    s = fg.max(axis=-1)
    c = fg.argmax(axis=-1).astype(np.int32)
    cand = (s > thresh) & ~(p[..., -1] > thresh)
    return s, c, cand


def decode(pred_box, default_box, in_size=300):
    """models/ssd_model.py:466-467: cxcy = (t_xy*d_wh + d_xy)*300; wh = exp(t_wh)*d_wh*300
    (f32 offsets, f64 priors -> f64, stored to f32)."""
    pred_box = np.asarray(pred_box)
    d = np.asarray(default_box)
    xy = (pred_box[..., :2] * d[..., 2:] + d[..., :2]) * in_size
    wh = np.exp(pred_box[..., 2:]) * d[..., 2:] * in_size
    return np.concatenate([xy, wh], axis=-1).astype(np.float32)


def iou_f32_rows(box, boxes):
    """iou_scalar vectorised: float32 (cx,cy,w,h) `box` against rows of `boxes`, all float32."""
    f = np.float32
    box = box.astype(f)
    boxes = boxes.astype(f)
    a1 = box[2] * box[3]
    a2 = boxes[:, 2] * boxes[:, 3]
    lo_x = np.maximum(box[0] - box[2] / f(2), boxes[:, 0] - boxes[:, 2] / f(2))
    lo_y = np.maximum(box[1] - box[3] / f(2), boxes[:, 1] - boxes[:, 3] / f(2))
    hi_x = np.minimum(box[0] + box[2] / f(2), boxes[:, 0] + boxes[:, 2] / f(2))
    hi_y = np.minimum(box[1] + box[3] / f(2), boxes[:, 1] + boxes[:, 3] / f(2))
    inter = np.maximum(f(0.0), hi_x - lo_x) * np.maximum(f(0.0), hi_y - lo_y)
    return inter / (a1 + a2 - inter + f(1e-10))


def nms(score_, cls_, box_, cand_, iou_thresh=0.45, max_cand=None):
    """Build-defined per-image, per-class greedy hard NMS (SURVEY.md row A9'; the reference has
    none).  Candidates are ordered by (score desc, anchor index asc); if max_cand is given only
    the first max_cand candidates of the image (in that order) take part.  A candidate is kept
    iff its float32 IoU (iou_scalar arithmetic) with every already-kept candidate of the same
    class is <= iou_thresh.  Returns keep bool[A]."""
    score_ = np.asarray(score_, np.float32)
    cls_ = np.asarray(cls_)
    box_ = np.asarray(box_, np.float32)
    idx = np.nonzero(np.asarray(cand_))[0]
    order = idx[np.lexsort((idx, -score_[idx].astype(np.float64)))]
    if max_cand is not None:
        order = order[:max_cand]
    keep = np.zeros(score_.shape[0], dtype=bool)
    kept = []
    thr = np.float32(iou_thresh)
    for i in order:
        same = [j for j in kept if cls_[j] == cls_[i]]
        if same:
            v = iou_f32_rows(box_[i], box_[same])
            if (v > thr).any():
                continue
        kept.append(i)
        keep[i] = True
    return keep


# --------------------------------------------------------------------------------------
# A7  per-tensor clip + Adam (PARITY UNPINNED: TF/Keras ops)
# --------------------------------------------------------------------------------------
def clip_by_norm(g, clip=0.01):
    """tf.clip_by_norm (models/ssd_model.py:249): g * clip / max(||g||_2, clip)."""
    g = np.asarray(g, np.float64)
    n = math.sqrt(float((g * g).sum()))
    return g * (clip / max(n, clip))


def exponential_decay(step, initial, decay_steps, decay_rate):
    """tf.keras ExponentialDecay, staircase=False (tools/train.py:31-35)."""
    return initial * decay_rate ** (step / decay_steps)


def polynomial_decay(step, initial, decay_steps, end, power=1.0):
    """tf.keras PolynomialDecay, cycle=False (tools/train.py:36-40)."""
    s = min(step, decay_steps)
    return (initial - end) * (1 - s / decay_steps) ** power + end


def adam_step(p, g, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-7):
    """Keras Adam (non-amsgrad) update number t (1-based): lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
    m,v EMA; p -= lr_t*m/(sqrt(v)+eps)."""
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    lr_t = lr * math.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    p = p - lr_t * m / (np.sqrt(v) + eps)
    return p, m, v


# --------------------------------------------------------------------------------------
# N1: input-side preprocessing.  PARITY UNPINNED for the resize: cv2 is not installable in the build
# image, so `resize_bilinear` restates OpenCV's published INTER_LINEAR rule for float images (what
# data_loaders/ssd/make_dataset.py:40 calls with the default interpolation) instead of being checked against it.
# --------------------------------------------------------------------------------------
def _linear_coords(n_dst, n_src):
    """cv2.resize INTER_LINEAR source taps of every destination index: fx = (float)((d + 0.5) * scale - 0.5),
    s = floor(fx), f = fx - s, with s < 0 -> (0, f = 0) and s >= n-1 -> (n-1, f = 0); second tap clamped to n-1."""
    scale = float(n_src) / float(n_dst)
    d = np.arange(n_dst, dtype=np.float64)
    fx = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(fx).astype(np.int64)
    f = (fx - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    s[lo] = 0
    f[lo] = 0.0
    hi = s >= n_src - 1
    s[hi] = n_src - 1
    f[hi] = 0.0
    s1 = np.minimum(s + 1, n_src - 1)
    return s, s1, f


def resize_bilinear(img_f32, size):
    """img_f32 [H,W,C] float32 -> [size,size,C] float32; horizontal pass then vertical pass, float32 products and sums
    (no fused multiply-add), as OpenCV's generic float path does."""
    H, W, _ = img_f32.shape
    x0, x1, fx = _linear_coords(size, W)
    y0, y1, fy = _linear_coords(size, H)
    a0 = (np.float32(1.0) - fx)[None, :, None]
    a1 = fx[None, :, None]
    rows = (img_f32[:, x0, :] * a0).astype(np.float32) + (img_f32[:, x1, :] * a1).astype(np.float32)   # [H, size, C]
    rows = rows.astype(np.float32)
    b0 = (np.float32(1.0) - fy)[:, None, None]
    b1 = fy[:, None, None]
    out = (rows[y0] * b0).astype(np.float32) + (rows[y1] * b1).astype(np.float32)
    return out.astype(np.float32)


def image_resize_prep(img_u8, size=300, normalize=True):
    """uint8 [H,W,3] -> float32 [size,size,3]: /255 in float64 then float32 (data_loaders/coco/make_dataset.py:117 +
    the float32 TensorSpec :140), cv2.resize (ssd/make_dataset.py:40), (x - 0.5) * 2 (models/ssd_model.py:214)."""
    x = (img_u8.astype(np.float64) / 255.0).astype(np.float32)
    x = resize_bilinear(x, size)
    if normalize:
        x = ((x - np.float32(0.5)) * np.float32(2.0)).astype(np.float32)
    return x


def box_prep(box_tlwh, h, w):
    """COCO [x, y, w, h] pixels -> centre form (coco/make_dataset.py:132) divided by [w, h, w, h] (ssd/make_dataset.py:43-44),
    float32 as the TensorSpecs make it."""
    b = np.asarray(box_tlwh, np.float32).copy()
    b[:, :2] = b[:, :2] + b[:, 2:] / np.float32(2.0)
    return (b / np.array([w, h, w, h], np.float32)).astype(np.float32)
