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

    def verify_grid(self):
        """Check the geometry against the prior array on the device (one host sync, at build time): a verified grid lets
        ssd_match_encode take its single-launch path.  Returns whether it was verified."""
        if self.grid is None:
            return False
        scratch = torch.zeros((1,), dtype=torch.int32, device=self.priors.device)
        _lib.check(_lib.lib().ssd_prior_grid_verify(_ptr(self.priors), self.A, ctypes.byref(self.grid), _ptr(scratch), _stream()))
        return self.grid.verified != 0


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


def prior_set_from(priors, grid=None, verify=False):
    """Wrap an existing device f64 [A,4] prior array (any geometry).  verify=True also checks `grid` against the array on the
    device (PriorSet.verify_grid: one launch + one host synchronisation) -- only the opt-in single-launch matcher needs a
    verified grid, so the default construction neither launches nor synchronises for it."""
    L = _lib.lib()
    priors = _dev(priors, torch.float64)
    A = priors.shape[0]
    enc0 = torch.empty((A, 4), dtype=torch.float32, device=priors.device)
    _lib.check(L.ssd_encode_zero(_ptr(priors), A, _ptr(enc0), _stream()))
    ps = PriorSet(priors, enc0, grid)
    if verify:
        ps.verify_grid()
    return ps


class MatchWorkspace:
    """Caller-owned scratch, grown on demand."""

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
    wsobj = ws or _loss_ws
    wbuf = wsobj.get(nbytes, conf.device)
    wsobj.clean_key = None                         # (this form leaves the histogram words dirty: see ssd_loss_heads)
    _lib.check(L.ssd_loss_fwd_bwd(_ptr(conf), _ptr(loc), dtype, _ptr(gt_cls), _ptr(gt_loc), _ptr(gt_mask), B, A, C,
                                  float(grad_scale), _ptr(out), _ptr(dconf), _ptr(dloc), _ptr(wbuf), wbuf.numel(),
                                  _stream()))
    return out, dconf, dloc


class HeadGradBuffers:
    """Device buffers behind an ssd_head_grads: per level the compact gradient rows [B*hw, npad] bf16, both index maps
    and the row counts.  `hw`, `per_cell`, `npad` are per-level lists; sum(hw*per_cell) == A."""

    def __init__(self, B, hw, per_cell, npad, device="cuda"):
        self.B, self.hw, self.per_cell, self.npad = B, list(hw), list(per_cell), list(npad)
        self.levels = len(self.hw)
        assert self.levels <= _lib.SSD_MAX_LEVELS
        self.rows = [torch.empty((B * h, p), dtype=torch.bfloat16, device=device) for h, p in zip(self.hw, self.npad)]
        self.row_of_pixel = [torch.empty((B * h,), dtype=torch.int32, device=device) for h in self.hw]
        self.pixel_of_row = [torch.empty((B * h,), dtype=torch.int32, device=device) for h in self.hw]
        self.count = torch.zeros((_lib.SSD_MAX_LEVELS,), dtype=torch.int32, device=device)
        c = _lib.HeadGrads()
        c.levels = self.levels
        for l in range(self.levels):
            c.hw[l], c.per_cell[l], c.npad[l] = self.hw[l], self.per_cell[l], self.npad[l]
            c.rows[l], c.row_of_pixel[l], c.pixel_of_row[l] = (self.rows[l].data_ptr(), self.row_of_pixel[l].data_ptr(),
                                                               self.pixel_of_row[l].data_ptr())
        c.count = self.count.data_ptr()
        self.c = c

    def dense(self, classes):
        """(dloc [B,A,4], dconf [B,A,classes]) scattered back from the rows (tests; one host sync)."""
        B = self.B
        counts = self.count.cpu().tolist()
        locs, confs = [], []
        for l in range(self.levels):
            h, n, p = self.hw[l], self.per_cell[l], self.npad[l]
            full = torch.zeros((B * h, p), dtype=torch.bfloat16, device=self.rows[l].device)
            k = counts[l]
            full[self.pixel_of_row[l][:k].long()] = self.rows[l][:k]
            locs.append(full[:, :n * 4].reshape(B, h * n, 4))
            confs.append(full[:, n * 4:n * (4 + classes)].reshape(B, h * n, classes))
        return torch.cat(locs, 1), torch.cat(confs, 1)


def ssd_loss_heads(conf, loc, gt_cls, gt_loc, gt_mask, hgb, grad_scale=1.0, ws=None):
    """ssd_loss_fwd_bwd_heads: the loss of ssd_loss with its gradient as compact per-level pixel rows in `hgb`
    (HeadGradBuffers).  Returns out8 (device f32[8])."""
    L = _lib.lib()
    B, A, C = conf.shape
    assert conf.dtype == loc.dtype == torch.bfloat16 and conf.is_cuda and conf.is_contiguous() and loc.is_contiguous()
    assert loc.shape == (B, A, 4) and gt_loc.shape == (B, A, 4)          # models/ssd_model.py:347-351
    assert gt_cls.shape == (B, A) and gt_mask.shape == (B, A) and hgb.B == B
    _dev(gt_cls, torch.int32); _dev(gt_loc, torch.float32); _dev(gt_mask, torch.uint8)
    out = torch.empty((8,), dtype=torch.float32, device=conf.device)
    nbytes = L.ssd_loss_heads_workspace_bytes(B, A, C)
    wsobj = ws or _loss_ws
    wbuf = wsobj.get(nbytes, conf.device)
    # ws_clean: a completed call leaves the histogram words of this layout zeroed (its last launch does it), so the next call on
    # the same buffer, same B * A and same stream order needs no memset node; anything else in between resets the claim
    key = (B * A, wbuf.data_ptr(), wbuf.numel())
    clean = getattr(wsobj, "clean_key", None) == key
    wsobj.clean_key = None
    _lib.check(L.ssd_loss_fwd_bwd_heads(_ptr(conf), _ptr(loc), 1, _ptr(gt_cls), _ptr(gt_loc), _ptr(gt_mask), B, A, C,
                                        float(grad_scale), _ptr(out), ctypes.byref(hgb.c), _ptr(wbuf), wbuf.numel(),
                                        1 if clean else 0, _stream()))
    wsobj.clean_key = key
    return out


def head_layers(x, w_tap, dx, dw, dbias, cout, relu_bits=None, relu_src=None):
    """ssd_head_layers from per-level lists of tensors (x bf16 [B,H,W,Cin]; w_tap bf16 [3,3,Cin,npad]; dx like x; dw f32
    [cout,3,3,Cin]; dbias f32 [cout] or None).  Returns (struct, keep-alive list)."""
    c = _lib.HeadLayers()
    c.levels = len(x)
    keep = []
    for l, xl in enumerate(x):
        _bf(xl); _bf(w_tap[l]); _bf(dx[l])
        _, H, W, Cin = xl.shape
        assert w_tap[l].shape[:3] == (3, 3, Cin) and dx[l].shape == xl.shape and dw[l].is_contiguous()
        c.H[l], c.W[l], c.Cin[l], c.cout[l] = H, W, Cin, cout[l]
        c.x[l], c.w_tap[l], c.dx[l], c.dw[l] = xl.data_ptr(), w_tap[l].data_ptr(), dx[l].data_ptr(), dw[l].data_ptr()
        c.dbias[l] = dbias[l].data_ptr() if dbias is not None and dbias[l] is not None else None
        rb = relu_bits[l] if relu_bits is not None else None
        rs = relu_src[l] if relu_src is not None else None
        c.relu_bits[l] = rb.data_ptr() if rb is not None else None
        c.relu_src[l] = rs.data_ptr() if rs is not None else None
        keep += [xl, w_tap[l], dx[l], dw[l], rb, rs]
    return c, keep


_heads_z_ws = MatchWorkspace()
_heads_w_ws = MatchWorkspace()


def heads_bwd_data_sparse(hgb, hl, ws=None, levels=None, prezeroed=False):
    """levels: None = every level, else the levels of this call (ssd_heads_bwd_data_sparse_levels; calls with disjoint sets may
    share `ws` on two streams).  prezeroed: the caller cleared those levels' dx maps; pixels without a gradient row are skipped."""
    L = _lib.lib()
    nbytes = L.ssd_heads_bwd_data_sparse_workspace_bytes(hgb.B, ctypes.byref(hl))
    wbuf = (ws or _heads_z_ws).get(nbytes, hgb.count.device)
    if levels is None and not prezeroed:
        _lib.check(L.ssd_heads_bwd_data_sparse(ctypes.byref(hgb.c), ctypes.byref(hl), hgb.B, _ptr(wbuf), wbuf.numel(), _stream()))
        return
    mask = 0
    for l in (levels if levels is not None else range(hgb.c.levels)):
        mask |= 1 << int(l)
    _lib.check(L.ssd_heads_bwd_data_sparse_levels(ctypes.byref(hgb.c), ctypes.byref(hl), hgb.B, mask, 1 if prezeroed else 0, _ptr(wbuf),
                                                  wbuf.numel(), _stream()))


def heads_bwd_weight_sparse(hgb, hl, ws=None):
    L = _lib.lib()
    nbytes = L.ssd_heads_bwd_weight_sparse_workspace_bytes(hgb.B, ctypes.byref(hgb.c), ctypes.byref(hl))
    wbuf = (ws or _heads_w_ws).get(nbytes, hgb.count.device)
    _lib.check(L.ssd_heads_bwd_weight_sparse(ctypes.byref(hgb.c), ctypes.byref(hl), hgb.B, _ptr(wbuf), wbuf.numel(), _stream()))


def weight_transpose_tap(w, cout_pad=None, out=None):
    """w bf16 [Cout,3,3,Cin] -> tap-major transposed copy [3,3,Cin,Cout_pad] (not flipped): the operand of the sparse head
    data gradient.  One-tensor call of ssd_weight_transpose_batched."""
    L = _lib.lib()
    _bf(w)
    Cout, k, _, Cin = w.shape
    cout_pad = cout_pad or (Cout + 7) // 8 * 8
    if out is None:
        out = torch.empty((k, k, Cin, cout_pad), dtype=torch.bfloat16, device=w.device)
    desc = torch.tensor([[w.data_ptr(), out.data_ptr(), Cout, k | 0x100, Cin, cout_pad]], dtype=torch.int64, device=w.device)
    tiles = ((cout_pad + 31) // 32) * ((Cin + 31) // 32) * k * k
    _lib.check(L.ssd_weight_transpose_batched(_ptr(desc), 1, tiles, _stream()))
    return out


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


# ------------------------------------------------------------------------------------------------
# convolution stack (NHWC bf16)
# ------------------------------------------------------------------------------------------------
_conv_ws = MatchWorkspace()


def same_pad(n, k, s):
    """TF 'SAME': out = ceil(n/s); pad_total = max((out-1)*s + k - n, 0); returns (out, pad_before)."""
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return out, total // 2


def valid_out(n, k, s):
    return (n - k) // s + 1


def _bf(t):
    assert t.is_cuda and t.dtype == torch.bfloat16 and t.is_contiguous(), (t.dtype, t.is_contiguous())
    return t


def image_prep(img, normalize=True, out=None):
    """f32 [B,H,W,3] in [0,1] -> bf16 [B,H,W,8] with (x-0.5)*2 (models/ssd_model.py:214), zero-padded channels."""
    L = _lib.lib()
    _dev(img, torch.float32)
    B, H, W, _ = img.shape
    if out is None:
        out = torch.empty((B, H, W, 8), dtype=torch.bfloat16, device=img.device)
    assert out.shape == (B, H, W, 8) and out.dtype == torch.bfloat16 and out.is_contiguous()
    _lib.check(L.ssd_image_prep(_ptr(img), _ptr(out), B, H, W, 1 if normalize else 0, _stream()))
    return out


def _splitk_ws(ws):
    buf = (ws or _conv_ws).get(1 << 25, torch.device("cuda", torch.cuda.current_device()))   # >= 32 MiB of scratch
    return buf


def conv2d_fwd(x, w, bias, stride, pad_t, pad_l, Ho, Wo, relu, out=None, ws=None):
    L = _lib.lib()
    _bf(x); _bf(w)
    B, H, W, Cin = x.shape
    Cout, k = w.shape[0], w.shape[1]
    assert w.shape == (Cout, k, k, Cin)
    if out is None:
        out = torch.empty((B, Ho, Wo, Cout), dtype=torch.bfloat16, device=x.device)
    wbuf = _splitk_ws(ws)
    _lib.check(L.ssd_conv2d_fwd(_ptr(x), _ptr(w), _ptr(bias), _ptr(out), B, H, W, Cin, Cout, k, stride, pad_t, pad_l,
                                Ho, Wo, 1 if relu else 0, _ptr(wbuf), wbuf.numel(), _stream()))
    return out


def conv2d_fwd_pool(x, w, bias, stride, pad_t, pad_l, Ho, Wo, relu, same, out=None, pool_out=None, code=None, ws=None,
                    pool_only=False):
    """conv2d_fwd followed by maxpool2x2_fwd_argmax of its output, fused on chip where the kernel allows.
    pool_only=True: the full-resolution map is not stored (returned as None); raises ValueError (SSD_ERR_VALUE) for a layer
    shape that no pooling kernel serves (nothing is launched then)."""
    L = _lib.lib()
    _bf(x); _bf(w)
    B, H, W, Cin = x.shape
    Cout, k = w.shape[0], w.shape[1]
    Hp, Wp = ((Ho + 1) // 2, (Wo + 1) // 2) if same else (Ho // 2, Wo // 2)
    if pool_only:
        out = None
    elif out is None:
        out = torch.empty((B, Ho, Wo, Cout), dtype=torch.bfloat16, device=x.device)
    if pool_out is None:
        pool_out = torch.empty((B, Hp, Wp, Cout), dtype=torch.bfloat16, device=x.device)
    if code is None:
        code = torch.empty((B, Hp, Wp, Cout // 8), dtype=torch.int32, device=x.device)
    wbuf = _splitk_ws(ws)
    _lib.check(L.ssd_conv2d_fwd_pool(_ptr(x), _ptr(w), _ptr(bias), _ptr(out), _ptr(pool_out), _ptr(code), B, H, W, Cin, Cout, k,
                                     stride, pad_t, pad_l, Ho, Wo, 1 if relu else 0, Hp, Wp, _ptr(wbuf), wbuf.numel(), _stream()))
    return out, pool_out, code


def conv2d_head_fwd(x, w, bias, loc, conf, per_cell, classes, level_off, ws=None):
    L = _lib.lib()
    _bf(x); _bf(w); _bf(loc); _bf(conf)
    B, H, W, Cin = x.shape
    A = loc.shape[1]
    assert w.shape == (per_cell * (4 + classes), 3, 3, Cin)
    wbuf = _splitk_ws(ws)
    _lib.check(L.ssd_conv2d_head_fwd(_ptr(x), _ptr(w), _ptr(bias), _ptr(loc), _ptr(conf), B, H, W, Cin, per_cell,
                                     classes, A, level_off, _ptr(wbuf), wbuf.numel(), _stream()))


def weight_transpose(w, cout_pad=None, out=None):
    L = _lib.lib()
    _bf(w)
    Cout, k, _, Cin = w.shape
    cout_pad = cout_pad or (Cout + 7) // 8 * 8
    if out is None:
        out = torch.empty((Cin, k, k, cout_pad), dtype=torch.bfloat16, device=w.device)
    _lib.check(L.ssd_weight_transpose(_ptr(w), _ptr(out), Cout, k, Cin, cout_pad, _stream()))
    return out


def conv2d_bwd_data(dy, w_t, relu_src, x_shape, stride, pad_t, pad_l, accumulate=False, out=None, ws=None):
    L = _lib.lib()
    _bf(dy); _bf(w_t)
    B, H, W, Cin = x_shape
    _, Ho, Wo, cpad = dy.shape
    k = w_t.shape[1]
    assert w_t.shape == (Cin, k, k, cpad)
    if out is None:
        assert not accumulate
        out = torch.empty(x_shape, dtype=torch.bfloat16, device=dy.device)
    wbuf = _splitk_ws(ws)
    _lib.check(L.ssd_conv2d_bwd_data(_ptr(dy), _ptr(w_t), _ptr(relu_src), _ptr(out), B, H, W, Cin, cpad, k, stride,
                                     pad_t, pad_l, Ho, Wo, 1 if accumulate else 0, _ptr(wbuf), wbuf.numel(), _stream()))
    return out


def chain_pack_weights(pairs):
    """ssd_chain_pack_weights: [(w, packed)] with w bf16 [N,k,k,C] (forward filters, or weight_transpose's copy for the data
    gradient) and packed a bf16 tensor of the same shape and element count that receives the fragment-packed copy conv_chain reads
    (None: allocated).  Returns the packed tensors.  One launch for up to 16 tensors."""
    L = _lib.lib()
    out = []
    for i0 in range(0, len(pairs), _lib.SSD_CHAIN_PACK_MAX):
        chunk = pairs[i0:i0 + _lib.SSD_CHAIN_PACK_MAX]
        arr = (_lib.ChainPack * len(chunk))()
        for d, (w, packed) in zip(arr, chunk):
            _bf(w)
            if packed is None:
                packed = torch.empty_like(w)
            _bf(packed)
            assert packed.numel() == w.numel()
            d.src, d.dst, d.N, d.K = _ptr(w), _ptr(packed), w.shape[0], w.numel() // w.shape[0]
            out.append(packed)
        _lib.check(L.ssd_chain_pack_weights(arr, len(chunk), _stream()))
    return out


def chain_prefetch(packed_list):
    """ssd_chain_prefetch: read the packed filter copies into every XCD's L2 (call on a second stream shortly before conv_chain)."""
    L = _lib.lib()
    arr = (_lib.ChainPack * len(packed_list))()
    for d, t in zip(arr, packed_list):
        _bf(t)
        d.src, d.dst, d.N, d.K = None, _ptr(t), t.shape[0], t.numel() // t.shape[0]
    _lib.check(L.ssd_chain_prefetch(arr, len(packed_list), _stream()))


def chain_layer_fwd(w, packed, bias, out, stride, pad_t, pad_l, relu=True, relu_bits=None):
    """One forward convolution of conv_chain: w bf16 [Cout,k,k,Cin] (shape only), packed = its chain_pack_weights copy (the
    operand), out bf16 [B,Ho,Wo,Cout]."""
    return dict(w=packed, bias=bias, out=out, ksize=w.shape[1], mul=stride, div=1, pad_t=pad_t, pad_l=pad_l, relu=relu,
                relu_bits=relu_bits, Kc=w.shape[3], N=w.shape[0])


def chain_layer_dgrad(w_t, packed, out, stride, pad_t, pad_l, accumulate=False, mask_bits=None, mask_src=None):
    """One data gradient of conv_chain: w_t bf16 [Cin,k,k,Cout_pad] (weight_transpose; shape only), packed = its
    chain_pack_weights copy, out bf16 [B,H,W,Cin] = the gradient w.r.t. the convolution's input (accumulated onto when
    `accumulate`), masked by the input activation's ReLU sign."""
    k = w_t.shape[1]
    return dict(w=packed, bias=None, out=out, ksize=k, mul=1, div=stride, pad_t=k - 1 - pad_t, pad_l=k - 1 - pad_l, relu=False,
                accumulate=accumulate, mask_bits=mask_bits, mask_src=mask_src, Kc=w_t.shape[3], N=w_t.shape[0])


def conv_chain(in0, layers):
    """ssd_conv_chain: layers (chain_layer_fwd / chain_layer_dgrad dicts) applied one after the other to in0 bf16 [B,H,W,C], one
    workgroup per image, in one launch; every layer's `out` is written.  NotImplementedError (SSD_ERR_UNSUPPORTED, nothing
    launched) where a layer does not fit the kernel."""
    L = _lib.lib()
    _bf(in0)
    B, Hi, Wi, Kc = in0.shape
    arr = (_lib.ChainLayer * len(layers))()
    for d, s in zip(arr, layers):
        out = s["out"]
        _bf(s["w"]); _bf(out)
        assert out.shape[0] == B and out.shape[3] == s["N"] and s["Kc"] == Kc, (tuple(out.shape), s["N"], s["Kc"], Kc)
        d.w, d.bias, d.out = _ptr(s["w"]), _ptr(s.get("bias")), _ptr(out)
        d.mask_bits, d.mask_src, d.relu_bits = _ptr(s.get("mask_bits")), _ptr(s.get("mask_src")), _ptr(s.get("relu_bits"))
        d.Hi, d.Wi, d.Kc, d.Ho, d.Wo, d.N = Hi, Wi, Kc, out.shape[1], out.shape[2], s["N"]
        d.ksize, d.mul, d.div, d.pad_t, d.pad_l = s["ksize"], s["mul"], s["div"], s["pad_t"], s["pad_l"]
        d.relu, d.accumulate = int(bool(s.get("relu"))), int(bool(s.get("accumulate")))
        Hi, Wi, Kc = out.shape[1], out.shape[2], s["N"]
    rc = L.ssd_conv_chain(_ptr(in0), arr, len(layers), B, _stream())
    if rc == _lib.SSD_ERR_UNSUPPORTED:
        raise NotImplementedError("ssd_conv_chain does not serve these layers")
    _lib.check(rc)


def conv2d_fwd_relubits(x, w, bias, stride, pad_t, pad_l, Ho, Wo, bits, out=None, ws=None):
    """conv2d_fwd with ReLU that also writes the sign bits of its output (uint8 [B,Ho,Wo,Cout/8]).  NotImplementedError
    (SSD_ERR_UNSUPPORTED, nothing launched) where the layer's kernel has no staged store."""
    L = _lib.lib()
    _bf(x); _bf(w)
    B, H, W, Cin = x.shape
    Cout, k = w.shape[0], w.shape[1]
    assert bits.dtype == torch.uint8 and bits.shape == (B, Ho, Wo, Cout // 8) and bits.is_contiguous()
    if out is None:
        out = torch.empty((B, Ho, Wo, Cout), dtype=torch.bfloat16, device=x.device)
    wbuf = _splitk_ws(ws)
    rc = L.ssd_conv2d_fwd_relubits(_ptr(x), _ptr(w), _ptr(bias), _ptr(out), _ptr(bits), B, H, W, Cin, Cout, k, stride, pad_t, pad_l,
                                   Ho, Wo, _ptr(wbuf), wbuf.numel(), _stream())
    if rc == _lib.SSD_ERR_UNSUPPORTED:
        raise NotImplementedError("this layer's kernel does not write sign bits")
    _lib.check(rc)
    return out


def conv2d_bwd_data_bits(dy, w_t, bits, x_shape, stride, pad_t, pad_l, accumulate=False, out=None, ws=None):
    """conv2d_bwd_data with the ReLU mask given as the sign bits of conv2d_fwd_relubits (same result, 16x fewer mask bytes)."""
    L = _lib.lib()
    _bf(dy); _bf(w_t)
    B, H, W, Cin = x_shape
    _, Ho, Wo, cpad = dy.shape
    k = w_t.shape[1]
    assert w_t.shape == (Cin, k, k, cpad) and bits.dtype == torch.uint8 and bits.shape == (B, H, W, Cin // 8)
    if out is None:
        assert not accumulate
        out = torch.empty(x_shape, dtype=torch.bfloat16, device=dy.device)
    wbuf = _splitk_ws(ws)
    rc = L.ssd_conv2d_bwd_data_bits(_ptr(dy), _ptr(w_t), _ptr(bits), _ptr(out), B, H, W, Cin, cpad, k, stride, pad_t, pad_l, Ho, Wo,
                                    1 if accumulate else 0, _ptr(wbuf), wbuf.numel(), _stream())
    if rc == _lib.SSD_ERR_UNSUPPORTED:
        raise NotImplementedError("this layer's kernel does not read sign bits")
    _lib.check(rc)
    return out


def conv2d_bwd_data_wgrad_first(dy, w_t, bits, image, dw=None, dbias=None, ws=None):
    """Data gradient of the second trunk layer (64 -> 64) fused with the weight gradient of the first (image -> 64): returns
    (dw0 [64,3,3,8] f32, dbias0 [64] f32); the gradient w.r.t. the first layer's output is consumed inside the kernel and never
    stored.  Same result as conv2d_bwd_data_bits + conv2d_bwd_weight up to fp32 summation order."""
    L = _lib.lib()
    _bf(dy); _bf(w_t); _bf(image)
    B, H, W, c = dy.shape
    assert c == 64 and w_t.shape == (64, 3, 3, 64) and image.shape == (B, H, W, 8)
    assert bits.dtype == torch.uint8 and bits.shape == (B, H, W, 8)
    if dw is None:
        dw = torch.empty((64, 3, 3, 8), dtype=torch.float32, device=dy.device)
    if dbias is None:
        dbias = torch.empty((64,), dtype=torch.float32, device=dy.device)
    assert dw.dtype == torch.float32 and dw.numel() == 64 * 72 and dw.is_contiguous() and dbias.dtype == torch.float32
    wbuf = (ws or _conv_ws).get(L.ssd_conv2d_bwd_data_wgrad_first_workspace_bytes(B, H, W), dy.device)
    rc = L.ssd_conv2d_bwd_data_wgrad_first(_ptr(dy), _ptr(w_t), _ptr(bits), _ptr(image), _ptr(dw), _ptr(dbias), B, H, W, _ptr(wbuf),
                                           wbuf.numel(), _stream())
    if rc == _lib.SSD_ERR_UNSUPPORTED:
        raise NotImplementedError("maps smaller than 16x16 take the two separate calls")
    _lib.check(rc)
    return dw, dbias


def conv2d_bwd_data_unpool(dy, w_t, relu_src, pool_code, full_shape, out=None, ws=None, pooled_out=None):
    """conv2d_bwd_data (3x3 / stride 1 / pad 1) w.r.t. a pooled map followed by maxpool2x2_bwd_argmax, in one launch: returns the
    gradient of the map BEFORE the pooling ([B,Hf,Wf,Cin]).  Raises NotImplementedError (SSD_ERR_UNSUPPORTED, nothing launched)
    when the layer is not served by an LDS-patch kernel: use the two calls then."""
    L = _lib.lib()
    _bf(dy); _bf(w_t)
    B, Hf, Wf, Cin = full_shape
    _, H, W, cpad = dy.shape
    assert w_t.shape == (Cin, 3, 3, cpad) and pool_code.shape == (B, H, W, Cin // 8) and pool_code.dtype == torch.int32
    if out is None:
        out = torch.empty(full_shape, dtype=torch.bfloat16, device=dy.device)
    wbuf = _splitk_ws(ws)
    if pooled_out is not None:                     # also keep the gradient of the pooled map (conv2d_bwd_weight_unpooled reads it)
        _bf(pooled_out)
        assert pooled_out.shape == (B, H, W, Cin)
    rc = L.ssd_conv2d_bwd_data_unpool(_ptr(dy), _ptr(w_t), _ptr(relu_src), _ptr(pool_code), _ptr(pooled_out), _ptr(out), B, H, W, Cin,
                                      cpad, Hf, Wf, _ptr(wbuf), wbuf.numel(), _stream())
    if rc == _lib.SSD_ERR_UNSUPPORTED:
        raise NotImplementedError("no LDS-patch kernel for this layer")
    _lib.check(rc)
    return out


def conv2d_bwd_weight(x, dy, Cout, k, stride, pad_t, pad_l, dw=None, dbias=None, want_bias=True, ws=None):
    L = _lib.lib()
    _bf(x); _bf(dy)
    B, H, W, Cin = x.shape
    _, Ho, Wo, ldy = dy.shape
    if dw is None:
        dw = torch.empty((Cout, k, k, Cin), dtype=torch.float32, device=x.device)
    if dbias is None and want_bias:
        dbias = torch.empty((Cout,), dtype=torch.float32, device=x.device)
    nbytes = L.ssd_conv2d_bwd_weight_workspace_bytes(B, Ho, Wo, Cin, Cout, ldy, k)
    wbuf = (ws or _conv_ws).get(nbytes, x.device)
    _lib.check(L.ssd_conv2d_bwd_weight(_ptr(x), _ptr(dy), _ptr(dw), _ptr(dbias), B, H, W, Cin, Cout, ldy, k, stride,
                                       pad_t, pad_l, Ho, Wo, _ptr(wbuf), wbuf.numel(), _stream()))
    return dw, dbias


def conv2d_bwd_weight_batched(layers, ws=None):
    """ssd_conv2d_bwd_weight_batched: layers = [(x, dy, Cout, k, stride, pad_t, pad_l, dw, dbias)] (arguments of conv2d_bwd_weight,
    dw / dbias given) in two launches; bit-identical to the separate calls.  NotImplementedError (SSD_ERR_UNSUPPORTED, nothing
    launched) when a layer is not one the generic small-layer kernel serves."""
    L = _lib.lib()
    arr = (_lib.WgradItem * len(layers))()
    for d, (x, dy, Cout, k, stride, pad_t, pad_l, dw, dbias) in zip(arr, layers):
        _bf(x); _bf(dy)
        B, H, W, Cin = x.shape
        _, Ho, Wo, ldy = dy.shape
        assert dw.dtype == torch.float32 and dw.shape == (Cout, k, k, Cin) and dw.is_contiguous()
        d.x, d.dy, d.dw, d.dbias = _ptr(x), _ptr(dy), _ptr(dw), _ptr(dbias)
        d.B, d.H, d.W, d.Cin, d.Cout, d.ldy, d.ksize, d.stride, d.pad_t, d.pad_l, d.Ho, d.Wo = B, H, W, Cin, Cout, ldy, k, stride, pad_t, pad_l, Ho, Wo
    nbytes = L.ssd_conv2d_bwd_weight_batched_workspace_bytes(arr, len(layers))
    wbuf = (ws or _conv_ws).get(nbytes, layers[0][0].device)
    rc = L.ssd_conv2d_bwd_weight_batched(arr, len(layers), _ptr(wbuf), wbuf.numel(), _stream())
    if rc == _lib.SSD_ERR_UNSUPPORTED:
        raise NotImplementedError("ssd_conv2d_bwd_weight_batched does not serve these layers")
    _lib.check(rc)


def conv2d_bwd_weight_unpooled(x, dpool, pool_code, dw=None, dbias=None, want_bias=True, ws=None):
    """Weight gradient of a 3x3 / stride 1 / pad 1 convolution whose output was 2x2 max-pooled, from the gradient of the pooled
    map and the winner codes (== conv2d_bwd_weight(x, maxpool2x2_bwd_argmax(pool_code, dpool, ...)) without the un-pooled zeros:
    structured-sparse MFMA).  Raises NotImplementedError (SSD_ERR_UNSUPPORTED, nothing launched) for shapes it does not serve."""
    L = _lib.lib()
    _bf(x); _bf(dpool)
    B, H, W, Cin = x.shape
    _, Hp, Wp, Cout = dpool.shape
    assert pool_code.shape == (B, Hp, Wp, Cout // 8) and pool_code.dtype == torch.int32
    if dw is None:
        dw = torch.empty((Cout, 3, 3, Cin), dtype=torch.float32, device=x.device)
    if dbias is None and want_bias:
        dbias = torch.empty((Cout,), dtype=torch.float32, device=x.device)
    nbytes = L.ssd_conv2d_bwd_weight_unpooled_workspace_bytes(B, H, W, Cin, Cout, Hp, Wp)
    if nbytes == 0:
        raise NotImplementedError("no structured-sparse weight-gradient kernel for this layer")
    wbuf = (ws or _conv_ws).get(nbytes, x.device)
    rc = L.ssd_conv2d_bwd_weight_unpooled(_ptr(x), _ptr(dpool), _ptr(pool_code), _ptr(dw), _ptr(dbias), B, H, W, Cin, Cout, Hp, Wp,
                                          _ptr(wbuf), wbuf.numel(), _stream())
    if rc == _lib.SSD_ERR_UNSUPPORTED:
        raise NotImplementedError("no structured-sparse weight-gradient kernel for this layer")
    _lib.check(rc)
    return dw, dbias


def image_resize_prep(src, src_off, src_hw, S=300, normalize=True, out=None):
    """Ragged batch of uint8 RGB images (src: flat u8 buffer, src_off i64 [B] byte offsets, src_hw i32 [B,2]) ->
    network input bf16 [B,S,S,8]: /255, cv2.resize INTER_LINEAR, (x-0.5)*2  (reference lines in include/ssd_hip.h)."""
    L = _lib.lib()
    _dev(src, torch.uint8); _dev(src_off, torch.int64); _dev(src_hw, torch.int32)
    B = src_hw.shape[0]
    if out is None:
        out = torch.empty((B, S, S, 8), dtype=torch.bfloat16, device=src.device)
    _lib.check(L.ssd_image_resize_prep(_ptr(src), _ptr(src_off), _ptr(src_hw), _ptr(out), B, S, 1 if normalize else 0, _stream()))
    return out


def box_prep(box_tlwh, gt_off, src_hw):
    """COCO [x,y,w,h] pixel boxes of a batch (concatenated) -> relative centre form, per image size."""
    L = _lib.lib()
    _dev(box_tlwh, torch.float32); _dev(gt_off, torch.int32); _dev(src_hw, torch.int32)
    total = box_tlwh.shape[0]
    out = torch.empty_like(box_tlwh)
    _lib.check(L.ssd_box_prep(_ptr(box_tlwh), _ptr(gt_off), _ptr(src_hw), _ptr(out), src_hw.shape[0], total, _stream()))
    return out


def maxpool2x2_fwd(x, same=False):
    L = _lib.lib()
    _bf(x)
    B, H, W, C = x.shape
    Ho, Wo = ((H + 1) // 2, (W + 1) // 2) if same else (H // 2, W // 2)
    y = torch.empty((B, Ho, Wo, C), dtype=torch.bfloat16, device=x.device)
    _lib.check(L.ssd_maxpool2x2_fwd(_ptr(x), _ptr(y), B, H, W, C, Ho, Wo, _stream()))
    return y


def maxpool2x2_fwd_argmax(x, same=False, out=None, code=None):
    """Pooling that also records the winner of every window (4-bit codes, u32 per 8 channels) for maxpool2x2_bwd_argmax."""
    L = _lib.lib()
    _bf(x)
    B, H, W, C = x.shape
    Ho, Wo = ((H + 1) // 2, (W + 1) // 2) if same else (H // 2, W // 2)
    if out is None:
        out = torch.empty((B, Ho, Wo, C), dtype=torch.bfloat16, device=x.device)
    if code is None:
        code = torch.empty((B, Ho, Wo, C // 8), dtype=torch.int32, device=x.device)
    _lib.check(L.ssd_maxpool2x2_fwd_argmax(_ptr(x), _ptr(out), _ptr(code), B, H, W, C, Ho, Wo, _stream()))
    return out, code


def maxpool2x2_bwd_argmax(code, dy, x_shape, out=None):
    L = _lib.lib()
    _bf(dy)
    B, H, W, C = x_shape
    _, Ho, Wo, _ = dy.shape
    if out is None:
        out = torch.empty(tuple(x_shape), dtype=torch.bfloat16, device=dy.device)
    _lib.check(L.ssd_maxpool2x2_bwd_argmax(_ptr(code), _ptr(dy), _ptr(out), B, H, W, C, Ho, Wo, _stream()))
    return out


def maxpool2x2_bwd(x, y, dy, out=None):
    L = _lib.lib()
    _bf(x); _bf(y); _bf(dy)
    B, H, W, C = x.shape
    _, Ho, Wo, _ = y.shape
    if out is None:
        out = torch.empty_like(x)
    _lib.check(L.ssd_maxpool2x2_bwd(_ptr(x), _ptr(y), _ptr(dy), _ptr(out), B, H, W, C, Ho, Wo, _stream()))
    return out


def quantize_mx_fp8(x):
    """bf16 tensor (last dimension a multiple of 32) -> (q uint8 same shape: OCP e4m3 bytes, scale uint8 [..., C/32]: E8M0)."""
    L = _lib.lib()
    _bf(x)
    assert x.shape[-1] % 32 == 0
    q = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    scale = torch.empty(x.shape[:-1] + (x.shape[-1] // 32,), dtype=torch.uint8, device=x.device)
    _lib.check(L.ssd_quantize_mx_fp8(_ptr(x), _ptr(q), _ptr(scale), x.numel(), _stream()))
    return q, scale


def dequantize_mx_fp8(q, scale):
    """The float32 values an MX-fp8 pair stands for (tests / reports; torch's float8_e4m3fn is the same OCP encoding)."""
    v = q.view(torch.float8_e4m3fn).float()
    s = torch.exp2(scale.float() - 127.0)
    return (v.view(*scale.shape, 32) * s.unsqueeze(-1)).view(q.shape)


def conv3x3_fwd_mxfp8(xq, xs, wq, ws, bias, relu=True, out=None):
    """3x3 / stride 1 / SAME forward on block-scaled fp8 operands (quantize_mx_fp8 of x [B,H,W,Cin] and w [Cout,3,3,Cin])."""
    L = _lib.lib()
    B, H, W, Cin = xq.shape
    Cout = wq.shape[0]
    assert wq.shape == (Cout, 3, 3, Cin) and xs.shape == (B, H, W, Cin // 32) and ws.shape == (Cout, 3, 3, Cin // 32)
    for t in (xq, xs, wq, ws):
        _dev(t, torch.uint8)
    if out is None:
        out = torch.empty((B, H, W, Cout), dtype=torch.bfloat16, device=xq.device)
    rc = L.ssd_conv3x3_fwd_mxfp8(_ptr(xq), _ptr(xs), _ptr(wq), _ptr(ws), _ptr(bias), _ptr(out), B, H, W, Cin, Cout, 1 if relu else 0,
                                 _stream())
    if rc == _lib.SSD_ERR_UNSUPPORTED:
        raise NotImplementedError("block-scaled fp8 forward needs Cin % 128 == 0")
    _lib.check(rc)
    return out


def add_relu_fwd(a, b, out=None):
    """out = relu(a + b): the residual add of a ResNet bottleneck (bf16, any equal shapes with a multiple of 8 elements)."""
    L = _lib.lib()
    _bf(a); _bf(b)
    assert a.shape == b.shape
    if out is None:
        out = torch.empty_like(a)
    _lib.check(L.ssd_add_relu_fwd(_ptr(a), _ptr(b), _ptr(out), a.numel(), _stream()))
    return out


def relu_mask_bwd(g, act, out=None, accumulate=False):
    """out (+)= g * (act > 0): the gradient of relu(a + b) w.r.t. an identity-skip input whose own activation is `act`."""
    L = _lib.lib()
    _bf(g); _bf(act)
    assert g.shape == act.shape
    if out is None:
        assert not accumulate
        out = torch.empty_like(g)
    _lib.check(L.ssd_relu_mask_bwd(_ptr(g), _ptr(act), _ptr(out), 1 if accumulate else 0, g.numel(), _stream()))
    return out


def maxpool3x3s2_fwd(x, out=None, code=None):
    """MaxPooling2D(3, strides=2, padding="same") (TF padding: the odd cell goes to the bottom / right); returns (y, code)."""
    L = _lib.lib()
    _bf(x)
    B, H, W, C = x.shape
    Ho, pt = same_pad(H, 3, 2)
    Wo, pl = same_pad(W, 3, 2)
    if out is None:
        out = torch.empty((B, Ho, Wo, C), dtype=torch.bfloat16, device=x.device)
    if code is None:
        code = torch.empty((B, Ho, Wo, C // 8), dtype=torch.int32, device=x.device)
    _lib.check(L.ssd_maxpool3x3s2_fwd(_ptr(x), _ptr(out), _ptr(code), B, H, W, C, Ho, Wo, pt, pl, _stream()))
    return out, code


def maxpool3x3s2_bwd(code, dy, x_shape, out=None):
    L = _lib.lib()
    _bf(dy)
    B, H, W, C = x_shape
    Ho, pt = same_pad(H, 3, 2)
    Wo, pl = same_pad(W, 3, 2)
    assert dy.shape == (B, Ho, Wo, C) and code.shape == (B, Ho, Wo, C // 8)
    if out is None:
        out = torch.empty(x_shape, dtype=torch.bfloat16, device=dy.device)
    _lib.check(L.ssd_maxpool3x3s2_bwd(_ptr(code), _ptr(dy), _ptr(out), B, H, W, C, Ho, Wo, pt, pl, _stream()))
    return out


def head_grad_pack(dloc, dconf, hw, per_cell, classes, npad, level_off, out=None):
    L = _lib.lib()
    _bf(dloc); _bf(dconf)
    B, A, _ = dloc.shape
    if out is None:
        out = torch.empty((B, hw, npad), dtype=torch.bfloat16, device=dloc.device)
    _lib.check(L.ssd_head_grad_pack(_ptr(dloc), _ptr(dconf), _ptr(out), B, hw, per_cell, classes, npad, A, level_off,
                                    _stream()))
    return out


def cast_bf16(src, dst=None):
    L = _lib.lib()
    _dev(src, torch.float32)
    if dst is None:
        dst = torch.empty(src.shape, dtype=torch.bfloat16, device=src.device)
    _lib.check(L.ssd_cast_bf16(_ptr(src), _ptr(dst), src.numel(), _stream()))
    return dst
