"""GPU parity: HIP anchor matching / encoding / priors / IoU through the C ABI vs the golden vectors
captured from the reference and vs the oracle.  Bit-exact for indices, classes, masks, IoU bits
and the division terms of the encoding; <= 1 float32 ulp for the log terms (device log vs numpy)."""
import hashlib

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import ssd_oracle as O                                   # noqa: E402
from tests.helpers import load, golden_priors, all_match_cases      # noqa: E402


@pytest.fixture(scope="module")
def ops():
    import ssd_object_detection_amd.ops as ops_
    return ops_


@pytest.fixture(scope="module")
def pset(ops):
    return ops.build_priors()


def ulp_diff_f32(a, b):
    a = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, np.int64(-2**31) - a, a)
    b = np.where(b < 0, np.int64(-2**31) - b, b)
    return np.abs(a - b)


def check_image(name, case, cls, loc, mask, owner):
    assert np.array_equal(mask.astype(bool), case["mask"]), name
    assert np.array_equal(cls, case["cls"]), name
    pos = case["mask"]
    # anchor -> gt row map: every positive must point at a row holding exactly the reference's box
    assert (owner[~pos] == -1).all(), name
    got_box = np.zeros_like(case["box"])
    got_box[pos] = case["gt_box"][owner[pos]]
    assert np.array_equal(got_box.view(np.uint32), case["box"].view(np.uint32)), name
    # encoding: xy (IEEE division) bit-exact, wh (log) within 1 ulp
    assert np.array_equal(loc[:, :2].view(np.uint32), case["enc"][:, :2].view(np.uint32)), name
    assert ulp_diff_f32(loc[:, 2:], case["enc"][:, 2:]).max() <= 1, name


def test_priors_bit_exact(pset):
    p = pset.priors.cpu().numpy()
    g = golden_priors()
    assert np.array_equal(p.view(np.uint64), g.view(np.uint64))
    assert hashlib.sha256(p.tobytes()).hexdigest()[:16] == "ee36650176f74738"
    enc0 = load("match_synth.npz")["enc_zero"]
    e = pset.enc_zero.cpu().numpy()
    assert np.array_equal(e[:, :2].view(np.uint32), enc0[:, :2].view(np.uint32))
    assert ulp_diff_f32(e[:, 2:], enc0[:, 2:]).max() <= 1


def test_iou_n_bit_exact(ops):
    z = load("iou_n.npz")
    out = ops.iou_n(torch.from_numpy(z["b1"]).cuda(), torch.from_numpy(z["b2"]).cuda()).cpu().numpy()
    assert np.array_equal(out.view(np.uint64), z["mixed"].view(np.uint64))


@pytest.mark.parametrize("use_hint", [True, False])
def test_golden_cases_one_ragged_batch(ops, pset, use_hint):
    """All golden images with thresh 0.5 in ONE ragged batch (n_t from 1 to 93, duplicates, zero-area,
    far-outside, 0.5-straddling pairs ...)."""
    cases = [(n, c) for n, c in all_match_cases() if c["thresh"] == 0.5]
    gt_box, gt_cls, gt_off, total, max_nt = ops.pack_gt([c["gt_box"] for _, c in cases],
                                                        [c["gt_cls"] for _, c in cases])
    ps = pset if use_hint else ops.PriorSet(pset.priors, pset.enc_zero, None)
    owner = torch.empty((len(cases), pset.A), dtype=torch.int32, device="cuda")
    cls, loc, mask = ops.match_encode(gt_box, gt_cls, gt_off, total, max_nt, ps, 0.5, owner=owner)
    cls, loc, mask, owner = cls.cpu().numpy(), loc.cpu().numpy(), mask.cpu().numpy(), owner.cpu().numpy()
    for i, (name, case) in enumerate(cases):
        check_image(name, case, cls[i], loc[i], mask[i], owner[i])


def test_golden_other_thresholds(ops, pset):
    for name, case in all_match_cases():
        if case["thresh"] == 0.5:
            continue
        gt_box, gt_cls, gt_off, total, max_nt = ops.pack_gt([case["gt_box"]], [case["gt_cls"]])
        owner = torch.empty((1, pset.A), dtype=torch.int32, device="cuda")
        cls, loc, mask = ops.match_encode(gt_box, gt_cls, gt_off, total, max_nt, pset, case["thresh"], owner=owner)
        check_image(name, case, cls[0].cpu().numpy(), loc[0].cpu().numpy(), mask[0].cpu().numpy(),
                    owner[0].cpu().numpy())


def test_wrong_hint_is_harmless(ops, pset):
    """The geometry hint only seeds a pruning bound: a wrong one must not change any output."""
    from ssd_object_detection_amd import _lib
    bad = _lib.PriorGrid()
    bad.levels = 3
    for i, (h, w, k) in enumerate([(7, 5, 3), (40, 40, 5), (2, 9, 1)]):
        bad.grid_h[i], bad.grid_w[i], bad.per_cell[i] = h, w, k
    cases = [(n, c) for n, c in all_match_cases() if c["thresh"] == 0.5][:24]
    gt_box, gt_cls, gt_off, total, max_nt = ops.pack_gt([c["gt_box"] for _, c in cases],
                                                        [c["gt_cls"] for _, c in cases])
    ps = ops.PriorSet(pset.priors, pset.enc_zero, bad)
    assert ps.verify_grid() is False                             # a wrong geometry is never trusted for more than a bound
    owner = torch.empty((len(cases), pset.A), dtype=torch.int32, device="cuda")
    cls, loc, mask = ops.match_encode(gt_box, gt_cls, gt_off, total, max_nt, ps, 0.5, owner=owner)
    cls, loc, mask, owner = cls.cpu().numpy(), loc.cpu().numpy(), mask.cpu().numpy(), owner.cpu().numpy()
    for i, (name, case) in enumerate(cases):
        check_image(name, case, cls[i], loc[i], mask[i], owner[i])


def test_reference_own_cases_through_dropin():
    """tests/utils/test_bbox.py:25-45 of the reference, through the drop-in utils.bbox.match_bbox."""
    from ssd_object_detection_amd.utils.bbox import match_bbox
    z = load("ref_test_cases.npz")
    for tag in ["smoke", "a", "b"]:
        gt = z[tag + "_gt"]
        c, b, m = match_bbox(gt[:, 0], gt[:, 1:], z[tag + "_priors"])
        assert c.dtype == np.int32 and b.dtype == np.float32 and m.dtype == np.bool_
        assert np.array_equal(c, z[tag + "_cls"]) and np.array_equal(m, z[tag + "_mask"])
        assert np.array_equal(b.view(np.uint32), z[tag + "_box"].view(np.uint32))
    gt = z["a_gt"]
    _, loc, _ = match_bbox(gt[:, 0], gt[:, 1:], z["a_priors"])
    np.testing.assert_almost_equal(loc, gt[:, 1:])
    gt = z["b_gt"]
    _, loc, _ = match_bbox(gt[:, 0], gt[:, 1:], z["b_priors"])
    np.testing.assert_almost_equal(loc, np.array([[15, 15, 14, 14], [15, 15, 13, 13], [0, 0, 0, 0]]))


def test_dropin_asserts_and_apply_anchor_box():
    from ssd_object_detection_amd.utils.bbox import match_bbox, apply_anchor_box, iou_n
    pri = golden_priors()
    with pytest.raises(AssertionError):
        match_bbox(np.zeros(3, np.float32), np.zeros((3, 4), np.float32), pri[:2])
    with pytest.raises(AssertionError):
        match_bbox(np.zeros(1, np.float32), np.full((1, 4), 0.5, np.float32), pri, 0.0)
    with pytest.raises(AssertionError):
        apply_anchor_box(np.zeros((3, 4), np.float32), pri[:2])
    rng = np.random.default_rng(5)
    box = rng.uniform(0.01, 1.0, (8732, 4)).astype(np.float32)
    box[::7, 2:] = 1e-7                                    # exercise the 1e-5 clamp
    got = apply_anchor_box(box, pri)
    want = O.encode(box, pri)
    assert got.dtype == np.float64
    assert np.array_equal(got[:, :2].view(np.uint64), want[:, :2].view(np.uint64))
    np.testing.assert_allclose(got[:, 2:], want[:, 2:], rtol=4e-16, atol=0)
    z = load("iou_n.npz")
    assert np.array_equal(iou_n(z["b1"], z["b2"]).view(np.uint64), z["mixed"].view(np.uint64))


def test_full_batch_vs_oracle(ops, pset):
    """BASELINE batch size (64 synthetic COCO-shaped images) against the oracle's closed form."""
    from ssd_object_detection_amd.data_loaders.synthetic import synth_batch_gt
    pri = golden_priors()
    cls_l, box_l = synth_batch_gt(5000, 64)
    gt_box, gt_cls, gt_off, total, max_nt = ops.pack_gt(box_l, cls_l)
    owner = torch.empty((64, pset.A), dtype=torch.int32, device="cuda")
    cls, loc, mask = ops.match_encode(gt_box, gt_cls, gt_off, total, max_nt, pset, 0.5, owner=owner)
    cls, loc, mask, owner = cls.cpu().numpy(), loc.cpu().numpy(), mask.cpu().numpy(), owner.cpu().numpy()
    for i in range(64):
        c, b, m = O.match_closed_form(cls_l[i], box_l[i], pri, 0.5)
        e = O.encode(b, pri).astype(np.float32)
        case = dict(mask=m, cls=c, box=b, enc=e, gt_box=box_l[i])
        check_image("img%d" % i, case, cls[i], loc[i], mask[i], owner[i])
        # size-independent properties: every gt owns at least one anchor; positives >= n_t
        assert set(range(len(cls_l[i]))) <= set(owner[i][owner[i] >= 0].tolist())


def test_conflict_heavy_and_empty_images(ops, pset):
    """Many identical / near-identical gts force the literal round-by-round phase 1; an image with no
    gt yields an all-negative row."""
    pri = golden_priors()
    rng = np.random.default_rng(11)
    imgs = []
    base = np.array([0.52, 0.47, 0.21, 0.33], np.float32)
    imgs.append(np.repeat(base[None], 20, 0))                                   # 20 duplicates
    jit = base[None] + rng.normal(0, 1e-3, (30, 4)).astype(np.float32)          # 30 near-duplicates
    imgs.append(jit.astype(np.float32))
    imgs.append(np.zeros((0, 4), np.float32))                                   # empty image
    imgs.append(np.concatenate([np.repeat(pri[4000][None].astype(np.float32), 5, 0),
                                np.repeat(pri[8700][None].astype(np.float32), 7, 0)], 0))
    cls_l = [np.arange(len(b), dtype=np.float32) % 80 for b in imgs]
    gt_box, gt_cls, gt_off, total, max_nt = ops.pack_gt(imgs, cls_l)
    owner = torch.empty((len(imgs), pset.A), dtype=torch.int32, device="cuda")
    cls, loc, mask = ops.match_encode(gt_box, gt_cls, gt_off, total, max_nt, pset, 0.5, owner=owner)
    cls, loc, mask, owner = cls.cpu().numpy(), loc.cpu().numpy(), mask.cpu().numpy(), owner.cpu().numpy()
    for i, b in enumerate(imgs):
        if len(b) == 0:
            assert mask[i].sum() == 0 and (cls[i] == 0).all() and (owner[i] == -1).all()
            continue
        c, bx, m = O.match_literal(cls_l[i], b, pri, 0.5)
        e = O.encode(bx, pri).astype(np.float32)
        check_image("conf%d" % i, dict(mask=m, cls=c, box=bx, enc=e, gt_box=b), cls[i], loc[i], mask[i], owner[i])


def test_error_codes(ops, pset):
    gt_box, gt_cls, gt_off, total, max_nt = ops.pack_gt([np.full((2, 4), 0.5, np.float32)], [np.zeros(2, np.float32)])
    with pytest.raises(AssertionError):
        ops.match_encode(gt_box, gt_cls, gt_off, total, max_nt, pset, thresh=0.0)
    with pytest.raises(AssertionError):
        ops.match_encode(gt_box, gt_cls, gt_off, total, pset.A + 1, pset, thresh=0.5)


def test_single_launch_path_equals_three_launch_path(ops, pset):
    """The single-launch path (every workgroup resolves its image's phase 1 itself from geometric windows, DESIGN.md
    section 5) against the three-launch path (lists in memory, separate phase-1 pass), bit for bit: batches of up to 700
    images, COCO-shaped and heavy box counts up to the path's limit of 64 per image, crowded boxes that force the literal
    phase-1 order, and the grid's verification flag deciding the path."""
    from ssd_object_detection_amd import _lib
    from ssd_object_detection_amd.data_loaders.synthetic import synth_batch_gt
    from tests.test_cfg5_gpu import _conflict_heavy
    L = _lib.lib()
    assert pset.verify_grid() is True                          # explicit: the default construction does not verify
    rng = np.random.default_rng(12)
    try:
        for it, (B, nt) in enumerate([(64, None), (256, None), (700, None), (64, 64), (33, 40), (9, 57), (512, None), (128, 17), (40, -1)]):
            if nt == -1:                                       # crowded boxes: shared best priors, exhausted candidate chains
                box_l = [_conflict_heavy(rng, int(rng.integers(2, 64))) for _ in range(B)]
                cls_l = [rng.integers(0, 80, len(b)).astype(np.float32) for b in box_l]
            else:
                cls_l, box_l = synth_batch_gt(int(rng.integers(0, 5000)), B, nt)
                cls_l, box_l = [c[:64] for c in cls_l], [b[:64] for b in box_l]
            gt = ops.pack_gt(box_l, cls_l)
            outs = {}
            for path in (1, 0):
                L.ssd_dev_knob(b"SSD_MATCH_FUSED", path)
                owner = torch.full((B, pset.A), -9, dtype=torch.int32, device="cuda")
                got = ops.match_encode(*gt, pset, 0.5, owner=owner)
                outs[path] = [t.clone() for t in got] + [owner]
            for name, a, b in zip(("cls", "loc", "mask", "owner"), outs[0], outs[1]):
                assert torch.equal(a, b), (it, B, nt, name, int((a != b).sum()))
    finally:
        L.ssd_dev_knob(b"SSD_MATCH_FUSED", 0)


def test_grid_verification(ops, pset):
    """ssd_prior_grid_verify accepts exactly the geometry the priors were generated with: a wrong grid, a permuted prior
    array or a perturbed prior are refused (and then the hint can only seed a pruning bound)."""
    from ssd_object_detection_amd import _lib
    assert pset.verify_grid() is True
    bad = _lib.PriorGrid()
    bad.levels = 3
    for i, (h, w, k) in enumerate([(7, 5, 3), (40, 40, 5), (2, 9, 1)]):
        bad.grid_h[i], bad.grid_w[i], bad.per_cell[i] = h, w, k
    assert ops.PriorSet(pset.priors, pset.enc_zero, bad).verify_grid() is False
    pri = pset.priors.clone()
    pri[4000, 0] = pri[4000, 0] * (1 + 2.0 ** -52)             # one ulp off the cell centre
    ps = ops.PriorSet(pri, pset.enc_zero, ops.make_grid(ops.SSD300_GRIDS, ops.SSD300_RATIOS))
    assert ps.verify_grid() is False and ps.grid.verified == 0
    pri = pset.priors.clone()
    pri[[10, 11]] = pri[[11, 10]]                                # two anchor types of one cell swapped
    assert ops.PriorSet(pri, pset.enc_zero, ops.make_grid(ops.SSD300_GRIDS, ops.SSD300_RATIOS)).verify_grid() is False
