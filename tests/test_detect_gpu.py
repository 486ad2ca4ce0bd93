"""GPU parity: scoring + decode (float tolerance vs the f64 oracle of models/ssd_model.py:466-467,479-488)
and NMS (bit-exact keep masks vs oracle/ssd_oracle.py:nms on identical inputs)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import ssd_oracle as O                                   # noqa: E402


@pytest.fixture(scope="module")
def ops():
    import ssd_object_detection_amd.ops as ops_
    return ops_


@pytest.fixture(scope="module")
def pset(ops):
    return ops.build_priors()


def synth_logits(B, A, C, n_hot, seed, one_class=None):
    """Background-dominated logits with n_hot boosted (anchor, class) entries per image, clustered so that
    neighbouring priors fire on the same class (NMS has work to do)."""
    rng = np.random.default_rng(seed)
    conf = rng.normal(0, 1, (B, A, C)).astype(np.float32)
    conf[..., C - 1] += 4.0
    for b in range(B):
        centres = rng.integers(0, A - 40, max(1, n_hot // 8))
        for c0 in centres:
            k = one_class if one_class is not None else int(rng.integers(0, C - 1))
            idx = c0 + rng.integers(0, 40, 8)
            conf[b, idx, k] += rng.uniform(7, 11, 8).astype(np.float32)
    loc = rng.normal(0, 0.2, (B, A, 4)).astype(np.float32)
    return conf, loc


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_score_decode_vs_oracle(ops, pset, dtype):
    B, A, C = 3, 8732, 81
    conf_np, loc_np = synth_logits(B, A, C, 300, 1)
    conf = torch.from_numpy(conf_np).cuda().to(dtype)
    loc = torch.from_numpy(loc_np).cuda().to(dtype)
    score, cls, box, cand = ops.score_decode(conf, loc, pset, 0.3)
    cf, lf = conf.float().cpu().numpy(), loc.float().cpu().numpy()
    s_ref, c_ref, cand_ref = O.score(cf, 0.3)
    score, cls, box, cand = score.cpu().numpy(), cls.cpu().numpy(), box.cpu().numpy(), cand.cpu().numpy().astype(bool)
    np.testing.assert_allclose(score, s_ref, rtol=2e-6, atol=1e-9)
    p = np.exp(O._log_softmax(cf))
    border = (np.abs(s_ref - 0.3) < 1e-6) | (np.abs(p[..., -1] - 0.3) < 1e-6)
    assert np.array_equal(cand[~border], cand_ref[~border])
    assert cand.sum() > 100 * B
    # class: argmax ties only by exact equality of logits (none in random data)
    assert np.array_equal(cls[cand], c_ref[cand])
    pri = pset.priors.cpu().numpy()
    want = O.decode(lf, pri[None], 300)
    got = box[cand]
    np.testing.assert_allclose(got, want[cand], rtol=3e-7, atol=0)
    assert (box[~cand] == 0).all()


def run_nms_case(ops, score, cls, box, cand, iou_thresh, max_cand):
    keep, count = ops.nms(score, cls, box, cand, iou_thresh, max_cand, want_count=True)
    keep = keep.cpu().numpy().astype(bool)
    s, c, bx, cd = score.cpu().numpy(), cls.cpu().numpy(), box.cpu().numpy(), cand.cpu().numpy()
    for b in range(s.shape[0]):
        want = O.nms(s[b], c[b], bx[b], cd[b], iou_thresh, max_cand)
        assert np.array_equal(keep[b], want), "image %d: %d vs %d kept" % (b, keep[b].sum(), want.sum())
        assert int(count[b]) == int(want.sum())
    return keep


def test_nms_bit_exact_typical(ops, pset):
    B, A, C = 4, 8732, 81
    conf_np, loc_np = synth_logits(B, A, C, 320, 2)
    conf, loc = torch.from_numpy(conf_np).cuda(), torch.from_numpy(loc_np).cuda()
    score, cls, box, cand = ops.score_decode(conf, loc, pset, 0.3)
    n_c = cand.sum(1).cpu().numpy()
    assert (n_c > 150).all() and (n_c < 1024).all()
    keep = run_nms_case(ops, score, cls, box, cand, 0.45, 1024)
    assert (keep.sum(1) < n_c).all()                    # something was suppressed
    run_nms_case(ops, score, cls, box, cand, 0.1, 1024)
    run_nms_case(ops, score, cls, box, cand, 0.9, 1024)


def test_nms_max_cand_cut_with_ties(ops, pset):
    """More candidates than max_cand, with exactly tied scores straddling the cut: the cut keeps the
    lowest anchor indices among equals."""
    B, A, C = 2, 8732, 81
    conf_np, loc_np = synth_logits(B, A, C, 2400, 3)
    # exact score ties: copy one hot row over many anchors
    for b in range(B):
        hot = np.nonzero(conf_np[b].max(-1) > 8)[0]
        src = hot[0]
        conf_np[b, hot[5:400:3]] = conf_np[b, src]
    conf, loc = torch.from_numpy(conf_np).cuda(), torch.from_numpy(loc_np).cuda()
    score, cls, box, cand = ops.score_decode(conf, loc, pset, 0.3)
    assert (cand.sum(1).cpu().numpy() > 1100).all()
    for max_cand in (50, 200, 333, 1024):
        run_nms_case(ops, score, cls, box, cand, 0.45, max_cand)


def test_nms_single_class_and_empty(ops, pset):
    B, A, C = 3, 8732, 81
    conf_np, loc_np = synth_logits(B, A, C, 600, 4, one_class=17)
    conf_np[2] = 0.0
    conf_np[2, :, C - 1] = 6.0                            # image 2: no candidate at all
    conf, loc = torch.from_numpy(conf_np).cuda(), torch.from_numpy(loc_np).cuda()
    score, cls, box, cand = ops.score_decode(conf, loc, pset, 0.3)
    assert int(cand[2].sum()) == 0
    keep = run_nms_case(ops, score, cls, box, cand, 0.45, 1024)
    assert keep[2].sum() == 0


def test_nms_argument_checks(ops, pset):
    z = torch.zeros((1, 16), device="cuda")
    with pytest.raises(ValueError):
        ops.nms(z, z.int(), torch.zeros((1, 16, 4), device="cuda"), z.to(torch.uint8), 0.45, 4096)


def test_eval_pass_at_batch_64_vs_c_oracle(ops, pset):
    """BASELINE configs[2]'s "NMS eval pass" at its full batch (64 images, the bench's input construction): scores within the
    float tolerance of the f64 oracle, candidate sets equal away from the threshold, and the keep mask of EVERY image
    bit-exact against the C restatement of the build-defined NMS (oracle/ssd_oracle.c) on the device's own candidates."""
    from oracle import c_oracle
    B, A, C = 64, 8732, 81
    conf_np, loc_np = synth_logits(B, A, C, 320, 64)
    conf, loc = torch.from_numpy(conf_np).cuda(), torch.from_numpy(loc_np).cuda()
    score, cls, box, cand = ops.score_decode(conf, loc, pset, 0.3)
    keep, count = ops.nms(score, cls, box, cand, 0.45, 400, want_count=True)
    s, c, bx, cd = score.cpu().numpy(), cls.cpu().numpy(), box.cpu().numpy(), cand.cpu().numpy()
    keep, count = keep.cpu().numpy().astype(bool), count.cpu().numpy()
    s_ref, c_ref, cand_ref = O.score(conf_np, 0.3)
    np.testing.assert_allclose(s, s_ref, rtol=2e-6, atol=1e-9)
    p_bg = np.exp(O._log_softmax(conf_np))[..., -1]
    border = (np.abs(s_ref - 0.3) < 1e-6) | (np.abs(p_bg - 0.3) < 1e-6)
    assert np.array_equal(cd.astype(bool)[~border], cand_ref[~border])
    pri = pset.priors.cpu().numpy()
    want_box = O.decode(loc_np, pri[None], 300)
    np.testing.assert_allclose(bx[cd.astype(bool)], want_box[cd.astype(bool)], rtol=3e-7, atol=0)
    n_c = cd.sum(1)
    assert (n_c > 150).all() and (n_c < 1024).all()
    for b in range(B):
        want = c_oracle.nms(s[b], c[b], bx[b], cd[b], 0.45, 400)
        assert np.array_equal(keep[b], want), "image %d: %d vs %d kept" % (b, keep[b].sum(), want.sum())
        assert int(count[b]) == int(want.sum())
    assert (keep.sum(1) < np.minimum(n_c, 400)).any()        # suppression happened
