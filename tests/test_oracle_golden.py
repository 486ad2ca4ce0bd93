"""CPU suite: pin the oracle (oracle/ssd_oracle.py) against vectors captured from the reference
itself (tests/golden/, made by tests/golden/gen_golden.py) and against the reference's own
test cases (tests/utils/test_bbox.py in the reference)."""
import hashlib

import numpy as np
import pytest

from oracle import ssd_oracle as O
from tests.helpers import load, golden_priors, all_match_cases

CASES = list(all_match_cases())


def test_priors_bit_exact():
    p = O.priors()
    g = golden_priors()
    assert p.dtype == np.float64 and p.shape == (8732, 4)
    assert np.array_equal(p.view(np.uint64), g.view(np.uint64))
    assert hashlib.sha256(p.tobytes()).hexdigest()[:16] == "ee36650176f74738"
    # level offsets, SURVEY.md A2
    assert p[0].tolist() == [0.5 / 38, 0.5 / 38, 0.07, 0.07]
    assert p[-1].tolist() == [0.5, 0.5, 0.6151828996322963, 1.2303657992645927]


def test_iou_n_bit_patterns():
    z = load("iou_n.npz")
    b1, b2 = z["b1"], z["b2"]
    mixed = O.iou_n(b1, b2)
    assert mixed.dtype == np.float64
    assert np.array_equal(mixed.view(np.uint64), z["mixed"].view(np.uint64))
    assert np.array_equal(O.iou_n(b1.astype(np.float64), b2).view(np.uint64), z["f64"].view(np.uint64))
    f32 = O.iou_n(b1, b2.astype(np.float32))
    assert f32.dtype == np.float32
    assert np.array_equal(f32.view(np.uint32), z["f32"].view(np.uint32))


def test_scalar_iou_known_answers():
    # reference tests/utils/test_bbox.py:10-17, places=4
    kat = [([10, 10, 2, 2], [10, 10, 2, 2], 1.0), ([10, 10, 1, 1], [20, 20, 1, 1], 0.0),
           ([10, 10, 2, 2], [10, 10, 4, 4], 0.25), ([10, 10, 0, 0], [20, 20, 0, 0], 0.0),
           ([10, 10, -1, -1], [10, 10, -1, -1], 0.0), ([10, 10, 2, 2], [11, 11, 2, 2], 1 / 7),
           ([10, 10, 6, 6], [13, 13, 2, 2], 1 / 39), ([10, -10, 1, 1], [10, -10, 1, 1], 1.0)]
    for a, b, want in kat:
        assert abs(float(O.iou_scalar(a, b)) - want) < 5e-5


def test_reference_own_match_cases():
    z = load("ref_test_cases.npz")
    for tag in ["smoke", "a", "b"]:
        pri, gt = z[tag + "_priors"], z[tag + "_gt"]
        for fn in (O.match_literal, O.match_closed_form):
            c, b, m = fn(gt[:, 0], gt[:, 1:], pri)
            assert np.array_equal(c, z[tag + "_cls"]) and np.array_equal(m, z[tag + "_mask"])
            assert np.array_equal(b.view(np.uint32), z[tag + "_box"].view(np.uint32))
    # the two asserted expectations, restated (reference test_bbox.py:35-44)
    gt = z["a_gt"]
    _, b, _ = O.match_literal(gt[:, 0], gt[:, 1:], z["a_priors"])
    np.testing.assert_almost_equal(b, gt[:, 1:])
    gt = z["b_gt"]
    _, b, _ = O.match_literal(gt[:, 0], gt[:, 1:], z["b_priors"])
    np.testing.assert_almost_equal(b, np.array([[15, 15, 14, 14], [15, 15, 13, 13], [0, 0, 0, 0]]))


@pytest.mark.parametrize("name,case", CASES, ids=[n for n, _ in CASES])
def test_match_encode_bit_exact(name, case):
    pri = golden_priors()
    lit = len(case["gt_cls"]) <= 16          # literal algorithm is O(n_pos * n_t * A): keep CPU suite fast
    fns = [O.match_closed_form] + ([O.match_literal] if lit else [])
    for fn in fns:
        c, b, m = fn(case["gt_cls"], case["gt_box"], pri, case["thresh"])
        assert np.array_equal(m, case["mask"]), name
        assert np.array_equal(c, case["cls"]), name
        assert np.array_equal(b.view(np.uint32), case["box"].view(np.uint32)), name
    e = O.encode(b, pri).astype(np.float32)
    assert np.array_equal(e.view(np.uint32), case["enc"].view(np.uint32)), name


def test_literal_equals_closed_form_large():
    # one n_t=93 fixture through the literal path as well
    pri = golden_priors()
    case = dict(CASES)["fix14"]
    c, b, m = O.match_literal(case["gt_cls"], case["gt_box"], pri, case["thresh"])
    assert np.array_equal(m, case["mask"]) and np.array_equal(c, case["cls"])
    assert np.array_equal(b.view(np.uint32), case["box"].view(np.uint32))


def test_match_asserts():
    pri = golden_priors()[:2]
    with pytest.raises(AssertionError):
        O.match_literal(np.zeros(3, np.float32), np.zeros((3, 4), np.float32), pri)        # n_t > A
    with pytest.raises(AssertionError):
        O.match_literal(np.zeros(1, np.float32), np.zeros((1, 4), np.float32), pri, 0.0)   # thresh <= 0
    with pytest.raises(AssertionError):
        O.encode(np.zeros((3, 4), np.float32), pri)                                        # shape mismatch
