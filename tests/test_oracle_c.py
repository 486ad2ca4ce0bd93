"""CPU suite: the oracle's C restatement (oracle/ssd_oracle.c, the CPU-baseline code) against the golden vectors
captured from the reference and against the numpy oracle."""
import numpy as np
import pytest

from oracle import c_oracle as C
from oracle import ssd_oracle as O
from tests.helpers import load, golden_priors, all_match_cases

CASES = list(all_match_cases())


def ulp_f32(a, b):
    a = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, np.int64(-2 ** 31) - a, a)
    b = np.where(b < 0, np.int64(-2 ** 31) - b, b)
    return np.abs(a - b)


def test_c_iou_n_bits():
    z = load("iou_n.npz")
    assert np.array_equal(C.iou_n(z["b1"], z["b2"]).view(np.uint64), z["mixed"].view(np.uint64))


@pytest.mark.parametrize("name,case", CASES, ids=[n for n, _ in CASES])
def test_c_match_literal_golden(name, case):
    pri = golden_priors()
    cls, box, mask, enc, owner = C.match_encode(case["gt_cls"], case["gt_box"], pri, case["thresh"])
    assert np.array_equal(mask, case["mask"]) and np.array_equal(cls, case["cls"])
    assert np.array_equal(box.view(np.uint32), case["box"].view(np.uint32))
    assert np.array_equal(enc[:, :2].view(np.uint32), case["enc"][:, :2].view(np.uint32))
    assert ulp_f32(enc[:, 2:], case["enc"][:, 2:]).max() <= 1          # libm log vs numpy log
    assert (owner[~mask] == -1).all()


def test_c_nms_matches_numpy_oracle():
    rng = np.random.default_rng(4)
    A = 600
    score = rng.uniform(0.3, 1.0, A).astype(np.float32)
    score[::7] = score[3]                                              # exact ties
    cls = rng.integers(0, 5, A).astype(np.int32)
    box = np.concatenate([rng.uniform(50, 250, (A, 2)), rng.uniform(20, 90, (A, 2))], 1).astype(np.float32)
    cand = rng.uniform(size=A) < 0.8
    for mc in (50, 1000):
        assert np.array_equal(C.nms(score, cls, box, cand, 0.45, mc), O.nms(score, cls, box, cand, 0.45, mc))
