"""N1 oracle: properties of the CPU restatement of the input pipeline (cv2 itself is absent: unpinned)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ssd_oracle as O                                   # noqa: E402


def test_resize_identity_and_constant():
    rng = np.random.default_rng(0)
    img = rng.random((300, 300, 3)).astype(np.float32)
    assert np.array_equal(O.resize_bilinear(img, 300), img)           # scale 1: taps fall on the pixels
    const = np.full((37, 53, 3), 0.625, np.float32)
    assert np.array_equal(O.resize_bilinear(const, 300), np.full((300, 300, 3), 0.625, np.float32))


def test_resize_known_answers():
    # 2 -> 4 upscale of a ramp: cv2's half-pixel rule gives [0, 0.25, 0.75, 1] per axis
    img = np.array([[[0.0], [1.0]], [[0.0], [1.0]]], np.float32)
    out = O.resize_bilinear(img, 4)[..., 0]
    assert np.allclose(out, np.tile(np.array([0.0, 0.25, 0.75, 1.0], np.float32), (4, 1)))
    # 4 -> 2 downscale of [0,1,2,3]: taps at 0.5 and 2.5
    img = np.tile(np.arange(4, dtype=np.float32)[None, :, None], (4, 1, 1))
    out = O.resize_bilinear(img, 2)[..., 0]
    assert np.allclose(out, np.array([[0.5, 2.5], [0.5, 2.5]], np.float32))


def test_prep_range_and_boxes():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (123, 211, 3), dtype=np.uint8)
    x = O.image_resize_prep(img, 300)
    assert x.shape == (300, 300, 3) and x.dtype == np.float32 and x.min() >= -1.0 and x.max() <= 1.0
    b = O.box_prep(np.array([[10, 20, 30, 40]], np.float32), h=100, w=200)
    assert np.allclose(b, [[(10 + 15) / 200, (20 + 20) / 100, 30 / 200, 40 / 100]])
