"""Synthetic COCO-shaped samples; there is no dataset or network here.

A sample has the contract of SSDDataLoader's iterables (reference data_loaders/ssd/make_dataset.py:
54-68): (image f32[300,300,3] RGB in [0,1], cls f32[n], box f32[n,4] (cx,cy,w,h) in [0,1])."""
import numpy as np


def synth_gt(image_index, n_t=None):
    """Ground truth of synthetic image `image_index`: n_t ~ clip(round(LogNormal(1.6, 0.8)), 1, 93)
    unless given; centre U(0.1,0.9); w,h log-uniform in [0.02,0.9] clipped to stay inside the image;
    class U{0..79} stored as float32.  Seed 4321 + image_index."""
    rng = np.random.default_rng(4321 + image_index)
    if n_t is None:
        n_t = int(np.clip(np.round(rng.lognormal(mean=1.6, sigma=0.8)), 1, 93))
    else:
        rng.lognormal(mean=1.6, sigma=0.8)
    cx = rng.uniform(0.1, 0.9, n_t)
    cy = rng.uniform(0.1, 0.9, n_t)
    w = np.exp(rng.uniform(np.log(0.02), np.log(0.9), n_t))
    h = np.exp(rng.uniform(np.log(0.02), np.log(0.9), n_t))
    w = np.minimum(w, 2.0 * np.minimum(cx, 1.0 - cx))
    h = np.minimum(h, 2.0 * np.minimum(cy, 1.0 - cy))
    cls = rng.integers(0, 80, n_t).astype(np.float32)
    box = np.stack([cx, cy, w, h], axis=1).astype(np.float32)
    return cls, box


def synth_image(image_index, size=300):
    """Uniform [0,1) RGB image, seed 1234 + image_index."""
    rng = np.random.default_rng(1234 + image_index)
    return rng.random((size, size, 3), dtype=np.float32)


def synth_batch_gt(first_index, batch, n_t=None):
    """Lists of (cls, box) for images first_index .. first_index+batch-1."""
    cls_list, box_list = [], []
    for i in range(batch):
        c, b = synth_gt(first_index + i, n_t)
        cls_list.append(c)
        box_list.append(b)
    return cls_list, box_list


def synth_raw_sample(image_index, n_t=None):
    """A sample as the COCO reader hands it over BEFORE the SSD loader's preprocessing (reference
    data_loaders/coco/make_dataset.py:108-134 without the /255): decoded uint8 RGB image of a COCO-like size
    (short side 240..480, long side <= 640), cls f32[n], box f32[n,4] = [x, y, w, h] top-left in pixels.
    Same ground truth as synth_gt(image_index) once preprocessed."""
    rng = np.random.default_rng(9876 + image_index)
    h = int(rng.integers(240, 481))
    w = int(rng.integers(240, 641))
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    cls, box = synth_gt(image_index, n_t)
    scale = np.array([w, h, w, h], np.float32)
    px = box * scale                                        # centre form in pixels
    tlwh = px.copy()
    tlwh[:, :2] = px[:, :2] - px[:, 2:] / np.float32(2.0)
    return img, cls, tlwh.astype(np.float32)
