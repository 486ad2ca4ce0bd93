"""Shared helpers for tests: golden-fixture access."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def golden_priors():
    return load("priors.npz")["priors"]


def dense_case(z, name, enc_zero):
    """Rebuild the reference's dense per-image outputs from the sparse fixture form
    (gen_golden.py asserts this reconstruction is exact)."""
    A = enc_zero.shape[0]
    idx = z[name + "_pos_idx"]
    cls = np.zeros((A,), np.int32)
    box = np.zeros((A, 4), np.float32)
    enc = enc_zero.copy()
    mask = np.zeros((A,), bool)
    cls[idx] = z[name + "_pos_cls"]
    box[idx] = z[name + "_pos_box"]
    enc[idx] = z[name + "_pos_enc"]
    mask[idx] = True
    return dict(gt_cls=z[name + "_gt_cls"], gt_box=z[name + "_gt_box"], thresh=float(z[name + "_thresh"]),
                cls=cls, box=box, enc=enc, mask=mask)


def all_match_cases():
    """Yield (name, case-dict) for every golden matching fixture (synthetic + edge)."""
    synth = load("match_synth.npz")
    enc_zero = synth["enc_zero"]
    for n in synth["names"]:
        yield str(n), dense_case(synth, str(n), enc_zero)
    edge = load("match_edge.npz")
    for n in edge["names"]:
        yield str(n), dense_case(edge, str(n), enc_zero)
