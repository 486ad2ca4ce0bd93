"""CPU suite: the C-ABI library builds, loads and exports every symbol include/ssd_hip.h declares
(no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "ssd_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(ssd_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    from ssd_object_detection_amd import _lib
    L = _lib.lib()
    names = declared_functions()
    assert len(names) >= 8
    for n in names:
        assert hasattr(L, n), "missing export: " + n
    assert set(_lib._SIGNATURES) == set(names), set(_lib._SIGNATURES) ^ set(names)
    assert L.ssd_hip_abi_version() >= 1
    assert L.ssd_status_string(0) == b"ok"


def test_host_side_argument_checks():
    """Entry points validate on the host before any launch: callable without a GPU."""
    from ssd_object_detection_amd import _lib
    L = _lib.lib()
    hw = (ctypes.c_int * 12)(38, 38, 19, 19, 10, 10, 5, 5, 3, 3, 1, 1)
    roff = (ctypes.c_int * 7)(0, 1, 3, 5, 7, 8, 9)
    assert L.ssd_priors_count(hw, 6, roff) == 8732                     # models/ssd_model.py:221
    assert L.ssd_priors_count(hw, 0, roff) == _lib.SSD_ERR_VALUE
    assert L.ssd_match_encode_workspace_bytes(64, 8732, 500) > 0
    # thresh <= 0 and n_t > A are the reference's asserts (utils/bbox.py:50-51)
    args = [None, None, None, 1, 0, 0, None, None, 8732, None]
    assert L.ssd_match_encode(*args, 0.0, None, None, None, None, None, 0, None) == _lib.SSD_ERR_ASSERT
    args[5] = 9000
    assert L.ssd_match_encode(*args, 0.5, None, None, None, None, None, 0, None) == _lib.SSD_ERR_ASSERT


def test_synthetic_generator_matches_fixture_inputs():
    from ssd_object_detection_amd.data_loaders.synthetic import synth_gt
    z = np.load(os.path.join(ROOT, "tests", "golden", "match_synth.npz"))
    for i in range(24):
        cls, box = synth_gt(i)
        assert np.array_equal(cls, z["mix%02d_gt_cls" % i]) and np.array_equal(box, z["mix%02d_gt_box" % i])
    cls, box = synth_gt(1000 + 14, 93)
    assert np.array_equal(box, z["fix14_gt_box"])
