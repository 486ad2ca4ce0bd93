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


def test_round4_entries_refuse_on_the_host():
    """ssd_conv_chain / ssd_chain_pack_weights / ssd_conv2d_bwd_weight_batched / ssd_heads_bwd_data_sparse_levels decide on the
    host whether they serve a call -- SSD_ERR_VALUE / SSD_ERR_UNSUPPORTED come back before anything touches a device."""
    from ssd_object_detection_amd import _lib
    L = _lib.lib()
    dummy = ctypes.c_void_p(0x1000)                                     # never dereferenced on these paths
    assert L.ssd_conv_chain(None, None, 0, 1, None) == _lib.SSD_ERR_VALUE
    lay = (_lib.ChainLayer * 2)()

    def fill(d, hi, kc, ho, n, k=1):
        d.w, d.out = dummy, dummy
        d.Hi, d.Wi, d.Kc, d.Ho, d.Wo, d.N, d.ksize, d.mul, d.div, d.pad_t, d.pad_l = hi, hi, kc, ho, ho, n, k, 1, 1, 0, 0

    fill(lay[0], 19, 256, 19, 128)                                      # 361 pixels: not an LDS-resident map
    assert L.ssd_conv_chain(dummy, lay, 1, 4, None) == _lib.SSD_ERR_UNSUPPORTED
    fill(lay[0], 4, 64, 4, 128)                                         # 64 input channels: not a multiple of 128
    assert L.ssd_conv_chain(dummy, lay, 1, 4, None) == _lib.SSD_ERR_UNSUPPORTED
    fill(lay[0], 4, 128, 4, 128)
    fill(lay[1], 4, 256, 4, 128)                                        # does not read what layer 0 writes
    assert L.ssd_conv_chain(dummy, lay, 2, 4, None) == _lib.SSD_ERR_VALUE
    assert L.ssd_conv_chain(dummy, lay, _lib.SSD_CHAIN_MAX_LAYERS + 1, 4, None) == _lib.SSD_ERR_VALUE
    packs = (_lib.ChainPack * 1)()
    assert L.ssd_chain_pack_weights(packs, 0, None) == _lib.SSD_ERR_VALUE
    packs[0].src, packs[0].dst, packs[0].N, packs[0].K = dummy, dummy, 24, 64          # N not a multiple of 16
    assert L.ssd_chain_pack_weights(packs, 1, None) == _lib.SSD_ERR_VALUE
    assert L.ssd_chain_prefetch(packs, _lib.SSD_CHAIN_PACK_MAX + 1, None) == _lib.SSD_ERR_VALUE
    items = (_lib.WgradItem * 9)()
    assert L.ssd_conv2d_bwd_weight_batched(None, 1, None, 0, None) == _lib.SSD_ERR_VALUE
    assert L.ssd_conv2d_bwd_weight_batched(items, 9, dummy, 1 << 30, None) == _lib.SSD_ERR_UNSUPPORTED    # more than 8 layers
    it = items[0]
    it.x, it.dy, it.dw = dummy, dummy, dummy
    it.B, it.H, it.W, it.Cin, it.Cout, it.ldy, it.ksize, it.stride, it.pad_t, it.pad_l, it.Ho, it.Wo = 64, 19, 19, 1024, 1024, 1024, 1, 1, 0, 0, 19, 19
    assert L.ssd_conv2d_bwd_weight_batched(items, 1, None, 0, None) == _lib.SSD_ERR_WORKSPACE
    need = L.ssd_conv2d_bwd_weight_batched_workspace_bytes(items, 1)
    assert need > 0
    assert L.ssd_conv2d_bwd_weight_batched(items, 1, dummy, need, None) == _lib.SSD_ERR_UNSUPPORTED       # the 256-wide tile kernel's layer
    assert L.ssd_heads_bwd_data_sparse_levels(None, None, 4, 3, 0, None, 0, None) != _lib.SSD_OK
    assert L.ssd_set_wgrad_reduce_stream(None) == _lib.SSD_OK


def test_synthetic_generator_matches_fixture_inputs():
    from ssd_object_detection_amd.data_loaders.synthetic import synth_gt
    z = np.load(os.path.join(ROOT, "tests", "golden", "match_synth.npz"))
    for i in range(24):
        cls, box = synth_gt(i)
        assert np.array_equal(cls, z["mix%02d_gt_cls" % i]) and np.array_equal(box, z["mix%02d_gt_box" % i])
    cls, box = synth_gt(1000 + 14, 93)
    assert np.array_equal(box, z["fix14_gt_box"])
