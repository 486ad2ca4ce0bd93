"""ctypes binding of libssd_hip.so (include/ssd_hip.h).  Fails loudly when the library is absent."""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# (SSD_HIP_LIB: development only -- another build of the library next to the default one, for same-box A/B runs)
LIB_PATH = os.path.join(HERE, os.environ.get("SSD_HIP_LIB", "libssd_hip.so"))

SSD_OK, SSD_ERR_ASSERT, SSD_ERR_VALUE, SSD_ERR_WORKSPACE, SSD_ERR_LAUNCH, SSD_ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5
SSD_MAX_LEVELS = 8

_c_int_p = ctypes.POINTER(ctypes.c_int)
_c_double_p = ctypes.POINTER(ctypes.c_double)
VP = ctypes.c_void_p


class PriorGrid(ctypes.Structure):
    _fields_ = [("levels", ctypes.c_int),
                ("grid_h", ctypes.c_int * SSD_MAX_LEVELS),
                ("grid_w", ctypes.c_int * SSD_MAX_LEVELS),
                ("per_cell", ctypes.c_int * SSD_MAX_LEVELS),
                ("verified", ctypes.c_int)]


class HeadGrads(ctypes.Structure):
    """ssd_head_grads: the loss gradient as compact per-level pixel rows."""
    _fields_ = [("levels", ctypes.c_int),
                ("hw", ctypes.c_int * SSD_MAX_LEVELS),
                ("per_cell", ctypes.c_int * SSD_MAX_LEVELS),
                ("npad", ctypes.c_int * SSD_MAX_LEVELS),
                ("rows", VP * SSD_MAX_LEVELS),
                ("row_of_pixel", VP * SSD_MAX_LEVELS),
                ("pixel_of_row", VP * SSD_MAX_LEVELS),
                ("count", VP)]


class HeadLayers(ctypes.Structure):
    """ssd_head_layers: operands and outputs of the head convolutions' backward pass."""
    _fields_ = [("levels", ctypes.c_int),
                ("H", ctypes.c_int * SSD_MAX_LEVELS),
                ("W", ctypes.c_int * SSD_MAX_LEVELS),
                ("Cin", ctypes.c_int * SSD_MAX_LEVELS),
                ("cout", ctypes.c_int * SSD_MAX_LEVELS),
                ("x", VP * SSD_MAX_LEVELS),
                ("w_tap", VP * SSD_MAX_LEVELS),
                ("relu_bits", VP * SSD_MAX_LEVELS),
                ("relu_src", VP * SSD_MAX_LEVELS),
                ("dx", VP * SSD_MAX_LEVELS),
                ("dw", VP * SSD_MAX_LEVELS),
                ("dbias", VP * SSD_MAX_LEVELS)]


class ChainLayer(ctypes.Structure):
    """ssd_chain_layer: one convolution of ssd_conv_chain."""
    _fields_ = [("w", VP), ("bias", VP), ("out", VP), ("mask_bits", VP), ("mask_src", VP), ("relu_bits", VP)] + \
               [(n, ctypes.c_int) for n in ("Hi", "Wi", "Kc", "Ho", "Wo", "N", "ksize", "mul", "div", "pad_t", "pad_l", "relu",
                                            "accumulate")]


class WgradItem(ctypes.Structure):
    """ssd_wgrad_item: one layer of ssd_conv2d_bwd_weight_batched."""
    _fields_ = [("x", VP), ("dy", VP), ("dw", VP), ("dbias", VP)] + \
               [(n, ctypes.c_int) for n in ("B", "H", "W", "Cin", "Cout", "ldy", "ksize", "stride", "pad_t", "pad_l", "Ho", "Wo")]


class ChainPack(ctypes.Structure):
    """ssd_chain_pack: one filter tensor of ssd_chain_pack_weights."""
    _fields_ = [("src", VP), ("dst", VP), ("N", ctypes.c_int), ("K", ctypes.c_int)]


SSD_CHAIN_MAX_LAYERS = 8
SSD_CHAIN_PACK_MAX = 16
_HG = ctypes.POINTER(HeadGrads)
_HL = ctypes.POINTER(HeadLayers)

_SIGNATURES = {
    "ssd_hip_abi_version": (ctypes.c_int, []),
    "ssd_status_string": (ctypes.c_char_p, [ctypes.c_int]),
    "ssd_priors_count": (ctypes.c_int, [_c_int_p, ctypes.c_int, _c_int_p]),
    "ssd_priors": (ctypes.c_int, [_c_int_p, ctypes.c_int, _c_double_p, _c_int_p, _c_int_p, ctypes.c_double, VP, VP]),
    "ssd_encode_zero": (ctypes.c_int, [VP, ctypes.c_int, VP, VP]),
    "ssd_prior_grid_verify": (ctypes.c_int, [VP, ctypes.c_int, ctypes.POINTER(PriorGrid), VP, VP]),
    "ssd_match_encode_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "ssd_match_encode": (ctypes.c_int, [VP, VP, VP, ctypes.c_int, ctypes.c_int, ctypes.c_int, VP, VP, ctypes.c_int,
                                        ctypes.POINTER(PriorGrid), ctypes.c_double, VP, VP, VP, VP, VP, ctypes.c_size_t, VP]),
    "ssd_apply_anchor_box": (ctypes.c_int, [VP, VP, ctypes.c_int, VP, VP]),
    "ssd_iou_n": (ctypes.c_int, [VP, VP, ctypes.c_int, VP, VP]),
    "ssd_score_decode": (ctypes.c_int, [VP, VP, ctypes.c_int, VP, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_float, ctypes.c_double, VP, VP, VP, VP, VP]),
    "ssd_nms_max_candidates": (ctypes.c_int, []),
    "ssd_nms": (ctypes.c_int, [VP, VP, VP, VP, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, VP, VP, VP]),
    "ssd_conv2d_fwd": (ctypes.c_int, [VP, VP, VP, VP] + [ctypes.c_int] * 12 + [VP, ctypes.c_size_t, VP]),
    "ssd_conv2d_fwd_pool": (ctypes.c_int, [VP, VP, VP, VP, VP, VP] + [ctypes.c_int] * 14 + [VP, ctypes.c_size_t, VP]),
    "ssd_conv2d_head_fwd": (ctypes.c_int, [VP, VP, VP, VP, VP] + [ctypes.c_int] * 8 + [VP, ctypes.c_size_t, VP]),
    "ssd_conv2d_bwd_data": (ctypes.c_int, [VP, VP, VP, VP] + [ctypes.c_int] * 12 + [VP, ctypes.c_size_t, VP]),
    "ssd_conv2d_fwd_relubits": (ctypes.c_int, [VP] * 5 + [ctypes.c_int] * 11 + [VP, ctypes.c_size_t, VP]),
    "ssd_conv2d_bwd_data_bits": (ctypes.c_int, [VP] * 4 + [ctypes.c_int] * 12 + [VP, ctypes.c_size_t, VP]),
    "ssd_conv2d_bwd_data_unpool": (ctypes.c_int, [VP] * 6 + [ctypes.c_int] * 7 + [VP, ctypes.c_size_t, VP]),
    "ssd_quantize_mx_fp8": (ctypes.c_int, [VP, VP, VP, ctypes.c_longlong, VP]),
    "ssd_conv3x3_fwd_mxfp8": (ctypes.c_int, [VP] * 6 + [ctypes.c_int] * 6 + [VP]),
    "ssd_chain_pack_weights": (ctypes.c_int, [ctypes.POINTER(ChainPack), ctypes.c_int, VP]),
    "ssd_chain_prefetch": (ctypes.c_int, [ctypes.POINTER(ChainPack), ctypes.c_int, VP]),
    "ssd_set_wgrad_reduce_stream": (ctypes.c_int, [VP]),
    "ssd_conv2d_bwd_weight_batched_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(WgradItem), ctypes.c_int]),
    "ssd_conv2d_bwd_weight_batched": (ctypes.c_int, [ctypes.POINTER(WgradItem), ctypes.c_int, VP, ctypes.c_size_t, VP]),
    "ssd_conv_chain": (ctypes.c_int, [VP, ctypes.POINTER(ChainLayer), ctypes.c_int, ctypes.c_int, VP]),
    "ssd_add_relu_fwd": (ctypes.c_int, [VP, VP, VP, ctypes.c_longlong, VP]),
    "ssd_relu_mask_bwd": (ctypes.c_int, [VP, VP, VP, ctypes.c_int, ctypes.c_longlong, VP]),
    "ssd_maxpool3x3s2_fwd": (ctypes.c_int, [VP, VP, VP] + [ctypes.c_int] * 8 + [VP]),
    "ssd_maxpool3x3s2_bwd": (ctypes.c_int, [VP, VP, VP] + [ctypes.c_int] * 8 + [VP]),
    "ssd_conv2d_bwd_weight_unpooled_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 7),
    "ssd_conv2d_bwd_weight_unpooled": (ctypes.c_int, [VP] * 5 + [ctypes.c_int] * 7 + [VP, ctypes.c_size_t, VP]),
    "ssd_conv2d_bwd_data_wgrad_first_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 3),
    "ssd_conv2d_bwd_data_wgrad_first": (ctypes.c_int, [VP] * 6 + [ctypes.c_int] * 3 + [VP, ctypes.c_size_t, VP]),
    "ssd_conv2d_bwd_weight_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 7),
    "ssd_conv2d_bwd_weight": (ctypes.c_int, [VP, VP, VP, VP] + [ctypes.c_int] * 12 + [VP, ctypes.c_size_t, VP]),
    "ssd_weight_transpose": (ctypes.c_int, [VP, VP, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, VP]),
    "ssd_cast_bf16": (ctypes.c_int, [VP, VP, ctypes.c_longlong, VP]),
    "ssd_image_prep": (ctypes.c_int, [VP, VP, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, VP]),
    "ssd_image_resize_prep": (ctypes.c_int, [VP, VP, VP, VP, ctypes.c_int, ctypes.c_int, ctypes.c_int, VP]),
    "ssd_box_prep": (ctypes.c_int, [VP, VP, VP, VP, ctypes.c_int, ctypes.c_int, VP]),
    "ssd_maxpool2x2_fwd": (ctypes.c_int, [VP, VP] + [ctypes.c_int] * 6 + [VP]),
    "ssd_maxpool2x2_bwd": (ctypes.c_int, [VP, VP, VP, VP] + [ctypes.c_int] * 6 + [VP]),
    "ssd_weight_transpose_batched": (ctypes.c_int, [VP, ctypes.c_int, ctypes.c_int, VP]),
    "ssd_maxpool2x2_fwd_argmax": (ctypes.c_int, [VP, VP, VP] + [ctypes.c_int] * 6 + [VP]),
    "ssd_maxpool2x2_bwd_argmax": (ctypes.c_int, [VP, VP, VP] + [ctypes.c_int] * 6 + [VP]),
    "ssd_head_grad_pack": (ctypes.c_int, [VP, VP, VP] + [ctypes.c_int] * 7 + [VP]),
    "ssd_opt_block_elems": (ctypes.c_int, []),
    "ssd_grad_clip_scales": (ctypes.c_int, [VP, ctypes.c_longlong, VP, ctypes.c_int, ctypes.c_float, VP, VP, VP, VP]),
    "ssd_grad_apply_scale": (ctypes.c_int, [VP, ctypes.c_longlong, VP, VP, VP]),
    "ssd_grad_accumulate": (ctypes.c_int, [VP, VP, ctypes.c_longlong, VP, VP, ctypes.c_int, VP]),
    "ssd_adam_step": (ctypes.c_int, [VP, VP, VP, VP, VP, ctypes.c_longlong, VP, VP] + [ctypes.c_float] * 5 + [VP]),
    "ssd_sgd_step": (ctypes.c_int, [VP, VP, VP, ctypes.c_longlong, VP, VP, ctypes.c_float, ctypes.c_float, VP]),
    "ssd_dev_knob": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int]),
    "ssd_dev_mfma_calibration_workgroups": (ctypes.c_int, []),
    "ssd_dev_mfma_calibration_flops": (ctypes.c_double, [ctypes.c_int]),
    "ssd_dev_mfma_calibration": (ctypes.c_int, [ctypes.c_int, VP, VP, VP]),
    "ssd_conv2d_fwd_plan": (ctypes.c_int, [ctypes.c_int] * 12 + [ctypes.c_size_t]),
    "ssd_conv2d_head_fwd_plan": (ctypes.c_int, [ctypes.c_int] * 6 + [ctypes.c_size_t]),
    "ssd_conv2d_bwd_data_plan": (ctypes.c_int, [ctypes.c_int] * 12 + [ctypes.c_size_t]),
    "ssd_conv2d_bwd_weight_plan": (ctypes.c_int, [ctypes.c_int] * 12),
    "ssd_conv_plan_name": (ctypes.c_char_p, [ctypes.c_int]),
    "ssd_loss_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "ssd_loss_heads_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "ssd_loss_fwd_bwd_heads": (ctypes.c_int, [VP, VP, ctypes.c_int, VP, VP, VP, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_float, VP, _HG, VP, ctypes.c_size_t, ctypes.c_int, VP]),
    "ssd_heads_bwd_data_sparse_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, _HL]),
    "ssd_heads_bwd_data_sparse": (ctypes.c_int, [_HG, _HL, ctypes.c_int, VP, ctypes.c_size_t, VP]),
    "ssd_heads_bwd_data_sparse_levels": (ctypes.c_int, [_HG, _HL, ctypes.c_int, ctypes.c_uint, ctypes.c_int, VP, ctypes.c_size_t, VP]),
    "ssd_heads_bwd_weight_sparse_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, _HG, _HL]),
    "ssd_heads_bwd_weight_sparse": (ctypes.c_int, [_HG, _HL, ctypes.c_int, VP, ctypes.c_size_t, VP]),
    "ssd_loss_fwd_bwd": (ctypes.c_int, [VP, VP, ctypes.c_int, VP, VP, VP, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_float, VP, VP, VP, VP, ctypes.c_size_t, VP]),
}

_lib = None


def lib():
    """Load (once) and return the C-ABI library.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libssd_hip.so is missing (%s): build it with `python ssd-object-detection_amd/build.py` "
                "or __graft_entry__.build(); there is no CPU fallback." % LIB_PATH)
        try:
            import torch  # noqa: F401  -- load torch's HIP runtime first so both share one libamdhip64
        except ImportError:
            pass
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status):
    """Map a C-ABI status to the exception type the reference raises at the same seam."""
    if status == SSD_OK:
        return
    msg = lib().ssd_status_string(status).decode()
    if status == SSD_ERR_ASSERT:
        raise AssertionError(msg)
    if status in (SSD_ERR_VALUE, SSD_ERR_UNSUPPORTED):
        raise ValueError(msg)
    raise RuntimeError("ssd_hip: %s (status %d)" % (msg, status))
