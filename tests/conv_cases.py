"""Convolution test shapes shared by tests/test_conv_gpu.py (numerics vs the fp32 reference, -m gpu) and
tests/test_conv_plan_cpu.py (dispatch coverage, CPU).  Every case names the kernels it is there to test -- (forward,
data gradient, weight gradient) as the library's dispatch query reports them (ssd_conv2d_*_plan: the dispatch code
itself with launching switched off, a host function, so it runs without a GPU).  Both test files assert the names."""

WS_BYTES = 1 << 25                      # the split-K workspace ops.py hands to every convolution call


def _geom(H, W, k, stride, mode):
    if mode == "same":
        def sp(n):
            out = -(-n // stride)
            return out, max((out - 1) * stride + k - n, 0) // 2
        (Ho, pt), (Wo, pl) = sp(H), sp(W)
    else:
        Ho, Wo, pt, pl = (H - k) // stride + 1, (W - k) // stride + 1, 0, 0
    return Ho, Wo, pt, pl


def plan_name(L, plan):
    if plan < 0:
        return "error %d" % plan
    flags = [n for bit, n in ((0x100, "flat"), (0x200, "rowflat"), (0x400, "splitk"), (0x800, "poolfused"), (0x1000, "s2"),
                              (0x2000, "rwide")) if plan & bit]
    return "+".join([L.ssd_conv_plan_name(plan).decode()] + flags)


def plan_names(case):
    """(forward, data-gradient, weight-gradient) kernel names of a (B, H, W, Cin, Cout, k, stride, mode) case, called the way
    the tests call the library (data / weight gradients with the output channels padded to a multiple of 8)."""
    from ssd_object_detection_amd import _lib
    L = _lib.lib()
    B, H, W, Cin, Cout, k, stride, mode = case
    Ho, Wo, pt, pl = _geom(H, W, k, stride, mode)
    cp = (Cout + 7) // 8 * 8
    return (plan_name(L, L.ssd_conv2d_fwd_plan(B, H, W, Cin, Cout, k, stride, pt, pl, Ho, Wo, 0, WS_BYTES)),
            plan_name(L, L.ssd_conv2d_bwd_data_plan(B, H, W, Cin, cp, k, stride, pt, pl, Ho, Wo, 0, WS_BYTES)),
            plan_name(L, L.ssd_conv2d_bwd_weight_plan(B, H, W, Cin, Cout, cp, k, stride, pt, pl, Ho, Wo)))


P32_64, P32_128, P512 = "k_conv3x3_patch32<64>", "k_conv3x3_patch32<128>", "k_conv3x3_p512"
DMA = "k_conv_igemm_dma<%s>"
WP16, WP6, WP10 = "k_conv3x3_wgrad_patch<16,2>", "k_conv3x3_wgrad_patch<6,5>", "k_conv3x3_wgrad_patch<10,3>"

CASES = [
    # B, H, W, Cin, Cout, k, stride, mode, (forward, data gradient, weight gradient)
    (2, 38, 38, 64, 128, 3, 1, "same", (P32_128 + "+rowflat", P32_64 + "+rowflat", WP6)),                  # 3x3 SAME, 6x40 wgrad blocks
    (2, 30, 30, 64, 64, 3, 1, "same", ("k_conv3x3_c64b", "k_conv3x3_c64b", WP16)),                          # 64 -> 64: weights in registers
    (3, 19, 19, 128, 256, 1, 1, "same", (DMA % "128,128", DMA % "128,128", "k_conv_wgrad")),                # 1x1, small
    (2, 38, 38, 64, 96, 3, 2, "same", (DMA % "128,128" + "+splitk", DMA % "128,64" + "+splitk", "k_conv_wgrad")),   # stride 2, pad (0,1) (38 -> 19); Cin = 64: no parity classes
    (2, 19, 19, 64, 72, 3, 2, "same", (DMA % "128,128" + "+splitk", DMA % "128,64" + "+splitk", "k_conv_wgrad")),   # stride 2, pad (1,1) (19 -> 10), N not a multiple of 16
    (4, 5, 5, 128, 256, 3, 1, "valid", (DMA % "128,128" + "+splitk", DMA % "128,128" + "+splitk", "k_conv_wgrad")),  # 3x3 VALID (5 -> 3)
    (5, 3, 3, 128, 256, 3, 1, "valid", (DMA % "128,128" + "+splitk", DMA % "128,128" + "+splitk", "k_conv_wgrad")),  # 3 -> 1, M = 5
    (2, 20, 20, 8, 64, 3, 1, "same", ("k_conv0_fwd", P32_64 + "+rowflat", "k_conv0_wgrad")),                 # Cin = 8: the image-layer kernels
    (2, 40, 40, 128, 64, 3, 1, "same", (P32_64, P32_128, WP6)),                                              # patch kernel, two chunks, BN = 64, partial edge tiles
    (1, 50, 35, 128, 128, 3, 1, "same", (P32_128, P32_128, WP6)),                                            # patch kernel, BN = 128, non-square, both dims partial
    (2, 33, 33, 192, 96, 3, 1, "same", (P32_128 + "+rowflat", DMA % "128,128" + "+splitk", WP6)),            # three chunks; the data gradient (96 in-channels) takes the generic GEMM
    (1, 70, 70, 256, 320, 3, 1, "same", (DMA % "128,128" + "+splitk", DMA % "128,128" + "+splitk", WP10)),   # N > 256 on a wide map: generic GEMM with split-K
    (2, 38, 38, 64, 128, 3, 2, "same", (DMA % "128,128" + "+splitk", DMA % "128,64" + "+s2", "k_conv_wgrad")),   # stride-2 data gradient by parity classes (even size, pad (0,1))
    (3, 19, 19, 128, 64, 3, 2, "same", (DMA % "128,64" + "+splitk", DMA % "128,128" + "+s2", "k_conv_wgrad")),   # parity classes, odd size, pad (1,1)
    (2, 19, 19, 128, 192, 3, 1, "same", (P32_128 + "+flat", P32_128 + "+rowflat", WP10)),                     # strip blocks forward; weight gradient with 10x24 blocks
    (1, 25, 20, 64, 64, 3, 1, "same", ("k_conv3x3_c64b", "k_conv3x3_c64b", WP10)),                           # 10x24 blocks, partial in both dims
    (3, 19, 19, 256, 320, 1, 1, "same", (DMA % "128,128", DMA % "128,128", "k_conv_wgrad")),                  # pointwise, ragged channel tile (M = 1083: below the 256x256 wgrad GEMM)
    (2, 21, 21, 64, 264, 3, 2, "same", (DMA % "128,128" + "+splitk", DMA % "128,64" + "+splitk", "k_conv_wgrad")),  # strided 3x3, 264 filters
    (2, 38, 38, 64, 192, 3, 1, "same", (P32_128 + "+flat", P32_64 + "+rowflat", WP6)),                        # strip blocks (narrow map, N > 128)
    (3, 19, 19, 128, 320, 3, 1, "same", (P32_128 + "+flat", P512 + "+rowflat", WP10)),                        # strip blocks spanning images, two chunks x three channel tiles; data gradient (320 in-channels) on the 512-pixel kernel with a ragged last chunk pair
    (5, 70, 45, 64, 64, 3, 1, "same", ("k_conv3x3_c64b", "k_conv3x3_c64b", WP10 + "+rwide")),                 # 64 -> 64 persistent kernel, ragged blocks; >= 32 wgrad splits
    (8, 48, 48, 64, 128, 3, 1, "same", (P32_128, P32_64, WP16 + "+rwide")),                                   # 16x16 wgrad blocks with >= 32 splits (the wide reduction, as block1-3)
    (2, 33, 18, 256, 128, 3, 1, "same", (P512 + "+rowflat", P32_128 + "+flat", WP10)),                        # 512-pixel kernel (>= 256 in-channels): 32-row strip blocks over two images, partial in both dims
    (3, 19, 19, 256, 256, 3, 1, "same", (P512 + "+flat", P512 + "+flat", WP10)),                              # 512-pixel kernel, strip-of-positions blocks spanning images, forward and data gradient
    (2, 40, 40, 256, 128, 3, 1, "same", (P512 + "+rowflat", DMA % "128,128" + "+splitk", WP6)),               # 512-pixel kernel, three column tiles, last one partial
]

# Shapes that reach the large-problem kernels; the first five and the last four are layers of the batch-64 SSD300 step.
FULL_SIZE_CASES = [
    (64, 38, 38, 512, 512, 1, 1, "same", ("k_pw_gemm", "k_pw_gemm", "k_conv_wgrad_tile+rwide")),                        # conv12 (1x1 at 38x38): the persistent pointwise GEMM
    (64, 19, 19, 1024, 1024, 1, 1, "same", ("k_pw_gemm", "k_pw_gemm", "k_conv_wgrad_tile")),                            # conv14 (1x1 at 19x19)
    (64, 38, 38, 512, 1024, 3, 2, "same", ("k_conv_igemm_8ph", DMA % "256,256" + "+s2", "k_conv_wgrad_tile")),          # conv13 (3x3 stride 2, 38 -> 19)
    (64, 19, 19, 1024, 256, 1, 1, "same", (DMA % "128,128", "k_pw_gemm", "k_conv_wgrad_tile+rwide")),                   # conv15 (91 tiles of 256x256 forward: the generic kernel; four k-tiles per tile backward)
    (70, 19, 19, 320, 320, 1, 1, "same", ("k_pw_gemm", "k_pw_gemm", "k_conv_wgrad_tile+rwide")),                              # pointwise GEMM with a ragged last pixel tile (M = 25270) and a ragged channel tile (320 = 256 + 64)
    (64, 38, 38, 512, 340, 3, 1, "same", (P512 + "+flat", "k_conv_igemm_8ph", WP6)),                                    # head 0 (data gradient from 344 padded channels)
    (8, 64, 64, 64, 320, 1, 1, "same", (DMA % "256,128", DMA % "128,64", "k_conv_wgrad+rwide")),                        # 256x128 tiles (N = 320 pads badly to 512)
    (6, 128, 128, 128, 64, 1, 1, "same", (DMA % "256,64", DMA % "256,128", "k_conv_wgrad+rwide")),                      # 256x64 tiles
    (11, 64, 64, 32, 512, 1, 1, "same", (DMA % "256,256", DMA % "128,64", "k_conv_wgrad+rwide")),                       # 256x256 LDS-DMA tiles without the 8-phase pipeline (Cin < 64)
    (64, 10, 10, 512, 128, 1, 1, "same", (DMA % "128,128" + "+splitk", DMA % "128,128", "k_conv_wgrad")),               # conv17: split-K + finalize at K = 512
    (64, 10, 10, 128, 256, 3, 2, "same", (DMA % "128,128" + "+splitk", DMA % "128,128" + "+s2", "k_conv_wgrad")),       # conv18
    (64, 5, 5, 128, 256, 3, 1, "valid", (DMA % "128,128" + "+splitk", DMA % "128,128" + "+splitk", "k_conv_wgrad")),    # conv20
    (64, 5, 5, 256, 510, 3, 1, "same", (DMA % "128,128" + "+splitk", DMA % "128,128" + "+splitk", "k_conv_wgrad")),     # head 3 (510 filters, 512 padded)
]

# tests/test_conv_gpu.py::test_first_layer_kernels_full_size (forward + weight gradient of the image layer at 300x300)
FIRST_LAYER_CASE = (4, 300, 300, 8, 64, 3, 1, "same", ("k_conv0_fwd", P32_64, "k_conv0_wgrad+rwide"))
