"""GPU: the data gradient of a convolution that follows a 2x2 max pooling, carried through the pooling in the convolution's
epilogue (ssd_conv2d_bwd_data_unpool), equals the two separate calls bit for bit -- the three pooling geometries of the SSD300
trunk (300 -> 150 and 150 -> 75 exact halves, 75 -> 38 SAME with a padded last row / column) at reduced batch, on every
LDS-patch kernel form (16x16 blocks, row strips, position strips, the 512-pixel kernel), with and without a ReLU mask; layers
that no patch kernel serves are refused before anything is launched."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import ssd_object_detection_amd.ops as ops_
    return ops_


# (B, Hf, Wf, C_pooled_map, Cout of the conv behind the pooling, same)
CASES = [(2, 300, 300, 64, 128, False),      # block1_pool -> block2_conv1 (patch32<64> data gradient)
         (3, 150, 150, 128, 256, False),     # block2_pool -> block3_conv1 (p512, row strips)
         (4, 75, 75, 256, 512, True),        # SAME 75 -> 38 -> block4_conv1 (p512, position strips)
         (2, 37, 45, 64, 128, True),         # odd sizes both ways, SAME
         (1, 64, 40, 128, 128, False)]


@pytest.mark.parametrize("case", CASES, ids=[str(c) for c in CASES])
def test_unpool_equals_two_calls(ops, case):
    B, Hf, Wf, C, Cout, same = case
    g = torch.Generator(device="cuda").manual_seed(Hf + C)
    full = torch.randn((B, Hf, Wf, C), generator=g, device="cuda").bfloat16()
    pooled, code = ops.maxpool2x2_fwd_argmax(full, same=same)
    H, W = pooled.shape[1:3]
    w = (torch.randn((Cout, 3, 3, C), generator=g, device="cuda") / np.sqrt(9 * C)).bfloat16()
    w_t = ops.weight_transpose(w)
    dy = torch.randn((B, H, W, Cout), generator=g, device="cuda").bfloat16()
    for mask in (None, torch.randn((B, H, W, C), generator=g, device="cuda").bfloat16()):
        dpool = ops.conv2d_bwd_data(dy, w_t, mask, (B, H, W, C), 1, 1, 1)
        want = ops.maxpool2x2_bwd_argmax(code, dpool, full.shape)
        got = torch.full_like(full, 7.0)                  # every element must be written
        ops.conv2d_bwd_data_unpool(dy, w_t, mask, code, full.shape, out=got)
        assert torch.equal(got, want)


def test_unpool_refused_without_a_patch_kernel(ops):
    B, Hf, C, Cout = 2, 20, 64, 64                         # pooled map 10x10: generic implicit GEMM
    full = torch.randn((B, Hf, Hf, C), device="cuda").bfloat16()
    pooled, code = ops.maxpool2x2_fwd_argmax(full, same=False)
    w_t = ops.weight_transpose((torch.randn((Cout, 3, 3, C), device="cuda") * 0.05).bfloat16())
    dy = torch.randn((B, 10, 10, Cout), device="cuda").bfloat16()
    with pytest.raises(NotImplementedError):
        ops.conv2d_bwd_data_unpool(dy, w_t, None, code, full.shape)
