"""GPU: ReLU sign bits.  ssd_conv2d_fwd_relubits writes, next to the activation, one byte per pixel and 8 channels (bit k:
channel 8c + k > 0); ssd_conv2d_bwd_data_bits masks the data gradient with those bytes instead of re-reading the bf16
activation.  Both must be bit-identical to the plain calls, on every kernel family that carries the bits (image layer,
64 -> 64 register-weight kernel, LDS-patch kernels, 512-pixel kernel, LDS-DMA GEMM, 8-phase GEMM), and refused -- before
anything is launched -- where a call resolves to split-K + finalize."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import ssd_object_detection_amd.ops as ops_
    return ops_


def pack_bits(y):
    """bool [..., C] -> uint8 [..., C/8], bit k of byte c = channel 8c + k"""
    b = (y > 0).to(torch.uint8).reshape(*y.shape[:-1], y.shape[-1] // 8, 8)
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=y.device)
    return (b * w).sum(-1).to(torch.uint8)


# (B, H, W, Cin, Cout, k): image layer; c64b; patch32 (row strips); p512; LDS-DMA GEMM (1x1); 8-phase GEMM (1x1, large)
CASES = [(2, 40, 40, 8, 64, 3), (2, 30, 30, 64, 64, 3), (2, 38, 38, 64, 128, 3), (3, 19, 19, 256, 256, 3), (3, 19, 19, 128, 256, 1),
         (64, 38, 38, 512, 512, 1)]


@pytest.mark.parametrize("case", CASES, ids=[str(c) for c in CASES])
def test_relubits_match_the_plain_calls(ops, case):
    B, H, W, Cin, Cout, k = case
    pad = (k - 1) // 2
    g = torch.Generator(device="cuda").manual_seed(H + Cin + k)
    x = torch.relu(torch.randn((B, H, W, Cin), generator=g, device="cuda")).bfloat16()
    w = (torch.randn((Cout, k, k, Cin), generator=g, device="cuda") / np.sqrt(k * k * Cin)).bfloat16()
    bias = torch.randn((Cout,), generator=g, device="cuda") * 0.1
    y = ops.conv2d_fwd(x, w, bias, 1, pad, pad, H, W, True)
    bits = torch.full((B, H, W, Cout // 8), 0xA5, dtype=torch.uint8, device="cuda")
    y2 = ops.conv2d_fwd_relubits(x, w, bias, 1, pad, pad, H, W, bits)
    assert torch.equal(y, y2)
    assert torch.equal(bits, pack_bits(y.float()))
    if Cin < 64:
        return                                              # no data gradient w.r.t. the image
    # data gradient of this layer, masked by the sign of its INPUT activation x
    xbits = pack_bits(x.float())
    dy = torch.randn((B, H, W, Cout), generator=g, device="cuda").bfloat16()
    w_t = ops.weight_transpose(w)
    base = torch.randn((B, H, W, Cin), generator=g, device="cuda").bfloat16()
    for accumulate in (False, True):
        want = base.clone()
        ops.conv2d_bwd_data(dy, w_t, x, (B, H, W, Cin), 1, pad, pad, accumulate=accumulate, out=want)
        got = base.clone()
        ops.conv2d_bwd_data_bits(dy, w_t, xbits, (B, H, W, Cin), 1, pad, pad, accumulate=accumulate, out=got)
        assert torch.equal(got, want), accumulate


def test_relubits_refused_on_split_k(ops):
    B, H, Cin, Cout = 4, 5, 128, 256                        # 3x3 VALID at 5x5: split-K + finalize
    x = torch.randn((B, H, H, Cin), device="cuda").bfloat16()
    w = (torch.randn((Cout, 3, 3, Cin), device="cuda") * 0.03).bfloat16()
    bits = torch.zeros((B, 3, 3, Cout // 8), dtype=torch.uint8, device="cuda")
    with pytest.raises(NotImplementedError):
        ops.conv2d_fwd_relubits(x, w, torch.zeros(Cout, device="cuda"), 1, 0, 0, 3, 3, bits)
