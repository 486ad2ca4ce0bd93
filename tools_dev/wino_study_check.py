"""GPU numerics of the Winograd F(2x2,3x3) kernels (csrc/wino.hip; not dispatched by the engine) through the
C ABI against the same plain PyTorch fp32 reference and with the same bounds as the direct kernels (tests/test_conv_gpu.py):
forward (+bias, +ReLU), fused pooling (bit-equal to pooling its own output), data gradient plain and with ReLU mask +
accumulation, the power-of-two operand scale (exact), and the shapes it refuses."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
import torch.nn.functional as F                                     # noqa: E402


@pytest.fixture(scope="module")
def ops():
    import ssd_object_detection_amd.ops as ops_
    return ops_


def ref(x, w, b, relu):
    y = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), b, padding=1)
    return (y.relu() if relu else y).permute(0, 2, 3, 1).contiguous()


# block layouts: one 16x16 block; partial blocks in both dims; row strip over images (odd height); two channel tiles; 75x75
SHAPES = [(1, 16, 16, 64, 64), (2, 20, 37, 64, 64), (3, 19, 19, 64, 128), (2, 33, 18, 128, 192), (1, 75, 75, 128, 64)]


@pytest.mark.parametrize("shape", SHAPES, ids=[str(s) for s in SHAPES])
def test_wino_fwd_bwd(ops, shape):
    B, H, W, Cin, Cout = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn((B, H, W, Cin), generator=g).bfloat16()
    w = (torch.randn((Cout, 3, 3, Cin), generator=g) / np.sqrt(9 * Cin)).bfloat16()
    bias = torch.randn((Cout,), generator=g) * 0.1
    xd, wd, bd = x.cuda(), w.cuda(), bias.cuda()
    assert ops.wino_supported(B, H, W, Cin, Cout)
    u = ops.wino_weights(wd)
    for relu in (True, False):
        y = ops.conv3x3_wino_fwd(xd, u, bd, Cout, relu).float().cpu()
        yr = ref(x, w, bias, relu)
        assert (y - yr).abs().max().item() <= 2 ** -7 * max(1.0, yr.abs().max().item()), ("fwd", relu)
        assert ((y - yr).norm() / yr.norm()).item() <= 2.5e-3                 # bf16 output rounding: 1.7e-3, as the direct kernels
    for mode in ("same", "valid"):
        y, yp, code = ops.conv3x3_wino_fwd(xd, u, bd, Cout, True, pool=mode)
        yp_ref, code_ref = ops.maxpool2x2_fwd_argmax(y, mode == "same")
        assert torch.equal(yp, yp_ref) and torch.equal(code, code_ref), ("pool", mode)
        none, yp2, code2 = ops.conv3x3_wino_fwd(xd, u, bd, Cout, True, pool=mode, pool_only=True)
        assert none is None and torch.equal(yp2, yp_ref) and torch.equal(code2, code_ref)
    if not ops.wino_supported(B, H, W, Cout, Cin):
        return
    dy = torch.randn((B, H, W, Cout), generator=g).bfloat16()
    xr = x.float().requires_grad_(True)
    ref(xr, w, bias, False).backward(dy.float())
    ut = ops.wino_weights(ops.weight_transpose(wd))
    dx = ops.conv3x3_wino_bwd_data(dy.cuda(), ut, None, (B, H, W, Cin)).float().cpu()
    scale = max(1.0, xr.grad.abs().max().item())
    assert (dx - xr.grad).abs().max().item() <= 2 ** -7 * scale, "dgrad"
    mask = torch.randn((B, H, W, Cin), generator=g).bfloat16()
    base = torch.randn((B, H, W, Cin), generator=g).bfloat16()
    acc = base.clone().cuda()
    ops.conv3x3_wino_bwd_data(dy.cuda(), ut, mask.cuda(), (B, H, W, Cin), accumulate=True, out=acc)
    want = (xr.grad + base.float()) * (mask.float() > 0)
    assert (acc.float().cpu() - want).abs().max().item() <= 2 ** -6 * max(1.0, want.abs().max().item()), "dgrad+mask+acc"
    # gradients far below fp16's normal range: the operand scale brings them back, and 2^k scaling is exact in both types
    tiny = (dy.float() * 2.0 ** -20).bfloat16().cuda()
    dx_s = ops.conv3x3_wino_bwd_data(tiny, ut, None, (B, H, W, Cin), in_shift=16).float().cpu() * 2.0 ** 20
    assert (dx_s - xr.grad).abs().max().item() <= 2 ** -7 * scale, "dgrad with operand scale"
    dx_w = ops.conv3x3_wino_bwd_data(dy.cuda(), ops.wino_weights(ops.weight_transpose(wd), w_shift=3), None, (B, H, W, Cin),
                                     w_shift=3).float().cpu()
    assert torch.equal(dx_w, dx), "weight scale is exact"


def test_wino_refuses_unsupported_shapes(ops):
    assert not ops.wino_supported(2, 16, 16, 32, 64)            # Cin % 64
    assert not ops.wino_supported(2, 16, 16, 64, 96)            # Cout % 64
    assert not ops.wino_supported(2, 10, 10, 64, 64)            # map smaller than a block
    x = torch.zeros((2, 10, 10, 64), dtype=torch.bfloat16, device="cuda")
    u = torch.zeros(16 * 64 * 64, dtype=torch.float16, device="cuda")
    with pytest.raises(ValueError):
        ops.conv3x3_wino_fwd(x, u, None, 64, True)
