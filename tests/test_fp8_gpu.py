"""GPU: block-scaled fp8 (MX: OCP e4m3 + E8M0 per 32 channels) forward convolution on v_mfma_scale_f32_16x16x128_f8f6f4 --
BASELINE configs[4]'s "fp8 MFMA convs".  No reference counterpart (fp32 TensorFlow convolutions): checked against
  (a) the fp32 restatement on the SAME dequantised operands -- the kernel's arithmetic, exact up to fp32 summation order and one
      bf16 output rounding: 2^-7 of the tensor maximum, the bound of every other forward convolution here;
  (b) the fp32 convolution of the original bf16 operands -- the quantisation error of the format itself, STATED, not hidden:
      relative L2 error <= 5e-2 (measured 3.7e-2: three mantissa bits per operand and power-of-two block scales, fp32 accumulation over 1152-4608 products);
  (c) the quantiser: scale = smallest power of two that fits the block into e4m3's range, elements = round-to-nearest e4m3."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import torch.nn.functional as F                                      # noqa: E402


@pytest.fixture(scope="module")
def ops():
    import ssd_object_detection_amd.ops as ops_
    return ops_


def test_quantiser(ops):
    g = torch.Generator(device="cuda").manual_seed(0)
    x = (torch.randn((4, 7, 9, 256), generator=g, device="cuda") * torch.rand((4, 7, 9, 1), generator=g, device="cuda") * 8).bfloat16()
    x[0, 0, 0, :32] = 0                                               # an all-zero block
    x[1, 2, 3, 64:96] = torch.tensor(448.0)                           # exactly the largest finite e4m3 value: scale 1
    q, s = ops.quantize_mx_fp8(x)
    xf = x.float().view(4, 7, 9, 8, 32)
    amax = xf.abs().amax(-1)
    want_e = torch.where(amax > 0, torch.ceil(torch.log2(amax / 448.0)), torch.zeros_like(amax))
    assert torch.equal(s.float() - 127.0, want_e)
    deq = ops.dequantize_mx_fp8(q, s)
    ref = (xf / torch.exp2(want_e).unsqueeze(-1)).to(torch.float8_e4m3fn).float() * torch.exp2(want_e).unsqueeze(-1)
    assert torch.equal(deq.view(xf.shape), ref)
    assert float((deq - x.float()).abs().max() / x.float().abs().max()) <= 2 ** -4


@pytest.mark.parametrize("shape", [(2, 19, 19, 256, 256), (3, 33, 20, 128, 136), (2, 64, 64, 512, 512), (16, 64, 64, 256, 512)],
                         ids=str)
def test_mxfp8_forward(ops, shape):
    B, H, W, Cin, Cout = shape
    g = torch.Generator(device="cuda").manual_seed(H + Cin)
    x = torch.randn((B, H, W, Cin), generator=g, device="cuda").relu().bfloat16()
    w = (torch.randn((Cout, 3, 3, Cin), generator=g, device="cuda") / np.sqrt(9 * Cin)).bfloat16()
    bias = torch.randn((Cout,), generator=g, device="cuda") * 0.1
    xq, xs = ops.quantize_mx_fp8(x)
    wq, ws = ops.quantize_mx_fp8(w)
    y = ops.conv3x3_fwd_mxfp8(xq, xs, wq, ws, bias, relu=True).float()
    y2 = ops.conv3x3_fwd_mxfp8(xq, xs, wq, ws, bias, relu=True).float()
    assert torch.equal(y, y2)

    def conv(xx, ww):
        return F.conv2d(xx.permute(0, 3, 1, 2), ww.permute(0, 3, 1, 2), bias, padding=1).relu().permute(0, 2, 3, 1)

    with torch.no_grad():
        ya = conv(ops.dequantize_mx_fp8(xq, xs), ops.dequantize_mx_fp8(wq, ws))      # fp32 on the device: the kernel's own arithmetic
        assert (y - ya).abs().max().item() <= 2 ** -7 * max(1.0, ya.abs().max().item()), "kernel vs fp32 on the dequantised operands"
        yb = conv(x.float(), w.float())
        err = float((y - yb).norm() / yb.norm())
        print("mxfp8 forward %s: relative L2 error vs the fp32 convolution of the bf16 operands %.4f" % (str(shape), err))
        assert err <= 5e-2
    # the bf16 kernel of the same layer for comparison (one bf16 rounding: ~3e-3)
    yh = ops.conv2d_fwd(x, w, bias, 1, 1, 1, H, W, True).float()
    assert float((yh - yb).norm() / yb.norm()) <= 5e-3
