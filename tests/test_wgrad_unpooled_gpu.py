"""GPU parity of ssd_conv2d_bwd_weight_unpooled (csrc/wgrad_sp.hip): the weight gradient of a 3x3 convolution whose output is
max-pooled, computed from the POOLED gradient + winner codes with the structured-sparse MFMA (v_smfmac_f32_16x16x64_bf16),
against (a) the plain PyTorch fp32 restatement -- unpool by the recorded winners, then correlate (Keras Conv2D / MaxPooling2D
gradient semantics, models/ssd_model.py:77-84; the network oracle is UNPINNED: TensorFlow is not installable here) -- and (b) the
dense kernels on the explicitly un-pooled gradient (ssd_maxpool2x2_bwd_argmax + ssd_conv2d_bwd_weight).  Same 1e-3 bound as
every other weight gradient (fp32 accumulation order), bitwise reproducible, refusal of shapes it does not serve."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import torch.nn.functional as F                                      # noqa: E402


@pytest.fixture(scope="module")
def ops():
    import ssd_object_detection_amd.ops as ops_
    return ops_


def make_case(ops, B, H, W, Cin, Cout, same, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn((B, H, W, Cin), generator=g, device="cuda").relu().bfloat16()
    # a realistic pre-pool activation (post-ReLU, with exact zeros and ties) -> pooled map + winner codes
    y = torch.randn((B, H, W, Cout), generator=g, device="cuda").relu().bfloat16()
    y[:, ::3, ::5] = 0                                                # whole windows of zeros: code 4 = no gradient
    yp, code = ops.maxpool2x2_fwd_argmax(y, same=same)
    dp = torch.randn(yp.shape, generator=g, device="cuda").bfloat16()
    return x, y, yp, code, dp


# B, H, W, Cin, Cout, SAME pooling
CASES = [(2, 32, 32, 64, 64, False), (2, 75, 75, 64, 128, True), (3, 38, 50, 128, 64, False), (1, 16, 16, 64, 64, False),
         (2, 47, 33, 64, 64, True), (16, 150, 150, 128, 128, False)]


@pytest.mark.parametrize("case", CASES, ids=[str(c) for c in CASES])
def test_unpooled_wgrad_vs_dense_and_fp32(ops, case):
    B, H, W, Cin, Cout, same = case
    x, y, yp, code, dp = make_case(ops, B, H, W, Cin, Cout, same, seed=H + Cin)
    dw, db = ops.conv2d_bwd_weight_unpooled(x, dp, code)
    dw2, db2 = ops.conv2d_bwd_weight_unpooled(x, dp, code)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    # (b) the dense kernels on the explicitly un-pooled gradient
    dy = ops.maxpool2x2_bwd_argmax(code, dp, y.shape)
    assert int((dy != 0).sum()) <= dp.numel()                           # at most one non-zero per window and channel
    dwd, dbd = ops.conv2d_bwd_weight(x, dy, Cout, 3, 1, 1, 1)
    sw, sb = max(1.0, dwd.abs().max().item()), max(1.0, dbd.abs().max().item())
    assert (dw - dwd).abs().max().item() <= 1e-3 * sw
    assert (db - dbd).abs().max().item() <= 1e-3 * sb
    # (a) fp32 restatement on the CPU
    if B * H * W * Cin * Cout <= 2e9:
        dyr = dy.float().cpu()
        w = torch.zeros((Cout, 3, 3, Cin), requires_grad=True)
        yr = F.conv2d(x.float().cpu().permute(0, 3, 1, 2), w.permute(0, 3, 1, 2), padding=1)
        yr.backward(dyr.permute(0, 3, 1, 2))
        assert (dw.cpu() - w.grad).abs().max().item() <= 1e-3 * max(1.0, w.grad.abs().max().item())
        dbr = dyr.sum((0, 1, 2))
        assert (db.cpu() - dbr).abs().max().item() <= 1e-3 * max(1.0, dbr.abs().max().item())


def test_unpooled_wgrad_refuses_what_it_does_not_serve(ops):
    x = torch.zeros((1, 20, 20, 32), dtype=torch.bfloat16, device="cuda")            # 32 input channels
    dp = torch.zeros((1, 10, 10, 64), dtype=torch.bfloat16, device="cuda")
    code = torch.zeros((1, 10, 10, 8), dtype=torch.int32, device="cuda")
    with pytest.raises(NotImplementedError):
        ops.conv2d_bwd_weight_unpooled(x, dp, code)
