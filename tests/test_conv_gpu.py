"""GPU numerics: MFMA implicit-GEMM convolution forward / data-gradient / weight-gradient and pooling through
the C ABI vs a plain PyTorch fp32 reference of the same op (torch CPU conv2d on the same bf16-rounded operands,
TF-SAME padding made explicit).  Parity for the network is UNPINNED by the reference (TensorFlow absent):
tolerances are bf16 output rounding (fwd, dgrad: 2^-8 relative) and fp32 accumulation order (wgrad: 1e-3)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
import torch.nn.functional as F                                     # noqa: E402


@pytest.fixture(scope="module")
def ops():
    import ssd_object_detection_amd.ops as ops_
    return ops_


def ref_conv(x, w, bias, k, stride, pad_t, pad_l, Ho, Wo, relu):
    """x [B,H,W,C] f32, w [Cout,k,k,Cin] f32 -> [B,Ho,Wo,Cout] f32 with explicit (possibly asymmetric) padding."""
    B, H, W, C = x.shape
    pad_b = max((Ho - 1) * stride + k - H - pad_t, 0)
    pad_r = max((Wo - 1) * stride + k - W - pad_l, 0)
    xn = F.pad(x.permute(0, 3, 1, 2), (pad_l, pad_r, pad_t, pad_b))
    y = F.conv2d(xn, w.permute(0, 3, 1, 2), bias, stride=stride)
    assert y.shape[2] == Ho and y.shape[3] == Wo
    if relu:
        y = y.relu()
    return y.permute(0, 2, 3, 1).contiguous()


from tests.conv_cases import CASES, FIRST_LAYER_CASE, FULL_SIZE_CASES, plan_names          # noqa: E402  (shared with the CPU coverage test)


def geometry(ops, H, W, k, stride, mode):
    if mode == "same":
        Ho, pt = ops.same_pad(H, k, stride)
        Wo, pl = ops.same_pad(W, k, stride)
    else:
        Ho, Wo, pt, pl = ops.valid_out(H, k, stride), ops.valid_out(W, k, stride), 0, 0
    return Ho, Wo, pt, pl


@pytest.mark.parametrize("case", CASES, ids=[str(c[:8]) for c in CASES])
def test_conv_fwd_bwd(ops, case):
    B, H, W, Cin, Cout, k, stride, mode = case[:8]
    assert plan_names(case[:8]) == case[8], "the dispatch rules moved: this case no longer tests the kernels it names"
    Ho, pt = geometry(ops, H, H, k, stride, mode)[0], geometry(ops, H, H, k, stride, mode)[2]
    Wo, pl = geometry(ops, W, W, k, stride, mode)[0], geometry(ops, W, W, k, stride, mode)[2]
    g = torch.Generator().manual_seed(hash(case[:8]) % 1000)
    x = torch.randn((B, H, W, Cin), generator=g).bfloat16()
    w = (torch.randn((Cout, k, k, Cin), generator=g) / np.sqrt(k * k * Cin)).bfloat16()
    bias = torch.randn((Cout,), generator=g) * 0.1
    dy = torch.randn((B, Ho, Wo, Cout), generator=g).bfloat16()
    xd, wd, bd, dyd = x.cuda(), w.cuda(), bias.cuda(), dy.cuda()

    # forward (with and without ReLU)
    for relu in (True, False):
        y = ops.conv2d_fwd(xd, wd, bd, stride, pt, pl, Ho, Wo, relu).float().cpu()
        yr = ref_conv(x.float(), w.float(), bias, k, stride, pt, pl, Ho, Wo, relu)
        err = (y - yr).abs().max().item()
        assert err <= 2 ** -7 * max(1.0, yr.abs().max().item()), ("fwd", relu, err)

    # reference gradients by autograd (fp32)
    xr = x.float().requires_grad_(True)
    wr = w.float().requires_grad_(True)
    br = bias.clone().requires_grad_(True)
    yr = ref_conv(xr, wr, br, k, stride, pt, pl, Ho, Wo, False)
    yr.backward(dy.float())

    # data gradient (no mask, then with a ReLU mask and accumulation)
    w_t = ops.weight_transpose(wd)
    dx = ops.conv2d_bwd_data(dyd, w_t, None, (B, H, W, Cin), stride, pt, pl).float().cpu()
    scale = max(1.0, xr.grad.abs().max().item())
    assert (dx - xr.grad).abs().max().item() <= 2 ** -7 * scale, "dgrad"
    mask_src = torch.randn((B, H, W, Cin), generator=g).bfloat16()
    base = torch.randn((B, H, W, Cin), generator=g).bfloat16()
    acc = base.clone().cuda()
    ops.conv2d_bwd_data(dyd, w_t, mask_src.cuda(), (B, H, W, Cin), stride, pt, pl, accumulate=True, out=acc)
    want = (xr.grad + base.float()) * (mask_src.float() > 0)
    assert (acc.float().cpu() - want).abs().max().item() <= 2 ** -6 * max(1.0, want.abs().max().item()), "dgrad+mask+acc"

    # weight / bias gradient
    dw, db = ops.conv2d_bwd_weight(xd, dyd, Cout, k, stride, pt, pl)
    ws = max(1.0, wr.grad.abs().max().item())
    assert (dw.cpu() - wr.grad).abs().max().item() <= 1e-3 * ws, "wgrad"
    assert (db.cpu() - br.grad).abs().max().item() <= 1e-3 * max(1.0, br.grad.abs().max().item()), "bias grad"
    # deterministic
    dw2, _ = ops.conv2d_bwd_weight(xd, dyd, Cout, k, stride, pt, pl)
    assert torch.equal(dw, dw2)


@pytest.mark.parametrize("case", FULL_SIZE_CASES, ids=[str(c[:8]) for c in FULL_SIZE_CASES])
def test_conv_full_size_dispatch(ops, case):
    """The kernels that only large problems reach -- the 8-phase 256x256 implicit GEMM, the 256-row LDS-DMA tiles, the
    256x256 weight-gradient GEMM, split-K + finalize and the wide split reduction at the network's real K -- against
    the fp32 reference with the same bounds as the small cases: forward, data gradient (plain, and ReLU mask +
    accumulation), weight and bias gradient.  Most shapes are layers of the batch-64 SSD300 step itself.  The expected
    kernels are asserted through the library's dispatch query first, so a future tile-rule change cannot silently
    orphan a case (tests/test_conv_plan_cpu.py checks the other direction: every kernel of the batch-64 step has a case)."""
    B, H, W, Cin, Cout, k, stride, mode = case[:8]
    assert plan_names(case[:8]) == case[8]
    Ho, Wo, pt, pl = geometry(ops, H, W, k, stride, mode)
    cp = (Cout + 7) // 8 * 8                                  # head gradients arrive channel-padded
    g = torch.Generator().manual_seed(B * 7 + Cin + Cout)
    x = torch.randn((B, H, W, Cin), generator=g).bfloat16()
    w = (torch.randn((Cout, k, k, Cin), generator=g) / np.sqrt(k * k * Cin)).bfloat16()
    bias = torch.randn((Cout,), generator=g) * 0.1
    dy = torch.zeros((B, Ho, Wo, cp), dtype=torch.bfloat16)
    dy[..., :Cout] = torch.randn((B, Ho, Wo, Cout), generator=g).bfloat16()
    xd, wd, bd, dyd = x.cuda(), w.cuda(), bias.cuda(), dy.cuda()
    y = ops.conv2d_fwd(xd, wd, bd, stride, pt, pl, Ho, Wo, True).float().cpu()
    xr = x.float().requires_grad_(True)
    wr = w.float().requires_grad_(True)
    br = bias.clone().requires_grad_(True)
    yr = ref_conv(xr, wr, br, k, stride, pt, pl, Ho, Wo, False)
    err = (y - yr.detach().relu()).abs().max().item()
    assert err <= 2 ** -7 * max(1.0, yr.abs().max().item()), ("fwd", err)
    yr.backward(dy[..., :Cout].float())
    del yr, y
    w_t = ops.weight_transpose(wd, cp)
    dx = ops.conv2d_bwd_data(dyd, w_t, None, (B, H, W, Cin), stride, pt, pl).float().cpu()
    scale = max(1.0, xr.grad.abs().max().item())
    assert (dx - xr.grad).abs().max().item() <= 2 ** -7 * scale, "dgrad"
    mask_src = torch.randn((B, H, W, Cin), generator=g).bfloat16()
    base = torch.randn((B, H, W, Cin), generator=g).bfloat16()
    acc = base.clone().cuda()
    ops.conv2d_bwd_data(dyd, w_t, mask_src.cuda(), (B, H, W, Cin), stride, pt, pl, accumulate=True, out=acc)
    want = (xr.grad + base.float()) * (mask_src.float() > 0)
    assert (acc.float().cpu() - want).abs().max().item() <= 2 ** -6 * max(1.0, want.abs().max().item()), "dgrad+mask+acc"
    dw, db = ops.conv2d_bwd_weight(xd, dyd, Cout, k, stride, pt, pl)
    ws = max(1.0, wr.grad.abs().max().item())
    assert (dw.cpu() - wr.grad).abs().max().item() <= 1e-3 * ws, "wgrad"
    assert (db.cpu() - br.grad).abs().max().item() <= 1e-3 * max(1.0, br.grad.abs().max().item()), "bias grad"


# (B, H, Cin, Cout, forward on k_pw_gemm, data gradient on k_pw_gemm)
PW_SHAPES = [(64, 38, 512, 512, True, True), (64, 19, 1024, 1024, True, True), (64, 19, 256, 1024, True, False),
             (64, 19, 1024, 256, False, True), (70, 19, 320, 320, True, True), (71, 19, 320, 512, True, True)]


@pytest.mark.parametrize("shape", PW_SHAPES, ids=[str(s) for s in PW_SHAPES])
def test_pw_gemm_equals_the_generic_kernels(ops, shape):
    """k_pw_gemm (persistent 1x1 GEMM: the LDS-DMA stream runs across tile boundaries, wave-private store stage, bias / ReLU
    sign bytes through LDS) accumulates every output element over the same 32-deep k chunks in the same order as the
    one-tile-per-workgroup implicit-GEMM kernels it replaces, so the two agree BIT FOR BIT -- forward with bias, ReLU and sign
    bytes; data gradient plain, with the ReLU mask as sign bytes, and accumulating onto an existing gradient -- at the batch-64
    layer shapes (conv12, conv14 and conv15's data-gradient shape, as forward and as data gradient) and on ragged pixel / channel tiles.  The generic
    kernels are compared with the fp32 reference in the cases above; the forward here once more, directly."""
    from ssd_object_detection_amd import _lib
    from tests.conv_cases import plan_name
    B, H, Cin, Cout, pw_fwd, pw_dgrad = shape
    L = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(H * 7 + Cin + Cout)
    x = torch.randn((B, H, H, Cin), generator=g, device="cuda").relu().bfloat16()
    w = (torch.randn((Cout, 1, 1, Cin), generator=g, device="cuda") / np.sqrt(Cin)).bfloat16()
    bias = torch.randn((Cout,), generator=g, device="cuda") * 0.1
    dy = torch.randn((B, H, H, Cout), generator=g, device="cuda").bfloat16()
    base = torch.randn((B, H, H, Cin), generator=g, device="cuda").bfloat16()
    w_t = ops.weight_transpose(w)
    outs, names = [], []
    for v in (0, 3):
        assert L.ssd_dev_knob(b"SSD_CONV_PW", v) == 0
        try:
            names.append((plan_name(L, L.ssd_conv2d_fwd_plan(B, H, H, Cin, Cout, 1, 1, 0, 0, H, H, 0, 1 << 25)),
                          plan_name(L, L.ssd_conv2d_bwd_data_plan(B, H, H, Cin, Cout, 1, 1, 0, 0, H, H, 1, 1 << 25))))
            bits = torch.zeros((B, H, H, Cout // 8), dtype=torch.uint8, device="cuda")
            y = ops.conv2d_fwd_relubits(x, w, bias, 1, 0, 0, H, H, bits)
            y0 = ops.conv2d_fwd(x, w, bias, 1, 0, 0, H, H, False)
            xbits = ((x > 0).view(B, H, H, Cin // 8, 8).to(torch.uint8) * (2 ** torch.arange(8, device="cuda", dtype=torch.uint8))).sum(-1).to(torch.uint8).contiguous()
            dx_plain = ops.conv2d_bwd_data(dy, w_t, None, (B, H, H, Cin), 1, 0, 0)
            dx_bits = ops.conv2d_bwd_data_bits(dy, w_t, xbits, (B, H, H, Cin), 1, 0, 0)
            acc = base.clone()
            ops.conv2d_bwd_data_bits(dy, w_t, xbits, (B, H, H, Cin), 1, 0, 0, accumulate=True, out=acc)
        finally:
            L.ssd_dev_knob(b"SSD_CONV_PW", 3)
        outs.append((y, bits, y0, dx_plain, dx_bits, acc))
    assert (names[1][0] == "k_pw_gemm", names[1][1] == "k_pw_gemm") == (pw_fwd, pw_dgrad), names
    assert "k_pw_gemm" not in names[0] and "splitk" not in "".join(names[0]), names
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    y, bits = outs[1][0], outs[1][1]
    want_bits = ((y > 0).view(B, H, H, Cout // 8, 8).to(torch.uint8) * (2 ** torch.arange(8, device="cuda", dtype=torch.uint8))).sum(-1).to(torch.uint8)
    assert torch.equal(bits, want_bits)
    assert torch.equal(outs[1][4], outs[1][3] * (x > 0))                       # the sign-byte mask is the ReLU mask
    with torch.no_grad():
        yr = (x.float().view(-1, Cin) @ w.float().view(Cout, Cin).t() + bias).relu().view(B, H, H, Cout)
        assert (y.float() - yr).abs().max().item() <= 2 ** -7 * max(1.0, yr.abs().max().item())
        dxr = dy.float().view(-1, Cout) @ w.float().view(Cout, Cin)
        assert (outs[1][3].float().view(-1, Cin) - dxr).abs().max().item() <= 2 ** -7 * max(1.0, dxr.abs().max().item())


def test_first_layer_kernels_full_size(ops):
    """The dedicated image-layer kernels (8 padded channels -> 64) at the real map size: more blocks than persistent
    workgroups (several iterations per workgroup), ragged right/bottom blocks, forward and weight gradient."""
    B, H, Cin, Cout = 4, 300, 8, 64
    assert FIRST_LAYER_CASE[:5] == (B, H, H, Cin, Cout) and plan_names(FIRST_LAYER_CASE[:8]) == FIRST_LAYER_CASE[8]
    g = torch.Generator().manual_seed(21)
    x = torch.zeros((B, H, H, Cin)).bfloat16()
    x[..., :3] = torch.randn((B, H, H, 3), generator=g).bfloat16()
    w = (torch.randn((Cout, 3, 3, Cin), generator=g) / np.sqrt(27)).bfloat16()
    bias = torch.randn((Cout,), generator=g) * 0.1
    dy = torch.randn((B, H, H, Cout), generator=g).bfloat16()
    y = ops.conv2d_fwd(x.cuda(), w.cuda(), bias.cuda(), 1, 1, 1, H, H, True).float().cpu()
    wr = w.float().requires_grad_(True)
    br = bias.clone().requires_grad_(True)
    yr = ref_conv(x.float(), wr, br, 3, 1, 1, 1, H, H, False)
    assert (y - yr.detach().relu()).abs().max().item() <= 2 ** -7 * max(1.0, yr.abs().max().item())
    yr.backward(dy.float())
    dw, db = ops.conv2d_bwd_weight(x.cuda(), dy.cuda(), Cout, 3, 1, 1, 1)
    assert (dw.cpu() - wr.grad).abs().max().item() <= 2e-3 * max(1.0, wr.grad.abs().max().item())
    assert (db.cpu() - br.grad).abs().max().item() <= 2e-3 * max(1.0, br.grad.abs().max().item())
    dw2, _ = ops.conv2d_bwd_weight(x.cuda(), dy.cuda(), Cout, 3, 1, 1, 1)
    assert torch.equal(dw, dw2)


@pytest.mark.parametrize("case", [(2, 30, 30, 64, 64), (3, 17, 23, 128, 96), (1, 38, 38, 64, 136)])
def test_patch_strip_blocks_forced(ops, case):
    """Strip blocks (SSD_CONV_PATCH_FLAT=2 forces them wherever the map is narrow enough) give the same forward and
    data gradient as the 16x16 blocks: the block shape is a tuning choice only."""
    from ssd_object_detection_amd import _lib
    B, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(3)
    x = torch.randn((B, H, W, Cin), generator=g).bfloat16()
    w = (torch.randn((Cout, 3, 3, Cin), generator=g) / np.sqrt(9 * Cin)).bfloat16()
    bias = torch.randn((Cout,), generator=g) * 0.1
    dy = torch.randn((B, H, W, Cout), generator=g).bfloat16()
    L = _lib.lib()
    outs = []
    for flat in (0, 2):
        assert L.ssd_dev_knob(b"SSD_CONV_PATCH_FLAT", flat) == 0
        try:
            y = ops.conv2d_fwd(x.cuda(), w.cuda(), bias.cuda(), 1, 1, 1, H, W, True)
            dx = ops.conv2d_bwd_data(dy.cuda(), ops.weight_transpose(w.cuda()), x.cuda(), (B, H, W, Cin), 1, 1, 1)
        finally:
            L.ssd_dev_knob(b"SSD_CONV_PATCH_FLAT", 1)
        outs.append((y.float().cpu(), dx.float().cpu()))
    yr = ref_conv(x.float(), w.float(), bias, 3, 1, 1, 1, H, W, True)
    assert (outs[1][0] - yr).abs().max().item() <= 2 ** -7 * max(1.0, yr.abs().max().item())
    if Cout <= 128:     # both runs use the patch kernel, same k order -> bitwise equal
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    else:               # the unforced run used the generic implicit GEMM (other summation order)
        assert (outs[0][0] - outs[1][0]).abs().max().item() <= 2 ** -7 * max(1.0, yr.abs().max().item())
        assert (outs[0][1] - outs[1][1]).abs().max().item() <= 2 ** -7 * max(1.0, outs[0][1].abs().max().item())


@pytest.mark.parametrize("case", [(3, 21, 21, 64, 128), (2, 45, 50, 128, 96), (5, 19, 40, 64, 72)])
def test_patch_row_strip_is_bitwise_neutral(ops, case):
    """Blocks over one strip of rows of all images (SSD_CONV_PATCH_ROWFLAT, default on: a block may straddle two images,
    which share one zero row) give the same forward and data gradient, bit for bit, as per-image 16x16 blocks."""
    from ssd_object_detection_amd import _lib
    B, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(11)
    x = torch.randn((B, H, W, Cin), generator=g).bfloat16().cuda()
    w = (torch.randn((Cout, 3, 3, Cin), generator=g) / np.sqrt(9 * Cin)).bfloat16().cuda()
    bias = (torch.randn((Cout,), generator=g) * 0.1).cuda()
    dy = torch.randn((B, H, W, Cout), generator=g).bfloat16().cuda()
    L = _lib.lib()
    outs = []
    for v in (0, 1):
        assert L.ssd_dev_knob(b"SSD_CONV_PATCH_ROWFLAT", v) == 0
        try:
            y = ops.conv2d_fwd(x, w, bias, 1, 1, 1, H, W, True)
            dx = ops.conv2d_bwd_data(dy, ops.weight_transpose(w), x, (B, H, W, Cin), 1, 1, 1)
        finally:
            L.ssd_dev_knob(b"SSD_CONV_PATCH_ROWFLAT", 1)
        outs.append((y, dx))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    yr = ref_conv(x.float().cpu(), w.float().cpu(), bias.cpu(), 3, 1, 1, 1, H, W, True)
    assert (outs[1][0].float().cpu() - yr).abs().max().item() <= 2 ** -7 * max(1.0, yr.abs().max().item())


# the batch-64 layers k_conv3x3_p512 serves (block3_conv2/3, block4_conv1/2 and their data gradients, head 1's data gradient
# shape) plus one small map with partial blocks in both dimensions
P512_LAYERS = [(64, 75, 75, 256, 256), (64, 38, 38, 256, 512), (64, 38, 38, 512, 512), (16, 19, 19, 1024, 512), (3, 33, 50, 256, 128)]


@pytest.mark.parametrize("case", P512_LAYERS, ids=[str(c) for c in P512_LAYERS])
def test_p512_equals_patch32_at_layer_size(ops, case):
    """k_conv3x3_p512 and k_conv3x3_patch32 accumulate in the same order (32-channel chunks, nine taps each), so they agree bit
    for bit: forward, forward with fused pooling, data gradient with ReLU mask and accumulation -- at the real layer sizes,
    where the 512-position blocks, the weight-slice ring and the per-tap vmcnt counts run for hundreds of steps.  patch32
    itself is compared with the fp32 reference in the cases above (and p512 at small shapes there too)."""
    from ssd_object_detection_amd import _lib
    from tests.conv_cases import plan_name
    B, H, W, Cin, Cout = case
    g = torch.Generator(device="cuda").manual_seed(H + Cin)
    x = torch.relu(torch.randn((B, H, W, Cin), generator=g, device="cuda")).bfloat16()
    w = (torch.randn((Cout, 3, 3, Cin), generator=g, device="cuda") / np.sqrt(9 * Cin)).bfloat16()
    bias = torch.randn((Cout,), generator=g, device="cuda") * 0.1
    dy = torch.randn((B, H, W, Cout), generator=g, device="cuda").bfloat16()
    base = torch.randn((B, H, W, Cin), generator=g, device="cuda").bfloat16()
    w_t = ops.weight_transpose(w)
    L = _lib.lib()
    outs, names = [], []
    for v in (0, 2):
        assert L.ssd_dev_knob(b"SSD_CONV_P512", v) == 0
        try:
            names.append(plan_name(L, L.ssd_conv2d_fwd_plan(B, H, W, Cin, Cout, 3, 1, 1, 1, H, W, 0, 1 << 25)))
            y = ops.conv2d_fwd(x, w, bias, 1, 1, 1, H, W, True)
            yp = ops.conv2d_fwd_pool(x, w, bias, 1, 1, 1, H, W, True, True)
            acc = base.clone()
            ops.conv2d_bwd_data(dy, w_t, x, (B, H, W, Cin), 1, 1, 1, accumulate=True, out=acc)
        finally:
            L.ssd_dev_knob(b"SSD_CONV_P512", 1)
        outs.append((y, yp[0], yp[1], yp[2], acc))
    assert names[0].startswith("k_conv3x3_patch32") and names[1].startswith("k_conv3x3_p512"), names
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_p512_row_strip_full_size_vs_fp32_oracle(ops):
    """The most-used launch of k_conv3x3_p512 -- block3_conv2/3 at batch 64: 75 x 75 x 256 -> 256 as one strip of rows over
    all images -- directly against the fp32 reference at that size: forward (+bias, ReLU) and the masked, accumulating data
    gradient.  (test_p512_equals_patch32_at_layer_size ties p512 to patch32 bit for bit; this closes the chain to the oracle
    at the size that runs.)"""
    from ssd_object_detection_amd import _lib
    from tests.conv_cases import plan_name
    B, H, W, Cin, Cout = 64, 75, 75, 256, 256
    L = _lib.lib()
    assert plan_name(L, L.ssd_conv2d_fwd_plan(B, H, W, Cin, Cout, 3, 1, 1, 1, H, W, 0, 1 << 25)) == "k_conv3x3_p512+rowflat"
    assert plan_name(L, L.ssd_conv2d_bwd_data_plan(B, H, W, Cin, Cout, 3, 1, 1, 1, H, W, 1, 1 << 25)) == "k_conv3x3_p512+rowflat"
    g = torch.Generator().manual_seed(75)
    x = torch.randn((B, H, W, Cin), generator=g).relu().bfloat16()
    w = (torch.randn((Cout, 3, 3, Cin), generator=g) / np.sqrt(9 * Cin)).bfloat16()
    bias = torch.randn((Cout,), generator=g) * 0.1
    dy = torch.randn((B, H, W, Cout), generator=g).bfloat16()
    base = torch.randn((B, H, W, Cin), generator=g).bfloat16()
    xd, wd = x.cuda(), w.cuda()
    y = ops.conv2d_fwd(xd, wd, bias.cuda(), 1, 1, 1, H, W, True).float().cpu()
    acc = base.clone().cuda()
    ops.conv2d_bwd_data(dy.cuda(), ops.weight_transpose(wd), xd, (B, H, W, Cin), 1, 1, 1, accumulate=True, out=acc)
    acc = acc.float().cpu()
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    with torch.no_grad():
        yr = ref_conv(x.float(), w.float(), bias, 3, 1, 1, 1, H, W, True)
        assert (y - yr).abs().max().item() <= 2 ** -7 * max(1.0, yr.abs().max().item())
        del yr, y
        # data gradient of a 3x3 / stride 1 / pad 1 convolution = the convolution of dy with the flipped, transposed filters
        wt = w.float().flip(1, 2).permute(3, 1, 2, 0).contiguous()                  # [Cin, 3, 3, Cout]
        dxr = ref_conv(dy.float(), wt, None, 3, 1, 1, 1, H, W, False)
        want = (dxr + base.float()) * (x.float() > 0)
    assert (acc - want).abs().max().item() <= 2 ** -6 * max(1.0, want.abs().max().item())


def test_wgrad_patch_full_size_vs_fp32_oracle(ops):
    """The step's dominant kernel at a size it runs at -- k_conv3x3_wgrad_patch<16,2> on block3_conv2/3's shape at batch 64
    (75 x 75 x 256 -> 256: 16 channel tiles x 16 pixel splits over the 256 workgroups, ragged 16x16 blocks on the right and
    bottom edges, XCD groups, the split reduction) -- directly against the fp32 reference (torch CPU autograd), same 1e-3 bound
    as the small cases, bias gradient too, and bitwise reproducible.  The other five launches of a step differ in channel
    counts only (64 ... 256: the same code with fewer tiles and more splits, covered at (8,48,48,64,128))."""
    from ssd_object_detection_amd import _lib
    from tests.conv_cases import plan_name
    B, H, W, Cin, Cout = 64, 75, 75, 256, 256
    L = _lib.lib()
    assert plan_name(L, L.ssd_conv2d_bwd_weight_plan(B, H, W, Cin, Cout, Cout, 3, 1, 1, 1, H, W)).startswith("k_conv3x3_wgrad_patch<16,2>")
    g = torch.Generator().manual_seed(7575)
    x = torch.randn((B, H, W, Cin), generator=g).relu().bfloat16()           # post-ReLU activations, as in the network
    dy = torch.randn((B, H, W, Cout), generator=g).bfloat16()
    xd, dyd = x.cuda(), dy.cuda()
    dw, db = ops.conv2d_bwd_weight(xd, dyd, Cout, 3, 1, 1, 1)
    dw2, db2 = ops.conv2d_bwd_weight(xd, dyd, Cout, 3, 1, 1, 1)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    w = torch.zeros((Cout, 3, 3, Cin), requires_grad=True)
    ref_conv(x.float(), w, None, 3, 1, 1, 1, H, W, False).backward(dy.float())
    assert (dw.cpu() - w.grad).abs().max().item() <= 1e-3 * max(1.0, w.grad.abs().max().item())
    dbr = dy.float().sum((0, 1, 2))
    assert (db.cpu() - dbr).abs().max().item() <= 1e-3 * max(1.0, dbr.abs().max().item())


@pytest.mark.parametrize("shape", [0, 1, 2])
def test_wgrad_patch_block_shapes(ops, shape):
    """Every block shape of the LDS-patch weight-gradient kernel (16x16, 6x40, 10x24) gives the same gradient;
    the shape is a tuning choice only (forced here through the development knob)."""
    from ssd_object_detection_amd import _lib
    B, H, W, Cin, Cout = 2, 23, 45, 64, 80
    g = torch.Generator().manual_seed(11 + shape)
    x = torch.randn((B, H, W, Cin), generator=g).bfloat16()
    dy = torch.randn((B, H, W, Cout), generator=g).bfloat16()
    w = torch.zeros((Cout, 3, 3, Cin), requires_grad=True)
    ref_conv(x.float(), w, None, 3, 1, 1, 1, H, W, False).backward(dy.float())
    L = _lib.lib()
    assert L.ssd_dev_knob(b"SSD_WGRAD_PATCH_SHAPE", shape) == 0
    try:
        dw, db = ops.conv2d_bwd_weight(x.cuda(), dy.cuda(), Cout, 3, 1, 1, 1)
    finally:
        L.ssd_dev_knob(b"SSD_WGRAD_PATCH_SHAPE", -1)
    assert (dw.cpu() - w.grad).abs().max().item() <= 1e-3 * max(1.0, w.grad.abs().max().item())
    assert (db.cpu() - dy.float().sum((0, 1, 2))).abs().max().item() <= 1e-3 * 60


def test_wgrad_padded_dy(ops):
    """Head gradients arrive in a channel-padded tensor (ldy > Cout)."""
    B, H, W, Cin, Cout, ldy = 2, 10, 10, 64, 340, 384
    g = torch.Generator().manual_seed(5)
    x = torch.randn((B, H, W, Cin), generator=g).bfloat16()
    dy = torch.zeros((B, H, W, ldy)).bfloat16()
    dy[..., :Cout] = torch.randn((B, H, W, Cout), generator=g).bfloat16()
    w = torch.zeros((Cout, 3, 3, Cin), requires_grad=True)
    yr = ref_conv(x.float(), w, None, 3, 1, 1, 1, H, W, False)
    yr.backward(dy[..., :Cout].float())
    dw, db = ops.conv2d_bwd_weight(x.cuda(), dy.cuda(), Cout, 3, 1, 1, 1)
    assert (dw.cpu() - w.grad).abs().max().item() <= 1e-3 * w.grad.abs().max().item()
    assert (db.cpu() - dy[..., :Cout].float().sum((0, 1, 2))).abs().max().item() <= 1e-3 * 30


@pytest.mark.parametrize("shape", [(2, 5, 5, 64, 6), (2, 19, 19, 256, 4), (2, 19, 19, 1024, 6)])   # generic GEMM; patch32 strip blocks (as head 0); k_conv3x3_p512 strip blocks (as head 1)
def test_head_fwd_layout(ops, shape):
    """Fused loc+conf head writes the reference's Reshape/Concatenate layout (models/ssd_model.py:166-167)."""
    (B, H, W, Cin, n), C = shape, 81
    A, off = 200 + H * W * n, 200
    g = torch.Generator().manual_seed(9)
    x = torch.randn((B, H, W, Cin), generator=g).bfloat16()
    w = (torch.randn((n * (4 + C), 3, 3, Cin), generator=g) / np.sqrt(9 * Cin)).bfloat16()
    bias = torch.randn((n * (4 + C),), generator=g) * 0.1
    loc = torch.zeros((B, A, 4), dtype=torch.bfloat16, device="cuda")
    conf = torch.zeros((B, A, C), dtype=torch.bfloat16, device="cuda")
    ops.conv2d_head_fwd(x.cuda(), w.cuda(), bias.cuda(), loc, conf, n, C, off)
    yr = ref_conv(x.float(), w.float(), bias, 3, 1, 1, 1, H, W, False)          # [B,H,W,n*85]
    loc_r = yr[..., :n * 4].reshape(B, H * W * n, 4)
    conf_r = yr[..., n * 4:].reshape(B, H * W * n, C)
    assert (loc[:, off:].float().cpu() - loc_r).abs().max().item() <= 2 ** -7 * max(1, loc_r.abs().max().item())
    assert (conf[:, off:].float().cpu() - conf_r).abs().max().item() <= 2 ** -7 * max(1, conf_r.abs().max().item())
    assert float(loc[:, :off].abs().max()) == 0 and float(conf[:, :off].abs().max()) == 0
    # and the inverse packing of the gradients
    dloc = torch.randn((B, A, 4), generator=g).bfloat16().cuda()
    dconf = torch.randn((B, A, C), generator=g).bfloat16().cuda()
    npad = 512
    packed = ops.head_grad_pack(dloc, dconf, H * W, n, C, npad, off).cpu()
    want = torch.cat([dloc[:, off:].cpu().reshape(B, H * W, n * 4), dconf[:, off:].cpu().reshape(B, H * W, n * C)], -1)
    assert torch.equal(packed[..., :n * 85], want) and float(packed[..., n * 85:].float().abs().max()) == 0


@pytest.mark.parametrize("H,same", [(20, False), (75, True), (38, False)])
def test_maxpool(ops, H, same):
    B, C = 2, 64
    g = torch.Generator().manual_seed(H)
    x = torch.randn((B, H, H, C), generator=g).relu().bfloat16()        # post-ReLU activations (many exact zeros)
    y = ops.maxpool2x2_fwd(x.cuda(), same=same)
    xn = x.float().permute(0, 3, 1, 2)
    if same and H % 2:
        xn = F.pad(xn, (0, 1, 0, 1), value=float("-inf"))
    yr = F.max_pool2d(xn, 2, 2).permute(0, 2, 3, 1)
    assert torch.equal(y.float().cpu(), yr)
    dy = torch.randn(y.shape, generator=g).bfloat16()
    dx = ops.maxpool2x2_bwd(x.cuda(), y, dy.cuda()).float().cpu()
    # reference: route to the first maximum of each window, then the ReLU mask x > 0
    xr = x.float().requires_grad_(True)
    xrn = xr.permute(0, 3, 1, 2)
    if same and H % 2:
        xrn = F.pad(xrn, (0, 1, 0, 1), value=float("-inf"))
    F.max_pool2d(xrn, 2, 2).backward(dy.float().permute(0, 3, 1, 2))
    want = xr.grad * (x.float() > 0)
    assert torch.equal(dx * (x.float() > 0), dx)
    assert (dx - want).abs().max().item() == 0
    # the recorded-winner form used in training gives the same bits
    y2, code = ops.maxpool2x2_fwd_argmax(x.cuda(), same=same)
    assert torch.equal(y2, y)
    dx2 = ops.maxpool2x2_bwd_argmax(code, dy.cuda(), x.shape).float().cpu()
    assert torch.equal(dx2, dx)


@pytest.mark.parametrize("case", [(2, 46, 64, 64, False), (2, 75, 128, 256, True), (1, 33, 64, 96, True), (2, 19, 128, 128, False),
                                  (2, 40, 256, 128, True), (1, 75, 256, 256, True)])      # the last two: k_conv3x3_p512's pooling epilogue
def test_conv_fwd_pool_fused(ops, case):
    """conv + ReLU + 2x2 pooling in one call == the separate calls, bit for bit (the fused form pools the tile the
    convolution kernel holds on chip; layers without a 16x16-block kernel fall back to two launches)."""
    B, H, Cin, Cout, same = case
    g = torch.Generator().manual_seed(H)
    x = torch.randn((B, H, H, Cin), generator=g).bfloat16().cuda()
    w = (torch.randn((Cout, 3, 3, Cin), generator=g) / np.sqrt(9 * Cin)).bfloat16().cuda()
    bias = (torch.randn((Cout,), generator=g) * 0.1).cuda()
    y, yp, code = ops.conv2d_fwd_pool(x, w, bias, 1, 1, 1, H, H, True, same)
    y_ref = ops.conv2d_fwd(x, w, bias, 1, 1, 1, H, H, True)
    yp_ref, code_ref = ops.maxpool2x2_fwd_argmax(y_ref, same=same)
    assert torch.equal(y, y_ref) and torch.equal(yp, yp_ref) and torch.equal(code, code_ref)
    # pool-only form (y == NULL in the C ABI): same pooled map and codes, no full-resolution store; a layer shape that no
    # pooling kernel serves is refused before anything is launched
    try:
        none, yp2, code2 = ops.conv2d_fwd_pool(x, w, bias, 1, 1, 1, H, H, True, same, pool_only=True)
    except ValueError:
        assert Cin % 64 != 0 or H < 16
    else:
        assert none is None and torch.equal(yp2, yp_ref) and torch.equal(code2, code_ref)


def test_image_prep(ops):
    img = torch.rand((2, 30, 30, 3))
    out = ops.image_prep(img.cuda()).float().cpu()
    assert torch.equal(out[..., :3], ((img - 0.5) * 2).bfloat16().float()) and float(out[..., 3:].abs().max()) == 0


@pytest.mark.parametrize("case", [(2, 300, 300), (3, 37, 52), (1, 16, 16), (70, 24, 40)])
def test_second_layer_dgrad_fused_with_first_layer_wgrad(ops, case):
    """ssd_conv2d_bwd_data_wgrad_first (k_conv3x3_c64b<DGRAD, W0>): block1_conv2's data gradient multiplied with the image
    patch in its store stage -- the gradient w.r.t. block1_conv1's output never reaches memory.  Against (a) the two separate
    calls it replaces (same bf16 rounding of the intermediate: only the fp32 summation order differs) and (b) the fp32
    reference of the composition.  Cases: the real map (ragged right / bottom blocks, several blocks per workgroup), an odd
    map, a single block, and more blocks than the 512 persistent workgroups at a small size; bitwise reproducible."""
    B, H, W = case
    g = torch.Generator().manual_seed(31 + H)
    img = torch.zeros((B, H, W, 8)).bfloat16()
    img[..., :3] = torch.randn((B, H, W, 3), generator=g).bfloat16()
    w0 = torch.zeros((64, 3, 3, 8)).bfloat16()
    w0[..., :3] = (torch.randn((64, 3, 3, 3), generator=g) / np.sqrt(27)).bfloat16()
    b0 = torch.randn((64,), generator=g) * 0.1
    w1 = (torch.randn((64, 3, 3, 64), generator=g) / 24).bfloat16()
    dy = torch.randn((B, H, W, 64), generator=g).bfloat16()
    imd, w1d, dyd = img.cuda(), w1.cuda(), dy.cuda()
    bits = torch.empty((B, H, W, 8), dtype=torch.uint8, device="cuda")
    a1 = ops.conv2d_fwd_relubits(imd, w0.cuda(), b0.cuda(), 1, 1, 1, H, W, bits)
    w1_t = ops.weight_transpose(w1d, 64)
    dw, db = ops.conv2d_bwd_data_wgrad_first(dyd, w1_t, bits, imd)
    torch.cuda.synchronize()
    # (a) the two calls
    dx = ops.conv2d_bwd_data_bits(dyd, w1_t, bits, (B, H, W, 64), 1, 1, 1)
    dw2, db2 = ops.conv2d_bwd_weight(imd, dx, 64, 3, 1, 1, 1)
    sw, sb = max(1.0, dw2.abs().max().item()), max(1.0, db2.abs().max().item())
    assert (dw - dw2).abs().max().item() <= 2e-5 * sw * np.sqrt(B * H * W / 256), "vs the separate calls"
    assert (db - db2).abs().max().item() <= 2e-5 * sb * np.sqrt(B * H * W / 256)
    assert torch.count_nonzero(dw[..., 3:]).item() == 0, "pad channels carry no gradient"
    # (b) fp32 reference: dX = conv_transpose(dy, w1) * (a1 > 0), rounded to bf16 as the kernels hold it; dW0 = corr(image, dX)
    if B * H * W <= 200000:
        a1r = a1.float().cpu()
        xr = a1r.clone().requires_grad_(True)
        ref_conv(xr, w1.float(), None, 3, 1, 1, 1, H, W, False).backward(dy.float())
        dxr = (xr.grad * (a1r > 0)).bfloat16().float()
        wr = w0.float().requires_grad_(True)
        br = b0.clone().requires_grad_(True)
        ref_conv(img.float(), wr, br, 3, 1, 1, 1, H, W, False).backward(dxr)
        # (a bf16 rounding of dX that falls the other way in the kernel moves one term by 2^-8 of its value)
        assert (dw.cpu() - wr.grad).abs().max().item() <= 2e-3 * max(1.0, wr.grad.abs().max().item())
        assert (db.cpu() - br.grad).abs().max().item() <= 2e-3 * max(1.0, br.grad.abs().max().item())
    dw3, db3 = ops.conv2d_bwd_data_wgrad_first(dyd, w1_t, bits, imd)
    assert torch.equal(dw, dw3) and torch.equal(db, db3)
