"""GPU parity of ssd_conv_chain (csrc/chain.hip): the reference's "extras" behind the 19x19 map (models/ssd_model.py:124-150:
1x1 512->128, 3x3/2 128->256, 1x1 256->128, 3x3 valid 128->256, 1x1 256->128, 3x3 valid 128->256 on 10x10 ... 1x1 maps), forward
and data gradient, as ONE launch with one workgroup per image -- against
  (a) the plain PyTorch fp32 restatement of every layer on the chain's own bf16 inputs (Keras Conv2D + ReLU / its tape.gradient;
      the network oracle is UNPINNED: TensorFlow is not installable here), bound 2^-7 of the layer's largest value (one bf16
      rounding of an fp32 sum), and
  (b) the per-layer kernels (ssd_conv2d_fwd_relubits / ssd_conv2d_bwd_data[_bits]) on the same inputs: same bound (the chain sums k
      in one pass, the per-layer kernels in split-K partials), sign bits identical to the stored activations' signs.
Bitwise reproducible; refusal of shapes it does not serve."""
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import torch.nn.functional as F                                      # noqa: E402

# (cin, cout, k, stride, mode) of trunk nodes 17-22 (engine.SSD300_TRUNK); input 10x10x512
EXTRAS = [(512, 128, 1, 1, "same"), (128, 256, 3, 2, "same"), (256, 128, 1, 1, "same"), (128, 256, 3, 1, "valid"),
          (256, 128, 1, 1, "same"), (128, 256, 3, 1, "valid")]


@pytest.fixture(scope="module")
def ops():
    import ssd_object_detection_amd.ops as ops_
    return ops_


def geometry(ops, h, spec):
    out = []
    for cin, cout, k, s, mode in spec:
        if mode == "same":
            ho, pt = ops.same_pad(h, k, s)
        else:
            ho, pt = ops.valid_out(h, k, s), 0
        out.append(dict(cin=cin, cout=cout, k=k, s=s, pt=pt, hin=h, hout=ho))
        h = ho
    return out


def make_net(ops, B, h0, spec, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    geo = geometry(ops, h0, spec)
    x = torch.randn((B, h0, h0, spec[0][0]), generator=g, device="cuda").relu().bfloat16()
    ws = [(torch.randn((d["cout"], d["k"], d["k"], d["cin"]), generator=g, device="cuda") * (2.0 / (d["k"] * d["k"] * d["cin"])) ** 0.5).bfloat16()
          for d in geo]
    bs = [0.1 * torch.randn((d["cout"],), generator=g, device="cuda") for d in geo]
    return geo, x, ws, bs


def run_fwd_chain(ops, B, x, geo, ws, bs):
    outs = [torch.full((B, d["hout"], d["hout"], d["cout"]), 7.0, dtype=torch.bfloat16, device="cuda") for d in geo]
    bits = [torch.full((B, d["hout"], d["hout"], d["cout"] // 8), 0xAA, dtype=torch.uint8, device="cuda") for d in geo]
    packed = ops.chain_pack_weights([(w, None) for w in ws])
    layers = [ops.chain_layer_fwd(w, pk, b, o, d["s"], d["pt"], d["pt"], relu=True, relu_bits=rb)
              for d, w, pk, b, o, rb in zip(geo, ws, packed, bs, outs, bits)]
    ops.conv_chain(x, layers)
    torch.cuda.synchronize()
    return outs, bits


def bits_of(t):
    B, H, W, C = t.shape
    b8 = (t > 0).view(B, H, W, C // 8, 8).to(torch.uint8)
    return (b8 * (2 ** torch.arange(8, device=t.device, dtype=torch.uint8))).sum(-1).to(torch.uint8)


@pytest.mark.parametrize("B", [1, 5])
def test_forward_chain_vs_fp32_and_per_layer_kernels(ops, B):
    geo, x, ws, bs = make_net(ops, B, 10, EXTRAS, 11 + B)
    outs, bits = run_fwd_chain(ops, B, x, geo, ws, bs)
    inp = x
    for d, w, b, o, rb in zip(geo, ws, bs, outs, bits):
        xr = inp.float().permute(0, 3, 1, 2)
        pad_hi = max((d["hout"] - 1) * d["s"] + d["k"] - d["hin"] - d["pt"], 0)
        ref = F.conv2d(F.pad(xr, (d["pt"], pad_hi, d["pt"], pad_hi)), w.float().permute(0, 3, 1, 2), b, stride=d["s"]).relu().permute(0, 2, 3, 1)
        assert ref.shape == o.shape
        tol = 2.0 ** -7 * float(ref.abs().max()) + 1e-12
        assert float((o.float() - ref).abs().max()) <= tol, (d, float((o.float() - ref).abs().max()), tol)
        ker = ops.conv2d_fwd(inp, w, b, d["s"], d["pt"], d["pt"], d["hout"], d["hout"], True)
        assert float((o.float() - ker.float()).abs().max()) <= tol
        assert torch.equal(rb, bits_of(o)), d
        inp = o                                              # the next layer is checked on the chain's own input


def run_dgrad_chain(ops, B, geo, ws, acts, g_last, heads, use_bits):
    """Data gradients from the last layer back to the chain's input; heads[i] (or None) = a gradient already sitting in the
    map that layer i's data gradient is accumulated onto (a head convolution's)."""
    n = len(geo)
    gouts, layers = [None] * n, []
    for i in range(n - 1, -1, -1):
        d = geo[i]
        go = heads[i].clone() if heads[i] is not None else torch.full((B, d["hin"], d["hin"], d["cin"]), 7.0, dtype=torch.bfloat16, device="cuda")
        gouts[i] = go
        wt = ops.weight_transpose(ws[i])
        layers.append(ops.chain_layer_dgrad(wt, ops.chain_pack_weights([(wt, None)])[0], go, d["s"], d["pt"], d["pt"], accumulate=heads[i] is not None,
                                            mask_bits=bits_of(acts[i]) if use_bits else None, mask_src=None if use_bits else acts[i]))
    ops.conv_chain(g_last, layers)
    torch.cuda.synchronize()
    return gouts


@pytest.mark.parametrize("B,use_bits", [(1, True), (4, True), (3, False)])
def test_dgrad_chain_vs_fp32_and_per_layer_kernels(ops, B, use_bits):
    geo, x, ws, bs = make_net(ops, B, 10, EXTRAS, 40 + B)
    outs, _ = run_fwd_chain(ops, B, x, geo, ws, bs)
    acts = [x] + outs[:-1]                                   # input activation of layer i
    g = torch.Generator(device="cuda").manual_seed(77)
    g_last = (torch.randn(outs[-1].shape, generator=g, device="cuda") * 0.01).bfloat16()
    # feature maps (inputs of layers 0, 2, 4: the 10x10, 5x5 and 3x3 maps) already hold a masked head gradient
    heads = [((torch.randn(a.shape, generator=g, device="cuda") * 0.01) * (a > 0)).bfloat16() if i % 2 == 0 else None
             for i, a in enumerate(acts)]
    gouts = run_dgrad_chain(ops, B, geo, ws, acts, g_last, heads, use_bits)
    gin = g_last
    for i in range(len(geo) - 1, -1, -1):
        d = geo[i]
        xr = acts[i].float().permute(0, 3, 1, 2).clone().requires_grad_(True)
        pad_hi = max((d["hout"] - 1) * d["s"] + d["k"] - d["hin"] - d["pt"], 0)
        y = F.conv2d(F.pad(xr, (d["pt"], pad_hi, d["pt"], pad_hi)), ws[i].float().permute(0, 3, 1, 2), None, stride=d["s"])
        y.backward(gin.float().permute(0, 3, 1, 2))
        ref = xr.grad.permute(0, 2, 3, 1)
        if heads[i] is not None:
            ref = ref + heads[i].float()
        ref = ref * (acts[i] > 0)
        tol = 2.0 ** -7 * float(ref.abs().max()) + 1e-12
        assert float((gouts[i].float() - ref).abs().max()) <= tol, (i, float((gouts[i].float() - ref).abs().max()), tol)
        ker = heads[i].clone() if heads[i] is not None else None
        ker = ops.conv2d_bwd_data(gin, ops.weight_transpose(ws[i]), acts[i], tuple(acts[i].shape), d["s"], d["pt"], d["pt"],
                                  accumulate=ker is not None, out=ker)
        assert float((gouts[i].float() - ker.float()).abs().max()) <= tol
        gin = gouts[i]


def test_bitwise_reproducible_and_other_geometry(ops):
    """8x8 -> 4x4 -> 2x2 -> 1x1 (the SSD512 recipe's far end) and two runs of the same chain."""
    spec = [(256, 128, 1, 1, "same"), (128, 256, 3, 2, "same"), (256, 128, 1, 1, "same"), (128, 256, 3, 2, "same"),
            (256, 128, 1, 1, "same"), (128, 256, 3, 2, "same")]
    B = 3
    geo, x, ws, bs = make_net(ops, B, 8, spec, 5)
    o1, b1 = run_fwd_chain(ops, B, x, geo, ws, bs)
    o2, b2 = run_fwd_chain(ops, B, x, geo, ws, bs)
    for a, b in zip(o1 + b1, o2 + b2):
        assert torch.equal(a.view(torch.uint8) if a.dtype == torch.uint8 else a.view(torch.int16), b.view(torch.uint8) if b.dtype == torch.uint8 else b.view(torch.int16))
    inp = x
    for d, w, b, o in zip(geo, ws, bs, o1):
        ker = ops.conv2d_fwd(inp, w, b, d["s"], d["pt"], d["pt"], d["hout"], d["hout"], True)
        assert float((o.float() - ker.float()).abs().max()) <= 2.0 ** -7 * float(ker.float().abs().max()) + 1e-12
        inp = o


def test_refusals(ops):
    B = 2
    x = torch.zeros((B, 19, 19, 256), dtype=torch.bfloat16, device="cuda")
    w = torch.zeros((128, 1, 1, 256), dtype=torch.bfloat16, device="cuda")
    out = torch.zeros((B, 19, 19, 128), dtype=torch.bfloat16, device="cuda")
    with pytest.raises(NotImplementedError):                 # 361 pixels: not an LDS-resident map
        ops.conv_chain(x, [ops.chain_layer_fwd(w, w, None, out, 1, 0, 0)])
    x = torch.zeros((B, 4, 4, 64), dtype=torch.bfloat16, device="cuda")
    w = torch.zeros((128, 1, 1, 64), dtype=torch.bfloat16, device="cuda")
    out = torch.zeros((B, 4, 4, 128), dtype=torch.bfloat16, device="cuda")
    with pytest.raises(NotImplementedError):                 # 64 input channels: not a multiple of 128
        ops.conv_chain(x, [ops.chain_layer_fwd(w, w, None, out, 1, 0, 0)])
    x = torch.zeros((B, 4, 4, 128), dtype=torch.bfloat16, device="cuda")
    w = torch.zeros((128, 1, 1, 128), dtype=torch.bfloat16, device="cuda")
    w2 = torch.zeros((128, 1, 1, 256), dtype=torch.bfloat16, device="cuda")
    with pytest.raises(AssertionError):                      # the second layer does not read what the first one writes
        ops.conv_chain(x, [ops.chain_layer_fwd(w, w, None, out, 1, 0, 0), ops.chain_layer_fwd(w2, w2, None, out.clone(), 1, 0, 0)])


def test_batched_weight_gradients_equal_the_separate_calls(ops):
    """ssd_conv2d_bwd_weight_batched: the six extras' weight gradients (+ bias gradients) in two launches, bit for bit what six
    ssd_conv2d_bwd_weight calls write; a layer of another kernel family is refused with nothing launched."""
    B = 16
    geo, x, ws, bs = make_net(ops, B, 10, EXTRAS, 91)
    outs, _ = run_fwd_chain(ops, B, x, geo, ws, bs)
    acts = [x] + outs[:-1]
    g = torch.Generator(device="cuda").manual_seed(92)
    dys = [(torch.randn(o.shape, generator=g, device="cuda") * 0.01).bfloat16() for o in outs]
    sep = [ops.conv2d_bwd_weight(a, dy, d["cout"], d["k"], d["s"], d["pt"], d["pt"]) for a, dy, d in zip(acts, dys, geo)]
    dws = [torch.full_like(w_, 7.0) for w_, _ in sep]
    dbs = [torch.full_like(b_, 7.0) for _, b_ in sep]
    ops.conv2d_bwd_weight_batched([(a, dy, d["cout"], d["k"], d["s"], d["pt"], d["pt"], dw, db)
                                   for a, dy, d, dw, db in zip(acts, dys, geo, dws, dbs)])
    torch.cuda.synchronize()
    for (w_, b_), dw, db in zip(sep, dws, dbs):
        assert torch.equal(w_, dw) and torch.equal(b_, db)
    xb = torch.zeros((B, 19, 19, 1024), dtype=torch.bfloat16, device="cuda")
    dyb = torch.zeros((B, 19, 19, 1024), dtype=torch.bfloat16, device="cuda")
    with pytest.raises(NotImplementedError):                 # the 256-wide tile kernel's case
        ops.conv2d_bwd_weight_batched([(xb, dyb, 1024, 1, 1, 0, 0, torch.zeros((1024, 1, 1, 1024), device="cuda"), torch.zeros(1024, device="cuda"))])
