"""GPU: the whole SSD300 network (forward, backward, optimizer step) on the HIP engine vs the plain-PyTorch
oracle (oracle/net_oracle.py) and the numpy optimizer oracle.  Network parity is UNPINNED by the reference
(no TensorFlow here); tolerances are bf16 storage noise: relative L2 error."""
import math

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import ssd_oracle as O                                   # noqa: E402
from oracle import net_oracle as N                                   # noqa: E402


def rel_l2(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def engine():
    from ssd_object_detection_amd.engine import SSDEngine
    return SSDEngine(classes=81, seed=3)


def gemm_arrays(engine):
    """The arrays the network computes with, by name: conv{i}/kernel|bias and the per-level fused head{l}/kernel|bias
    ([loc filters ; conf filters], one GEMM) -- NOT the clip/optimizer partition (engine.tensors)."""
    out = [t for i in sorted(engine.conv_params) for t in engine.conv_params[i]]
    return out + [t for pair in engine.head_params for t in pair]


def reference_variables(engine):
    """The reference's trainable variables as (name, flat offset, numel), built from its layer list and NOT from
    engine.tensors: one kernel + one bias per Conv2D of models/ssd_model.py:77-151 (20 layers) and, per feature level,
    a loc Conv2D and a conf Conv2D with kernel + bias each (:155-162) -> 64 variables, each clipped on its own (:249).
    Locations come from the GEMM layout: rows [0, 4n) of a level's fused filter array are the loc layer's."""
    from ssd_object_detection_amd.engine import SSD300_TRUNK, SSD300_NUM_PRIORS
    out = []
    for i, (kind, cin, cout, k, _, _, _) in enumerate(SSD300_TRUNK):
        if kind != "conv":
            continue
        wt, bt = engine.conv_params[i]
        out += [("conv%d/kernel" % i, wt.offset, cout * k * k * cin), ("conv%d/bias" % i, bt.offset, cout)]
    for lvl, n in enumerate(SSD300_NUM_PRIORS):
        wt, bt = engine.head_params[lvl]
        c = wt.shape[-1]
        nl, nc = n * 4, n * engine.classes
        out += [("loc%d/kernel" % lvl, wt.offset, nl * 9 * c), ("loc%d/bias" % lvl, bt.offset, nl),
                ("conf%d/kernel" % lvl, wt.offset + nl * 9 * c, nc * 9 * c), ("conf%d/bias" % lvl, bt.offset + nl, nc)]
    return out


def oracle_params(engine, requires_grad=False):
    p = {}
    host = engine.param_bf16.float().cpu()
    host_f32 = engine.param.cpu()
    for t in gemm_arrays(engine):
        src = host if t.name.endswith("kernel") else host_f32          # biases are applied in fp32
        p[t.name] = src[t.offset:t.offset + t.numel].view(t.shape).clone().requires_grad_(requires_grad)
    return p


def test_static_plan(engine):
    assert engine.A == 8732 and engine.level_off == [0, 5776, 7942, 8542, 8692, 8728, 8732]
    assert engine.grids == ((38, 38), (19, 19), (10, 10), (5, 5), (3, 3), (1, 1))
    # 25 128 118 parameters in the reference + the 5 zero-padded input channels of conv0
    assert engine.n_params == 25128118 + 64 * 3 * 3 * 5
    # the clip / optimizer partition is the reference's variable list: 64 tensors (models/ssd_model.py:155-162, :249)
    ref = reference_variables(engine)
    assert len(engine.tensors) == len(ref) == 64
    assert sorted((t.offset, t.numel) for t in engine.tensors) == sorted((o, n) for _, o, n in ref)
    bt = engine.block_tensor.cpu().numpy()
    for t in engine.tensors:                                           # every optimizer block belongs to one variable
        b0, b1 = t.offset // engine.block, (t.offset + t.numel - 1) // engine.block + 1
        assert (bt[b0:b1] == t.index).all() and (bt == t.index).sum() == b1 - b0


def test_forward_backward_vs_oracle(engine):
    import ssd_object_detection_amd.ops as ops
    from ssd_object_detection_amd.engine import SSD300_TRUNK, SSD300_NUM_PRIORS
    B = 2
    g = torch.Generator().manual_seed(1)
    img = torch.rand((B, 300, 300, 3), generator=g)
    x = ops.image_prep(img.cuda())
    loc, conf = engine.forward(x)
    params = oracle_params(engine, requires_grad=True)
    loc_r, conf_r = N.forward(SSD300_TRUNK, SSD300_NUM_PRIORS, 81, params, x.float().cpu())
    assert loc.shape == (B, 8732, 4) and conf.shape == (B, 8732, 81)
    assert rel_l2(loc.float().cpu(), loc_r.detach()) < 1e-2
    assert rel_l2(conf.float().cpu(), conf_r.detach()) < 1e-2
    # backward from a synthetic upstream gradient
    dloc = (torch.randn((B, 8732, 4), generator=g) * 1e-3).bfloat16()
    dconf = (torch.randn((B, 8732, 81), generator=g) * 1e-3).bfloat16()
    engine.backward(dloc.cuda(), dconf.cuda())
    (loc_r * dloc.float()).sum().add((conf_r * dconf.float()).sum()).backward()
    gflat = engine.grad.cpu()
    worst = 0.0
    for t in gemm_arrays(engine):
        got = gflat[t.offset:t.offset + t.numel].view(t.shape)
        want = params[t.name].grad
        if t.name == "conv0/kernel":
            assert float(got[..., 3:].abs().max()) == 0.0            # padded input channels never get gradient
        err = rel_l2(got, want)
        cos = float((got * want).sum() / (got.norm() * want.norm() + 1e-30))
        worst = max(worst, err)
        print("%-16s rel L2 err %.4f  cos %.5f" % (t.name, err, cos))
        # Heads see exact inputs: tight.  Trunk gradients pass through up to 22 ReLU masks evaluated on
        # activations that differ from the oracle's by bf16 rounding; an activation that lands on the other side
        # of zero flips a whole gradient element, so the end-to-end bound is loose (the tight per-op bounds are
        # in test_conv_gpu.py).
        assert err < (1e-2 if t.name.startswith("head") else 0.15) and cos > 0.985, (t.name, err, cos)
    print("worst relative gradient error", worst)


def test_two_stream_schedule_is_bitwise_neutral(engine):
    """The side-stream schedule (weight gradients and the large heads next to the main chain) must not change a single
    bit: every launch sums in a fixed order and the only shared accumulation target is ordered by events.  Run at a
    batch where kernels of both streams really share the GPU (this is what exposed a missing LDS-read wait once)."""
    import ssd_object_detection_amd.ops as ops
    B = 8
    g = torch.Generator().manual_seed(17)
    x = ops.image_prep(torch.rand((B, 300, 300, 3), generator=g).cuda())
    dloc = (torch.randn((B, 8732, 4), generator=g) * 1e-3).bfloat16().cuda()
    dconf = (torch.randn((B, 8732, 81), generator=g) * 1e-3).bfloat16().cuda()

    def run(overlap):
        engine.overlap_heads = overlap
        engine.grad.zero_()
        loc, conf = engine.forward(x)
        engine.backward(dloc, dconf)
        torch.cuda.synchronize()
        return engine.grad.clone(), loc.clone(), conf.clone()

    saved = engine.overlap_heads
    try:
        ref = run(False)
        for _ in range(4):
            got = run(True)
            assert torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2])
            assert torch.equal(got[0], ref[0])
    finally:
        engine.overlap_heads = saved


def test_optimizer_vs_oracle(engine):
    """clip_by_norm per variable + Adam on the flat buffer vs the numpy oracle (Keras formulas), with the oracle's
    partition taken from the reference's layer list (reference_variables), not from the engine's own table."""
    g = torch.Generator().manual_seed(2)
    engine.init_params(seed=3)
    p0 = engine.param.cpu().numpy().copy()
    grad = torch.zeros(engine.n_flat)
    ref = reference_variables(engine)
    for i, (name, off, numel) in enumerate(ref):
        scale = 10.0 ** (-(i % 5))                                    # some variables above, some below the 0.01 clip norm
        grad[off:off + numel] = torch.randn(numel, generator=g) * scale / math.sqrt(numel)
    engine.grad.copy_(grad)
    engine.clip_scales(0.01)
    by_offset = {t.offset: t.index for t in engine.tensors}
    norms = engine.grad_norms.cpu().numpy()
    scales = engine.clip_scale.cpu().numpy()
    m = np.zeros_like(p0); v = np.zeros_like(p0); p = p0.astype(np.float64)
    gnp = grad.numpy().astype(np.float64)
    for step in (1, 2):
        engine.adam(1e-3, engine.grad, grad_scale=0.5, use_clip_scale=True)
        for name, off, numel in ref:
            sl = slice(off, off + numel)
            gt = gnp[sl]
            i = by_offset[off]
            assert abs(norms[i] - np.linalg.norm(gt)) <= 1e-5 * np.linalg.norm(gt), name
            clipped = O.clip_by_norm(gt, 0.01) * 0.5
            assert abs(scales[i] - 0.01 / max(np.linalg.norm(gt), 0.01)) < 1e-5, name
            p[sl], m_, v_ = O.adam_step(p[sl], clipped, m[sl], v[sl], step, 1e-3)
            m[sl], v[sl] = m_, v_
        got = engine.param.cpu().numpy()
        assert np.abs(got - p).max() <= 2e-6, step
    bf = engine.param_bf16.float().cpu().numpy()
    assert np.abs(bf - got).max() <= 2 ** -8 * np.abs(got).max()
    engine.init_params(seed=3)


def test_loc_and_conf_heads_are_clipped_separately(engine):
    """||g_loc|| << ||g_conf|| on every level: the reference clips the loc and conf layers' variables on their own
    (models/ssd_model.py:155-162, :249), so BOTH come out at norm 0.01 -- clipping their concatenation would leave the loc
    part far below it.  Checked on the in-place clip (the data-parallel path) and on the micro-batch accumulation."""
    engine.grad.zero_()
    g = torch.Generator().manual_seed(9)
    ref = [r for r in reference_variables(engine) if r[0].startswith(("loc", "conf"))]
    for name, off, numel in ref:
        target = 0.05 if name.startswith("loc") else 50.0                 # both above the clip norm, 1000x apart
        v = torch.randn(numel, generator=g)
        engine.grad[off:off + numel] = (v * (target / v.norm())).cuda()
    g0 = engine.grad.clone()
    engine.clip_scales(0.01)
    engine.accumulate_clipped(first=True)
    t0 = engine.head_params[0][0].index
    engine.clip_range_in_place(t0, len(engine.tensors), 0.01)
    torch.cuda.synchronize()
    for buf in (engine.grad, engine.grad_acc):
        for name, off, numel in ref:
            nrm = float(buf[off:off + numel].double().norm())
            assert abs(nrm - 0.01) < 1e-6, (name, nrm)
    assert torch.equal(engine.grad, engine.grad_acc)
    engine.grad.copy_(g0)
    engine.init_params(seed=3)


def test_full_batch_is_image_independent(engine):
    """Size-independent property at the benchmark's full size (batch 64): a sample's predictions and the gradient that
    reaches its activations do not depend on where it sits in the batch or on its neighbours -- blocks of the convolution
    kernels span image boundaries (row strips, strip blocks, parity classes), split-K and three streams are in play, and
    none of that may leak between images: permuting the batch permutes predictions and activation gradients bit for bit
    (same kernels, same tiling), and leaves the weight gradient unchanged up to fp32 summation order."""
    import ssd_object_detection_amd.ops as ops
    B = 64
    g = torch.Generator(device="cuda").manual_seed(77)
    img = torch.rand((B, 300, 300, 3), generator=g, device="cuda")
    x = ops.image_prep(img)
    dloc = (torch.randn((B, 8732, 4), generator=g, device="cuda") * 1e-3).bfloat16()
    dconf = (torch.randn((B, 8732, 81), generator=g, device="cuda") * 1e-3).bfloat16()
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(5)).cuda()

    def run(xb, dl, dc):
        loc, conf = engine.forward(xb)
        loc, conf = loc.clone(), conf.clone()
        engine.backward(dl, dc)
        torch.cuda.synchronize()
        return loc, conf, engine._acts(B)["gacts"][1].clone(), engine.grad.clone()

    loc, conf, g1, grad = run(x, dloc, dconf)
    loc_p, conf_p, g1_p, grad_p = run(x[perm].contiguous(), dloc[perm].contiguous(), dconf[perm].contiguous())
    assert torch.equal(loc_p, loc[perm]) and torch.equal(conf_p, conf[perm])
    assert torch.equal(g1_p, g1[perm])                       # gradient w.r.t. block1_conv1's output, per image
    err = float((grad_p - grad).norm() / grad.norm())
    assert err < 1e-3, err                                   # the same products summed in another order


def test_dense_head_backward_matches_the_sparse_path():
    """engine.backward's dense head branch (sparse_heads=False: gradient packing, head data / weight gradients on the dense
    kernels, the large levels on the side stream with event-ordered accumulation) against the shipped sparse-row branch on the
    same loss gradient: the head gradients come from the same products in another order (1e-3), the trunk gradients differ by
    one bf16 rounding of each feature-map gradient.  Keeps the 70 lines of the fallback schedule from rotting untested."""
    import ssd_object_detection_amd.ops as ops
    from ssd_object_detection_amd.engine import SSDEngine
    from ssd_object_detection_amd.data_loaders.synthetic import synth_batch_gt
    B = 4
    sparse = SSDEngine(classes=81, seed=5, sparse_heads=True)
    dense = SSDEngine(classes=81, seed=5, sparse_heads=False)
    assert sparse.sparse_heads and not dense.sparse_heads and torch.equal(sparse.param, dense.param)
    g = torch.Generator().manual_seed(23)
    x = ops.image_prep(torch.rand((B, 300, 300, 3), generator=g).cuda())
    pset = ops.build_priors()
    cls_l, box_l = synth_batch_gt(300, B)
    targets = ops.match_encode(*ops.pack_gt(box_l, cls_l), pset, 0.5)
    loc_s, conf_s = sparse.forward(x)
    loc_d, conf_d = dense.forward(x)
    assert torch.equal(loc_s, loc_d) and torch.equal(conf_s, conf_d)
    out, dconf, dloc = ops.ssd_loss(conf_d, loc_d, *targets, grad_scale=64.0)
    hgb = sparse.head_grad_buffers(B)
    out_s = ops.ssd_loss_heads(conf_s, loc_s, *targets, hgb, grad_scale=64.0)
    assert torch.equal(out, out_s)
    assert dense.head_grad_buffers(B) is None
    dense.backward(dloc, dconf)
    sparse.backward(None, None, heads=hgb)
    torch.cuda.synchronize()
    gs, gd = sparse.grad.cpu(), dense.grad.cpu()
    for t in gemm_arrays(sparse):
        a, b = gs[t.offset:t.offset + t.numel], gd[t.offset:t.offset + t.numel]
        err = rel_l2(a, b)
        assert err < (2e-3 if t.name.startswith("head") else 2e-2), (t.name, err)
    # the feature-map gradients the two paths hand to the trunk: one bf16 rounding apart
    for ni, _, _ in sparse.fm:
        a = sparse._acts(B)["gacts"][ni + 1].float()
        b = dense._acts(B)["gacts"][ni + 1].float()
        assert (a - b).abs().max() <= 2.0 ** -6 * b.abs().max() + 1e-12, ni


def test_chain_launch_matches_the_per_layer_schedule():
    """The extras' six layers as one launch per direction (ops.conv_chain, engine.chain) against the per-layer kernels: the
    chain sums k in one pass where the per-layer kernels use split-K partial sums, so activations may differ by one bf16
    rounding -- predictions, and every gradient, agree to bf16 accuracy; the chain really is selected for nodes 17-22."""
    import ssd_object_detection_amd.ops as ops
    from ssd_object_detection_amd.engine import SSDEngine
    B = 3
    a = SSDEngine(classes=81, seed=9)
    b = SSDEngine(classes=81, seed=9)
    assert a.chain_start == 17 and a.chain == {"fwd", "bwd"}
    b.chain = set()
    g = torch.Generator().manual_seed(31)
    x = ops.image_prep(torch.rand((B, 300, 300, 3), generator=g).cuda())
    dloc = (torch.randn((B, 8732, 4), generator=g) * 1e-3).bfloat16().cuda()
    dconf = (torch.randn((B, 8732, 81), generator=g) * 1e-3).bfloat16().cuda()
    la, ca = a.forward(x)
    lb, cb = b.forward(x)
    assert a.chain == {"fwd", "bwd"}                         # not refused
    assert rel_l2(la.float(), lb.float()) < 2e-3 and rel_l2(ca.float(), cb.float()) < 2e-3
    assert torch.equal(a._acts(B)["acts"][17], b._acts(B)["acts"][17])     # in front of the chain: the same launch sequence
    for i in range(17, 23):
        ya, yb = a._acts(B)["acts"][i + 1].float(), b._acts(B)["acts"][i + 1].float()
        assert (ya - yb).abs().max() <= 2.0 ** -6 * yb.abs().max() + 1e-12, i
    a.backward(dloc, dconf)
    b.backward(dloc, dconf)
    torch.cuda.synchronize()
    assert a.chain == {"fwd", "bwd"}
    ga, gb = a.grad.cpu(), b.grad.cpu()
    for t in gemm_arrays(a):
        err = rel_l2(ga[t.offset:t.offset + t.numel], gb[t.offset:t.offset + t.numel])
        assert err < 2e-2, (t.name, err)


def test_schedule_switches_are_bitwise_neutral():
    """Every schedule switch the engine keeps (round 4: most of them measured slower and left off) only moves the SAME launches
    between streams, groups them or batches them into launches that run the same blocks: predictions and every gradient must
    equal the default schedule bit for bit.  Batch 48: both large head levels (38x38, 19x19) take the split path."""
    import ssd_object_detection_amd.ops as ops
    from ssd_object_detection_amd.engine import SSDEngine
    B = 48
    eng = SSDEngine(classes=81, seed=11)
    g = torch.Generator().manual_seed(41)
    x = ops.image_prep(torch.rand((B, 300, 300, 3), generator=g).cuda())
    dloc = (torch.randn((B, 8732, 4), generator=g) * 1e-3).bfloat16().cuda()
    dconf = (torch.randn((B, 8732, 81), generator=g) * 1e-3).bfloat16().cuda()

    def run():
        eng.grad.zero_()
        loc, conf = eng.forward(x)
        eng.backward(dloc, dconf)
        torch.cuda.synchronize()
        return eng.grad.clone(), loc.clone(), conf.clone()

    ref = run()
    assert eng.chain == {"fwd", "bwd"} and eng.batch_chain_wgrads          # the default really is the chained / batched schedule
    switches = [("wgrad_group", 1), ("batch_chain_wgrads", False), ("batch_chain_front", True), ("split_heads_dgrad", 0),
                ("split_heads_dgrad", 1), ("prezero_maps", True), ("reduce_stream", True), ("side_streams", 2),
                ("wgrad_on_main", {14}), ("chain_heads_split", False), ("chain_prefetch", False), ("pack_side", False)]
    for name, value in switches:
        saved = getattr(eng, name)
        setattr(eng, name, value)
        try:
            got = run()
        finally:
            setattr(eng, name, saved)
        assert torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2]), name
        assert torch.equal(got[0], ref[0]), name
    assert torch.equal(run()[0], ref[0])


def test_train_step_at_batch_64():
    """BASELINE configs[2] itself, with assertions (the bench's loss_check asserts nothing): one full `_train_step` at batch
    64 on the shipped path (sparse head rows, fused first-layer pair, per-bucket clip + Adam inside the backward pass, three
    streams), then
      * the compact gradient rows the step's loss wrote, scattered back, equal the dense-gradient form of the loss on the
        step's own logits BIT FOR BIT, and so do the loss scalars;
      * the loss scalars agree with the float64 oracle (models/ssd_model.py:341-396 restated) on those logits to 1e-4;
      * weights, both Adam moments, bf16 and transposed copies equal those of the unfused sequence (backward, clip_scales,
        adam on the whole flat buffer, targets assigned up front) bit for bit."""
    from oracle import ssd_oracle as O
    from ssd_object_detection_amd import ops, optimizers
    from ssd_object_detection_amd.data_loaders.synthetic import synth_batch_gt
    from ssd_object_detection_amd.models import SSDObjectDetectionModel
    B = 64
    gen = torch.Generator(device="cuda").manual_seed(64)
    img = torch.rand((B, 300, 300, 3), generator=gen, device="cuda")
    cls_l, box_l = synth_batch_gt(6400, B)
    gt = ops.pack_gt(box_l, cls_l)

    def run(fused):
        model = SSDObjectDetectionModel(classes=80, log_dir="gpurun_out/t64", timestamp_dir=False, seed=2)
        model.fused_optimizer = fused
        opt = optimizers.Adam(1e-3)
        x = ops.image_prep(img, normalize=True)
        if fused:
            tgt = model.match_async(gt)
        else:
            tgt = ops.match_encode(*gt, model._pset, 0.5)
        conf, loc, info = model._train_step(x, *tgt, opt)
        torch.cuda.synchronize()
        return model, tgt, conf, loc, info

    m1, tgt, conf, loc, info = run(True)
    eng = m1.get_engine()
    assert eng.sparse_heads and eng.fuse_first and float(info["status"]) == 0.0
    # rows == dense gradient, scalars equal
    hgb = eng.head_grad_buffers(B)
    out_d, dconf, dloc = ops.ssd_loss(conf, loc, *tgt)
    sl, sc = hgb.dense(81)
    assert torch.equal(sl.view(torch.int16), dloc.view(torch.int16))
    assert torch.equal(sc.view(torch.int16), dconf.view(torch.int16))
    raw = m1._last_raw
    assert torch.equal(raw, out_d)
    P = int(tgt[2].sum())
    assert int(raw[4]) == P and int(raw[5]) >= 3 * P
    assert sum(hgb.count.cpu().tolist()[:6]) <= P + int(raw[5])
    # float64 oracle on the step's own logits
    ref = O.ssd_loss(tgt[0].cpu().numpy(), tgt[1].cpu().numpy(), tgt[2].cpu().numpy(), loc.float().cpu().numpy(),
                     conf.float().cpu().numpy())
    for key, want in (("loc loss", ref["loc"]), ("cls loss pos", ref["pos"]), ("cls loss neg", ref["neg"])):
        got = float(info[key])
        assert abs(got - want) <= 1e-4 * abs(want), (key, got, want)
    assert ref["num_pos"] == P
    # fused per-bucket optimizer == the unfused sequence
    m2, _, conf2, loc2, info2 = run(False)
    assert torch.equal(conf, conf2) and torch.equal(loc, loc2)
    a, b = eng, m2.get_engine()
    assert a.step_count == b.step_count == 1
    for name in ("grad", "param", "adam_m", "adam_v", "param_bf16", "clip_scale", "grad_norms"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    for i in a.w_t:
        assert torch.equal(a.w_t[i], b.w_t[i]), i
    for u, v in zip(a.head_w_t, b.head_w_t):
        assert torch.equal(u, v)
    delta = (a.param - SSDObjectDetectionModel(classes=80, log_dir="gpurun_out/t64", timestamp_dir=False, seed=2).get_engine().param)
    assert 0 < float(delta.abs().max()) < 2e-3                    # Adam's first step: ~lr per touched weight
