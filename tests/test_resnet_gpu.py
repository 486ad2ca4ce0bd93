"""GPU: BASELINE configs[4]'s network -- SSD512 on a ResNet-50 trunk (resnet_engine.ResNet50SSDEngine) -- forward and backward
against the plain-PyTorch fp32 restatement (oracle/net_oracle.py:forward_graph) at batch 2, plus the element-wise / pooling
kernels it adds (csrc/eltwise.hip) against torch.  There is NO reference counterpart (the reference hard-codes its 300 x 300 VGG
network, models/ssd_model.py:46,75-97): parity is vs this build's own restatement, UNPINNED.  Tolerances as for the VGG engine:
heads see exact inputs (1e-2), trunk gradients pass through ~50 ReLU masks evaluated on bf16-rounded activations."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import torch.nn.functional as F                                      # noqa: E402
from oracle import net_oracle as N                                   # noqa: E402


def rel_l2(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def ops():
    import ssd_object_detection_amd.ops as ops_
    return ops_


def test_add_relu_and_its_gradient(ops):
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.randn((3, 17, 19, 64), generator=g, device="cuda").bfloat16()
    b = torch.randn((3, 17, 19, 64), generator=g, device="cuda").bfloat16()
    out = ops.add_relu_fwd(a, b)
    assert torch.equal(out, (a.float() + b.float()).relu().bfloat16())
    gr = torch.randn(a.shape, generator=g, device="cuda").bfloat16()
    base = torch.randn(a.shape, generator=g, device="cuda").bfloat16()
    assert torch.equal(ops.relu_mask_bwd(gr, out), gr * (out > 0))
    acc = base.clone()
    ops.relu_mask_bwd(gr, out, out=acc, accumulate=True)
    assert torch.equal(acc, (base.float() + (gr * (out > 0)).float()).bfloat16())


@pytest.mark.parametrize("H,W", [(64, 64), (37, 50), (9, 9)])
def test_maxpool3x3s2(ops, H, W):
    B, C = 2, 64
    g = torch.Generator().manual_seed(H)
    x = torch.randn((B, H, W, C), generator=g).relu().bfloat16()       # post-ReLU (ties at 0, exact zeros)
    y, code = ops.maxpool3x3s2_fwd(x.cuda())
    Ho, pt = ops.same_pad(H, 3, 2)
    Wo, pl = ops.same_pad(W, 3, 2)
    pb, pr = max((Ho - 1) * 2 + 3 - H, 0) - pt, max((Wo - 1) * 2 + 3 - W, 0) - pl
    xr = x.float().requires_grad_(True)
    yr = F.max_pool2d(F.pad(xr.permute(0, 3, 1, 2), (pl, pr, pt, pb), value=float("-inf")), 3, 2)
    assert yr.shape[2:] == (Ho, Wo)
    assert torch.equal(y.float().cpu(), yr.detach().permute(0, 2, 3, 1))
    dy = torch.randn((B, Ho, Wo, C), generator=g).bfloat16()
    dx = ops.maxpool3x3s2_bwd(code, dy.cuda(), x.shape).float().cpu()
    yr.backward(dy.float().permute(0, 3, 1, 2))
    want = (xr.grad * (x.float() > 0))                                   # winners that are 0 carry no gradient (ReLU in front)
    # torch routes a tie to the first maximum in scan order as well; sums of up to four bf16 gradients round once here
    assert (dx - want).abs().max().item() <= 2 ** -7 * max(1.0, want.abs().max().item())
    assert torch.equal(dx * (x.float() > 0), dx)


@pytest.fixture(scope="module")
def engine():
    from ssd_object_detection_amd.resnet_engine import ResNet50SSDEngine
    return ResNet50SSDEngine(classes=81, seed=5)


def test_resnet50_ssd512_plan(engine):
    assert engine.A == 24564 and engine.grids == ((64, 64), (32, 32), (16, 16), (8, 8), (4, 4), (2, 2), (1, 1))     # BASELINE configs[4]
    convs = [nd for nd in engine.nodes if nd["kind"] == "conv"]
    assert len(convs) == 1 + (3 + 4 + 6) * 3 + 3 + 10                   # stem, 13 bottlenecks x 3, 3 projections, 5 extra stages x 2
    assert sum(nd["kind"] == "add" for nd in engine.nodes) == 13
    assert [c for _, _, c in engine.fm] == [512, 1024, 512, 256, 256, 256, 256]


def test_resnet50_ssd512_forward_backward_vs_oracle(engine):
    import ssd_object_detection_amd.ops as ops
    from ssd_object_detection_amd.engine import SSD512_NUM_PRIORS
    B = 2
    g = torch.Generator().manual_seed(3)
    x = ops.image_prep(torch.rand((B, 512, 512, 3), generator=g).cuda())
    loc, conf = engine.forward(x)
    host, host32 = engine.param_bf16.float().cpu(), engine.param.cpu()
    arrays = [t for i in sorted(engine.conv_params) for t in engine.conv_params[i]] + [t for pair in engine.head_params for t in pair]
    params = {}
    for t in arrays:
        src = host if t.name.endswith("kernel") else host32
        params[t.name] = src[t.offset:t.offset + t.numel].view(t.shape).clone().requires_grad_(True)
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    loc_r, conf_r = N.forward_graph(engine.graph, SSD512_NUM_PRIORS, 81, params, x.float().cpu())
    assert loc.shape == (B, 24564, 4) and conf.shape == (B, 24564, 81)
    assert rel_l2(loc.float().cpu(), loc_r.detach()) < 2e-2
    assert rel_l2(conf.float().cpu(), conf_r.detach()) < 2e-2
    dloc = (torch.randn((B, 24564, 4), generator=g) * 1e-3).bfloat16()
    dconf = (torch.randn((B, 24564, 81), generator=g) * 1e-3).bfloat16()
    engine.backward(dloc.cuda(), dconf.cuda())
    (loc_r * dloc.float()).sum().add((conf_r * dconf.float()).sum()).backward()
    gflat = engine.grad.cpu()
    worst = 0.0
    for t in arrays:
        got = gflat[t.offset:t.offset + t.numel].view(t.shape)
        want = params[t.name].grad
        err = rel_l2(got, want)
        cos = float((got * want).sum() / (got.norm() * want.norm() + 1e-30))
        worst = max(worst, err)
        print("%-16s rel L2 err %.4f  cos %.5f" % (t.name, err, cos))
        if t.name.startswith("head"):
            tol, cmin = 2e-2, 0.999                       # (the deep levels read activations 50 bf16 roundings away from the oracle's)
        else:
            # the last stages see 2 x 2 and 1 x 1 maps at batch 2 (8 and 2 pixels): one ReLU whose bf16-rounded input lands on the
            # other side of zero flips a sizeable part of such a layer's gradient; the wide maps average this out
            hout = engine.nodes[int(t.name[4:].split("/")[0])]["hout"]
            tol, cmin = (0.2, 0.975) if hout >= 8 else (0.45, 0.9)
        assert err < tol and cos > cmin, (t.name, err, cos)
    print("worst relative gradient error", worst)
    # one optimizer step moves the weights and keeps the copies consistent
    p0 = engine.param.clone()
    engine.clip_scales(0.01)
    engine.adam(1e-3, engine.grad, 1.0, True)
    torch.cuda.synchronize()
    assert 0 < float((engine.param - p0).abs().max()) < 2e-3
