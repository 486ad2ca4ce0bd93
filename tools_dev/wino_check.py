"""Dev helper: Winograd kernel vs the fp32 reference (small shapes) and vs the direct kernel (timing, full shapes).
usage: wino_check.py [check] [time]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
import ssd_object_detection_amd.ops as ops

from ssd_object_detection_amd import _lib
L = _lib.lib()
ABLATE = [int(v) for v in os.environ.get("WINO_ABLATE", "").split(",") if v]
torch.manual_seed(0)


def ref(x, w, b, relu):
    y = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), b, padding=1)
    if relu:
        y = y.relu()
    return y.permute(0, 2, 3, 1).contiguous()


def rel(a, b):
    return ((a - b).norm() / b.norm()).item()


def check():
    for (B, H, W, Cin, Cout) in [(1, 16, 16, 64, 64), (2, 20, 37, 64, 64), (3, 19, 19, 64, 128), (2, 33, 18, 128, 192), (1, 75, 75, 128, 64)]:
        x = torch.relu(torch.randn((B, H, W, Cin))).bfloat16()
        w = (torch.randn((Cout, 3, 3, Cin)) / np.sqrt(9 * Cin)).bfloat16()
        b = torch.randn(Cout) * 0.1
        xd, wd, bd = x.cuda(), w.cuda(), b.cuda()
        u = ops.wino_weights(wd)
        for relu in (True, False):
            y = ops.conv3x3_wino_fwd(xd, u, bd, Cout, relu).float().cpu()
            yd = ops.conv2d_fwd(xd, wd, bd, 1, 1, 1, H, W, relu).float().cpu()
            yr = ref(x, w, b, relu)
            print("fwd", (B, H, W, Cin, Cout), "relu", relu, "wino rel %.2e max %.3e | direct rel %.2e max %.3e" % (
                rel(y, yr), (y - yr).abs().max().item(), rel(yd, yr), (yd - yr).abs().max().item()), flush=True)
        # fused pooling
        for mode in ("same", "valid"):
            y, yp, code = ops.conv3x3_wino_fwd(xd, u, bd, Cout, True, pool=mode)
            y2, yp2, code2 = ops.conv2d_fwd_pool(xd, wd, bd, 1, 1, 1, H, W, True, mode == "same")
            yp_ref, code_ref = ops.maxpool2x2_fwd_argmax(y, mode == "same") if hasattr(ops, "maxpool2x2_fwd_argmax") else (None, None)
            if yp_ref is not None:
                print("  pool", mode, "pooled == pool(y):", torch.equal(yp, yp_ref), "codes:", torch.equal(code, code_ref), flush=True)
        # data gradient
        if not ops.wino_supported(B, H, W, Cout, Cin):
            continue
        dy = torch.randn((B, H, W, Cout)).bfloat16()
        wt = ops.weight_transpose(wd)
        ut = ops.wino_weights(wt)
        xr = x.float().requires_grad_(True)
        ref(xr, w, b, False).backward(dy.float())
        dx = ops.conv3x3_wino_bwd_data(dy.cuda(), ut, None, (B, H, W, Cin)).float().cpu()
        dxd = ops.conv2d_bwd_data(dy.cuda(), wt, None, (B, H, W, Cin), 1, 1, 1).float().cpu()
        print("  dgrad wino rel %.2e | direct rel %.2e" % (rel(dx, xr.grad), rel(dxd, xr.grad)), flush=True)
        small = (dy.float() * 2.0 ** -14).bfloat16()
        dx0 = ops.conv3x3_wino_bwd_data(small.cuda(), ut, None, (B, H, W, Cin)).float().cpu() * 2.0 ** 14
        dx1 = ops.conv3x3_wino_bwd_data(small.cuda(), ut, None, (B, H, W, Cin), in_shift=12).float().cpu() * 2.0 ** 14
        print("  dgrad of dy*2^-14: shift 0 rel %.2e | shift 12 rel %.2e" % (rel(dx0, xr.grad), rel(dx1, xr.grad)), flush=True)
        mask = torch.randn((B, H, W, Cin)).bfloat16()
        base = torch.randn((B, H, W, Cin)).bfloat16()
        acc = base.clone().cuda()
        ops.conv3x3_wino_bwd_data(dy.cuda(), ut, mask.cuda(), (B, H, W, Cin), accumulate=True, out=acc)
        want = (xr.grad + base.float()) * (mask.float() > 0)
        print("  dgrad+mask+acc rel %.2e" % rel(acc.float().cpu(), want), flush=True)


def timed(run, reps=10):
    run(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(ts)[2]


def time_():
    for (B, H, Cin, Cout) in [(64, 75, 256, 256), (64, 75, 128, 256), (64, 150, 128, 128), (64, 150, 64, 128), (64, 38, 512, 512), (64, 38, 256, 512)]:
        x = torch.relu(torch.randn((B, H, H, Cin), device="cuda")).bfloat16()
        w = (torch.randn((Cout, 3, 3, Cin), device="cuda") / np.sqrt(9 * Cin)).bfloat16()
        b = torch.zeros(Cout, device="cuda")
        u = ops.wino_weights(w)
        y = ops.conv2d_fwd(x, w, b, 1, 1, 1, H, H, True)
        y2 = ops.conv3x3_wino_fwd(x, u, b, Cout, True)
        r = rel(y2.float(), y.float())
        td = timed(lambda: ops.conv2d_fwd(x, w, b, 1, 1, 1, H, H, True, out=y))
        tw = timed(lambda: ops.conv3x3_wino_fwd(x, u, b, Cout, True, out=y2))
        abl = []
        for a in ABLATE:
            L.ssd_dev_knob(b"SSD_ABLATE", a)
            abl.append("abl%d %.1f" % (a, timed(lambda: ops.conv3x3_wino_fwd(x, u, b, Cout, True, out=y2))))
        for a in [int(v) for v in os.environ.get("DIRECT_ABLATE", "").split(",") if v]:
            L.ssd_dev_knob(b"SSD_ABLATE", a)
            abl.append("direct abl%d %.1f" % (a, timed(lambda: ops.conv2d_fwd(x, w, b, 1, 1, 1, H, H, True, out=y))))
        L.ssd_dev_knob(b"SSD_ABLATE", 0)
        if abl:
            print("   ", " | ".join(abl), flush=True)
        tu = timed(lambda: ops.wino_weights(w, out=u))
        fl = 2.0 * B * H * H * Cout * 9 * Cin
        print("fwd B%d %dx%d %d->%d: direct %.1f us (%.0f TF/s) | wino %.1f us (%.0f eff TF/s, x%.2f) | weights %.1f us | rel diff %.2e" % (
            B, H, H, Cin, Cout, td, fl / td / 1e6, tw, fl / tw / 1e6, td / tw, tu, r), flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["check", "time"]
    if "check" in what:
        check()
    if "time" in what:
        time_()
