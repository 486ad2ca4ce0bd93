"""Dev helper: the pre-pool layers' weight gradient, dense kernel on the un-pooled gradient vs the structured-sparse kernel on
the pooled gradient + codes (batch 64), plus the other 3x3 weight gradients of the step (dense kernel) for reference."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ssd_object_detection_amd.ops as ops
B = 64
def timed(fn, reps=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(3):
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(ts)[1]
tot_d = tot_s = 0.0
for name, H, Cin, Cout, same in (("conv1 block1_conv2", 300, 64, 64, False), ("conv4 block2_conv2", 150, 128, 128, False), ("conv8 block3_conv3", 75, 256, 256, True)):
    x = torch.randn((B, H, H, Cin), device="cuda").relu().bfloat16()
    y = torch.randn((B, H, H, Cout), device="cuda").relu().bfloat16()
    yp, code = ops.maxpool2x2_fwd_argmax(y, same=same)
    dp = torch.randn(yp.shape, device="cuda").bfloat16()
    dy = ops.maxpool2x2_bwd_argmax(code, dp, y.shape)
    del y
    dw, db = ops.conv2d_bwd_weight(x, dy, Cout, 3, 1, 1, 1)
    td = timed(lambda: ops.conv2d_bwd_weight(x, dy, Cout, 3, 1, 1, 1, dw=dw, dbias=db))
    dw2, db2 = ops.conv2d_bwd_weight_unpooled(x, dp, code)
    ts = timed(lambda: ops.conv2d_bwd_weight_unpooled(x, dp, code, dw=dw2, dbias=db2))
    err = float((dw - dw2).abs().max() / dw.abs().max())
    fl = 2.0 * B * H * H * Cout * 9 * Cin
    print("%-20s dense %.1f us (%.0f TF/s) | unpooled-sparse %.1f us (%.0f dense-equivalent TF/s)  rel err %.2e" % (name, td, fl / td / 1e6, ts, fl / ts / 1e6, err), flush=True)
    tot_d += td; tot_s += ts
    del x, dy, dp
print("sum: dense %.1f us, sparse %.1f us" % (tot_d, tot_s))
for name, H, Cin, Cout in (("conv3 block2_conv1", 150, 64, 128), ("conv6 block3_conv1", 75, 128, 256), ("conv7 block3_conv2", 75, 256, 256), ("conv10 38x38", 38, 256, 512), ("conv11 38x38", 38, 512, 512)):
    x = torch.randn((B, H, H, Cin), device="cuda").relu().bfloat16()
    dy = torch.randn((B, H, H, Cout), device="cuda").bfloat16()
    dw, db = ops.conv2d_bwd_weight(x, dy, Cout, 3, 1, 1, 1)
    td = timed(lambda: ops.conv2d_bwd_weight(x, dy, Cout, 3, 1, 1, 1, dw=dw, dbias=db))
    fl = 2.0 * B * H * H * Cout * 9 * Cin
    print("%-20s dense %.1f us (%.0f TF/s)" % (name, td, fl / td / 1e6), flush=True)
