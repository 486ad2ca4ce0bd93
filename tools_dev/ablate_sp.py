"""Dev: timing-only ablations of k_conv3x3_wgrad_unpool (SSD_SP_ABLATE bits: 1 no DMA after the first block, 2 no smfmac, 4 no B reads, 8 no producer)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ssd_object_detection_amd.ops as ops
from ssd_object_detection_amd import _lib
L = _lib.lib()
B = 64
def timed(fn, reps=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for name, H, Cin, Cout, same in (("conv4", 150, 128, 128, False), ("conv8", 75, 256, 256, True)):
    x = torch.randn((B, H, H, Cin), device="cuda").relu().bfloat16()
    y = torch.randn((B, H, H, Cout), device="cuda").relu().bfloat16()
    yp, code = ops.maxpool2x2_fwd_argmax(y, same=same)
    dp = torch.randn(yp.shape, device="cuda").bfloat16()
    dw, db = ops.conv2d_bwd_weight_unpooled(x, dp, code)
    out = []
    for abl in (0, 1, 2, 4, 8, 3, 6, 7, 15):
        L.ssd_dev_knob(b"SSD_SP_ABLATE", abl)
        out.append("%d:%.0f" % (abl, timed(lambda: ops.conv2d_bwd_weight_unpooled(x, dp, code, dw=dw, dbias=db))))
    L.ssd_dev_knob(b"SSD_SP_ABLATE", 0)
    print(name, " ".join(out), flush=True)
