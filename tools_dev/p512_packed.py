"""Dev experiment: k_conv3x3_p512 with weight slices pre-packed contiguously (SSD_ABLATE bit 256) vs the [N][3][3][C] layout."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ssd_object_detection_amd.ops as ops
from ssd_object_detection_amd import _lib
L = _lib.lib()
torch.manual_seed(0)
def timed(run, reps=10):
    run(); torch.cuda.synchronize(); ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): run()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(ts)[2]
for (B, H, Cin, Cout) in [(64, 75, 256, 256), (64, 38, 512, 512), (64, 150, 128, 128), (64, 75, 128, 256)]:
    x = torch.relu(torch.randn((B, H, H, Cin), device="cuda")).bfloat16()
    w = (torch.randn((Cout, 3, 3, Cin), device="cuda") / np.sqrt(9 * Cin)).bfloat16()
    wp = w.view(Cout // 128, 128, 9, Cin // 32, 32).permute(0, 3, 2, 1, 4).contiguous()
    b = torch.zeros(Cout, device="cuda")
    L.ssd_dev_knob(b"SSD_CONV_P512", 0)
    y0 = ops.conv2d_fwd(x, w, b, 1, 1, 1, H, H, True)
    t0 = timed(lambda: ops.conv2d_fwd(x, w, b, 1, 1, 1, H, H, True, out=y0))
    L.ssd_dev_knob(b"SSD_CONV_P512", 1)
    y1 = ops.conv2d_fwd(x, w, b, 1, 1, 1, H, H, True)
    t1 = timed(lambda: ops.conv2d_fwd(x, w, b, 1, 1, 1, H, H, True, out=y1))
    L.ssd_dev_knob(b"SSD_ABLATE", 256)
    y2 = ops.conv2d_fwd(x, wp.view(Cout, 3, 3, Cin), b, 1, 1, 1, H, H, True)
    t2 = timed(lambda: ops.conv2d_fwd(x, wp.view(Cout, 3, 3, Cin), b, 1, 1, 1, H, H, True, out=y2))
    L.ssd_dev_knob(b"SSD_ABLATE", 0); L.ssd_dev_knob(b"SSD_CONV_P512", 0)
    print("B%d %dx%d %d->%d: patch32 %.1f us | p512 %.1f | p512 packed weights %.1f | equal %s %s" % (B, H, H, Cin, Cout, t0, t1, t2,
          torch.equal(y0, y1), torch.equal(y0, y2)), flush=True)
