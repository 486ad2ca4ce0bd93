"""Dev: k_conv3x3_p512 with weight / patch requests fetching nothing (SSD_ABLATE 16 / 32 / 48), same instruction stream."""
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
LAYERS = [(64, 75, 256, 256), (64, 38, 512, 512), (64, 150, 128, 128)][:int(os.environ.get('P512_LAYERS', '3'))]
for (B, H, Cin, Cout) in LAYERS:
    x = torch.relu(torch.randn((B, H, H, Cin), device="cuda")).bfloat16()
    w = (torch.randn((Cout, 3, 3, Cin), device="cuda") / np.sqrt(9 * Cin)).bfloat16()
    b = torch.zeros(Cout, device="cuda")
    y = ops.conv2d_fwd(x, w, b, 1, 1, 1, H, H, True)
    out = []
    for kern in (0, 1):
        L.ssd_dev_knob(b"SSD_CONV_P512", kern)
        for a in (0, 16, 32, 48):
            L.ssd_dev_knob(b"SSD_ABLATE", a)
            out.append("%s abl%d %.1f" % ("p512" if kern else "p32", a, timed(lambda: ops.conv2d_fwd(x, w, b, 1, 1, 1, H, H, True, out=y))))
    L.ssd_dev_knob(b"SSD_ABLATE", 0); L.ssd_dev_knob(b"SSD_CONV_P512", 0)
    print("B%d %dx%d %d->%d: " % (B, H, H, Cin, Cout) + " | ".join(out), flush=True)
