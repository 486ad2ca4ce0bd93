"""dev: wgrad of nodes 12-16 under SSD_WGTILE_STAGES 2 / 4, with results compared bit for bit"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ssd_object_detection_amd.ops as ops
from ssd_object_detection_amd import _lib
LAYERS = [("conv12", 38, 512, 512, 1, 1), ("conv13", 38, 512, 1024, 3, 2), ("conv14", 19, 1024, 1024, 1, 1),
          ("conv15", 19, 1024, 256, 1, 1), ("conv16", 19, 256, 512, 3, 2)]
B, reps = 64, 20
for name, H, Cin, Cout, k, st in LAYERS:
    x = torch.randn((B, H, H, Cin), device="cuda").bfloat16()
    Ho, pt = ops.same_pad(H, k, st)
    dy = torch.randn((B, Ho, Ho, Cout), device="cuda").bfloat16()
    res = {}
    for stages in (2, 4):
        _lib.check(_lib.lib().ssd_dev_knob(b"SSD_WGTILE_STAGES", stages))
        dw, db = ops.conv2d_bwd_weight(x, dy, Cout, k, st, pt, pt)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.conv2d_bwd_weight(x, dy, Cout, k, st, pt, pt, dw=dw, dbias=db)
        e1.record(); torch.cuda.synchronize()
        res[stages] = (e0.elapsed_time(e1) / reps * 1e3, dw.clone(), db.clone())
    same = torch.equal(res[2][1], res[4][1]) and torch.equal(res[2][2], res[4][2])
    print("%s: 2 stages %.1f us, 4 stages %.1f us, bitwise equal %s" % (name, res[2][0], res[4][0], same), flush=True)
