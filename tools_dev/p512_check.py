"""Dev helper: k_conv3x3_p512 (SSD_CONV_P512=1) against k_conv3x3_patch32 (=0): same accumulation order, so bit-identical."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ssd_object_detection_amd.ops as ops
from ssd_object_detection_amd import _lib
from tests.conv_cases import plan_name
L = _lib.lib()
torch.manual_seed(0)
K = b"SSD_CONV_P512"


def both(fn):
    out = []
    for v in (0, 1):
        L.ssd_dev_knob(K, v)
        out.append(fn())
    L.ssd_dev_knob(K, 0)
    return out


def eq(a, b):
    if a is None and b is None:
        return True
    return torch.equal(a, b)


for (B, H, W, Cin, Cout) in [(1, 16, 16, 64, 128), (2, 20, 37, 64, 128), (3, 19, 19, 64, 256), (2, 33, 18, 128, 128), (2, 75, 75, 128, 256),
                             (3, 38, 38, 256, 512), (2, 150, 150, 64, 128), (5, 17, 40, 64, 128)]:
    x = torch.relu(torch.randn((B, H, W, Cin), device="cuda")).bfloat16()
    w = (torch.randn((Cout, 3, 3, Cin), device="cuda") / np.sqrt(9 * Cin)).bfloat16()
    bias = torch.randn(Cout, device="cuda") * 0.1
    plans = both(lambda: plan_name(L, L.ssd_conv2d_fwd_plan(B, H, W, Cin, Cout, 3, 1, 1, 1, H, W, 0, 1 << 25)))
    y0, y1 = both(lambda: ops.conv2d_fwd(x, w, bias, 1, 1, 1, H, W, True))
    res = ["fwd " + str(eq(y0, y1))]
    for same in (True, False):
        a, b = both(lambda: ops.conv2d_fwd_pool(x, w, bias, 1, 1, 1, H, W, True, same))
        res.append("pool%d %s" % (same, all(eq(p, q) for p, q in zip(a, b))))
    try:
        a, b = both(lambda: ops.conv2d_fwd_pool(x, w, bias, 1, 1, 1, H, W, True, True, pool_only=True))
        res.append("poolonly %s" % all(eq(p, q) for p, q in zip(a, b)))
    except ValueError:
        res.append("poolonly n/a")
    dy = torch.randn((B, H, W, Cout), device="cuda").bfloat16()
    wt = ops.weight_transpose(w)
    mask = torch.randn((B, H, W, Cin), device="cuda").bfloat16()
    base = torch.randn((B, H, W, Cin), device="cuda").bfloat16()
    if Cin % 128 == 0:
        d0, d1 = both(lambda: ops.conv2d_bwd_data(dy, wt, None, (B, H, W, Cin), 1, 1, 1))
        res.append("dgrad %s" % eq(d0, d1))
        def acc():
            o = base.clone()
            ops.conv2d_bwd_data(dy, wt, mask, (B, H, W, Cin), 1, 1, 1, accumulate=True, out=o)
            return o
        d0, d1 = both(acc)
        res.append("dgrad+mask+acc %s" % eq(d0, d1))
    print((B, H, W, Cin, Cout), plans, " ".join(res), flush=True)
