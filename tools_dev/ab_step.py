"""Dev helper: A/B of a development knob on the whole conv fwd+bwd+optimizer step inside one process (box-to-box spread is
larger than most kernel-level gains).  usage: ab_step.py KNOB v0,v1[,..] [batch]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ssd_object_detection_amd.ops as ops
from ssd_object_detection_amd import _lib
from ssd_object_detection_amd.engine import SSDEngine
L = _lib.lib()
knob, vals = sys.argv[1].encode(), [int(v) for v in sys.argv[2].split(",")]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
eng = SSDEngine(classes=81, seed=0)
x = ops.image_prep(torch.rand((B, 300, 300, 3), device="cuda"))
dloc = (torch.randn((B, 8732, 4), device="cuda") * 1e-3).bfloat16()
dconf = (torch.randn((B, 8732, 81), device="cuda") * 1e-3).bfloat16()
def step():
    eng.forward(x); eng.backward(dloc, dconf); eng.clip_scales(0.01); eng.adam(1e-3, eng.grad, 1.0, True)
def timed(n=10):
    step(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): step()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
res = {v: [] for v in vals}
for rnd in range(6):
    for v in vals:
        L.ssd_dev_knob(knob, v)
        res[v].append(timed())
for v in vals:
    a = sorted(res[v])
    print("%s=%d: median %.3f ms  (min %.3f, max %.3f)" % (knob.decode(), v, a[len(a) // 2], a[0], a[-1]), flush=True)
