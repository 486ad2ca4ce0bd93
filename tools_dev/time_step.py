import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ssd_object_detection_amd.ops as ops
from ssd_object_detection_amd.engine import SSDEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
eng = SSDEngine(classes=81, seed=0)
img = torch.rand((B, 300, 300, 3), device="cuda")
x = ops.image_prep(img)
dloc = (torch.randn((B, 8732, 4), device="cuda") * 1e-3).bfloat16()
dconf = (torch.randn((B, 8732, 81), device="cuda") * 1e-3).bfloat16()
def ev(): return torch.cuda.Event(enable_timing=True)
for it in range(3):
    e = [ev() for _ in range(4)]
    e[0].record(); eng.forward(x); e[1].record(); eng.backward(dloc, dconf); e[2].record()
    eng.clip_scales(0.01); eng.adam(1e-3, eng.grad, 1.0, True); e[3].record()
    torch.cuda.synchronize()
    print(f"B={B} fwd {e[0].elapsed_time(e[1]):.2f} ms  bwd {e[1].elapsed_time(e[2]):.2f} ms  opt {e[2].elapsed_time(e[3]):.2f} ms  total {e[0].elapsed_time(e[3]):.2f} ms -> {B/e[0].elapsed_time(e[3])*1e3:.0f} img/s", flush=True)
