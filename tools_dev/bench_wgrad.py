"""time one conv layer wgrad: bench_wgrad.py B H Cin Cout k [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ssd_object_detection_amd.ops as ops
B, H, Cin, Cout, k = map(int, sys.argv[1:6])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 10
x = torch.randn((B, H, H, Cin), device="cuda").bfloat16()
Ho, pt = ops.same_pad(H, k, 1)
dy = torch.randn((B, Ho, Ho, Cout), device="cuda").bfloat16()
dw, db = ops.conv2d_bwd_weight(x, dy, Cout, k, 1, pt, pt)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): ops.conv2d_bwd_weight(x, dy, Cout, k, 1, pt, pt, dw=dw, dbias=db)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
fl = 2.0 * B * Ho * Ho * Cout * k * k * Cin
print(f"ABL={os.environ.get('SSD_ABLATE','0')} VAR={os.environ.get('SSD_CONV_VARIANT','1')} wgrad {H}x{H} {Cin}->{Cout} k{k}: {us:.1f} us  {fl/us/1e6:.0f} TF/s", flush=True)
