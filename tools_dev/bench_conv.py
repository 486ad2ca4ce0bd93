"""time one conv layer fwd: bench_conv.py B H Cin Cout k [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ssd_object_detection_amd.ops as ops
B, H, Cin, Cout, k = map(int, sys.argv[1:6])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 20
x = torch.randn((B, H, H, Cin), device="cuda").bfloat16()
w = (torch.randn((Cout, k, k, Cin), device="cuda") * 0.05).bfloat16()
b = torch.zeros(Cout, device="cuda")
Ho, pt = ops.same_pad(H, k, 1)
y = ops.conv2d_fwd(x, w, b, 1, pt, pt, Ho, Ho, True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): ops.conv2d_fwd(x, w, b, 1, pt, pt, Ho, Ho, True, out=y)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
fl = 2.0 * B * Ho * Ho * Cout * k * k * Cin
print(f"ABL={os.environ.get('SSD_ABLATE','0')} PATCH={os.environ.get('SSD_CONV_PATCH','-')} conv {H}x{H} {Cin}->{Cout} k{k}: {us:.1f} us  {fl/us/1e6:.0f} TF/s", flush=True)
