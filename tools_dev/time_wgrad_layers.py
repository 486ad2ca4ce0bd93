"""Weight gradients of the 19x19 / 38x38 GEMM-shaped layers (trunk nodes 12-16) timed alone, batch 64:
time_wgrad_layers.py [reps]   (run under rocprofv3 --kernel-trace --stats for the per-kernel split: tile kernel vs slab reduction)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ssd_object_detection_amd.ops as ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
LAYERS = [("conv12", 38, 512, 512, 1, 1), ("conv13", 38, 512, 1024, 3, 2), ("conv14", 19, 1024, 1024, 1, 1),
          ("conv15", 19, 1024, 256, 1, 1), ("conv16", 19, 256, 512, 3, 2)]
B = 64
for name, H, Cin, Cout, k, st in LAYERS:
    x = torch.randn((B, H, H, Cin), device="cuda").bfloat16()
    Ho, pt = ops.same_pad(H, k, st)
    dy = torch.randn((B, Ho, Ho, Cout), device="cuda").bfloat16()
    dw, db = ops.conv2d_bwd_weight(x, dy, Cout, k, st, pt, pt)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.conv2d_bwd_weight(x, dy, Cout, k, st, pt, pt, dw=dw, dbias=db)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    fl = 2.0 * B * Ho * Ho * Cout * k * k * Cin
    print(f"{name} wgrad {H}x{H} {Cin}->{Cout} k{k} s{st}: {us:.1f} us  {fl/us/1e6:.0f} TF/s  plan {ops.last_plan_name() if hasattr(ops,'last_plan_name') else ''}", flush=True)
