"""Dev: one-hot channel sweep through ssd_conv3x3_fwd_mxfp8 -- which weight channel / which scale meets input channel k (this is how the
second-half k mapping of v_mfma_scale_f32_16x16x128_f8f6f4 was found: channels 16..63 came out scaled by another block's scale)."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import torch.nn.functional as F
import ssd_object_detection_amd.ops as ops
B,H,W,C,N = 1,12,12,256,128
def run(x, w):
    xq, xs = ops.quantize_mx_fp8(x); wq, ws = ops.quantize_mx_fp8(w)
    return ops.conv3x3_fwd_mxfp8(xq, xs, wq, ws, None, relu=False).float()
# w[n, centre, c] = (c % 16) + 1 for all n; x one-hot at channel k (all pixels) -> y = (k % 16) + 1 if the k order agrees
w = torch.zeros((N,3,3,C), device="cuda"); w[:,1,1,:] = (torch.arange(C, device="cuda") % 16 + 1).float()
res = []
for k in range(0, 64):
    x = torch.zeros((B,H,W,C), device="cuda"); x[..., k] = 1
    y = run(x.bfloat16(), w.bfloat16())
    res.append(int(y[0,5,5,0].item()))
print("x one-hot channel k -> w label seen (expect k%16+1):", res)
# and which channel block: w label = c // 16 + 1 (1..16)
w = torch.zeros((N,3,3,C), device="cuda"); w[:,1,1,:] = (torch.arange(C, device="cuda") // 16 + 1).float()
res = []
for k in range(0, 256, 8):
    x = torch.zeros((B,H,W,C), device="cuda"); x[..., k] = 1
    y = run(x.bfloat16(), w.bfloat16())
    res.append(int(y[0,5,5,0].item()))
print("x one-hot channel k (step 8) -> w 16-block label seen (expect k//16+1):", res)
