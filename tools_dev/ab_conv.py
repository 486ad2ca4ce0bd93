"""A/B timing of conv variants inside ONE process: ab_conv.py mode B H Cin Cout k knob v0 v1 [rounds]
mode: fwd | dgrad | head(per_cell classes derived: Cout = per_cell*85)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ssd_object_detection_amd.ops as ops
from ssd_object_detection_amd import _lib
mode = sys.argv[1]
B, H, Cin, Cout, k = map(int, sys.argv[2:7])
knob, v0, v1 = sys.argv[7], int(sys.argv[8]), int(sys.argv[9])
rounds = int(sys.argv[10]) if len(sys.argv) > 10 else 7
L = _lib.lib()
torch.manual_seed(0)
x = torch.relu(torch.randn((B, H, H, Cin), device="cuda")).bfloat16()
S = int(os.environ.get("AB_STRIDE", "1"))
Ho, pt = ops.same_pad(H, k, S)
if mode == "fwd":
    w = (torch.randn((Cout, k, k, Cin), device="cuda") * 0.05).bfloat16(); b = torch.zeros(Cout, device="cuda")
    y = ops.conv2d_fwd(x, w, b, 1, pt, pt, Ho, Ho, True)
    run = lambda: ops.conv2d_fwd(x, w, b, 1, pt, pt, Ho, Ho, True, out=y)
elif mode == "fwdpool":                                  # conv + fused 2x2 pooling, pooled map only
    w = (torch.randn((Cout, k, k, Cin), device="cuda") * 0.05).bfloat16(); b = torch.zeros(Cout, device="cuda")
    _, yp, code = ops.conv2d_fwd_pool(x, w, b, 1, pt, pt, Ho, Ho, True, Ho % 2 == 1, pool_only=True)
    run = lambda: ops.conv2d_fwd_pool(x, w, b, 1, pt, pt, Ho, Ho, True, Ho % 2 == 1, pool_out=yp, code=code, pool_only=True)
elif mode == "head":
    per_cell = Cout // 85
    w = (torch.randn((Cout, 3, 3, Cin), device="cuda") * 0.05).bfloat16(); b = torch.zeros(Cout, device="cuda")
    A = H * H * per_cell
    loc = torch.empty((B, A, 4), device="cuda", dtype=torch.bfloat16); conf = torch.empty((B, A, 81), device="cuda", dtype=torch.bfloat16)
    run = lambda: ops.conv2d_head_fwd(x, w, b, loc, conf, per_cell, 81, 0)
elif mode == "wgrad":
    dy = torch.randn((B, Ho, Ho, Cout), device="cuda").bfloat16()
    dw, db = ops.conv2d_bwd_weight(x, dy, Cout, k, S, pt, pt)
    run = lambda: ops.conv2d_bwd_weight(x, dy, Cout, k, S, pt, pt, dw=dw, dbias=db)
else:
    # data gradient of a Cin->Cout conv: dy [B,H,H,Cout], w_t [Cin][k][k][Cout]
    dy = torch.randn((B, H, H, Cout), device="cuda").bfloat16()
    w_t = (torch.randn((Cin, k, k, Cout), device="cuda") * 0.05).bfloat16()
    dx = torch.empty((B, H, H, Cin), device="cuda", dtype=torch.bfloat16)
    mask = None if os.environ.get('AB_NOMASK') else x
    run = lambda: ops.conv2d_bwd_data(dy, w_t, mask, (B, H, H, Cin), 1, pt, pt, accumulate=False, out=dx)
def timed(v, reps=10):
    L.ssd_dev_knob(knob.encode(), v)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
t = {v0: [], v1: []}
for r in range(rounds):
    for v in (v0, v1): t[v].append(timed(v))
fl = 2.0 * B * Ho * Ho * Cout * k * k * Cin
for v in (v0, v1):
    a = sorted(t[v]); med = a[len(a) // 2]
    print(f"{mode} {H}x{H} {Cin}->{Cout} k{k} {knob}={v}: med {med:.1f} us min {a[0]:.1f}  {fl/med/1e6:.0f} TF/s", flush=True)
