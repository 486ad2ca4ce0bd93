"""Dev helper: A/B timing of convolution variants for many layers inside ONE process.
usage: ab_conv_multi.py "mode B H Cin Cout k stride knob v0,v1,..." ...      mode: fwd | dgrad | wgrad | head"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ssd_object_detection_amd.ops as ops
from ssd_object_detection_amd import _lib
if os.environ.get('AB_LIB'):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ['AB_LIB'])      # dev: time another build of the library
from tests.conv_cases import plan_name
L = _lib.lib()
for kv in os.environ.get('AB_PRESET', '').split():          # other development knobs held fixed: "NAME=value ..."
    k_, v_ = kv.split('=')
    L.ssd_dev_knob(k_.encode(), int(v_))
torch.manual_seed(0)
for spec in sys.argv[1:]:
    mode, B, H, Cin, Cout, k, S, knob, vals = spec.split()
    B, H, Cin, Cout, k, S = map(int, (B, H, Cin, Cout, k, S))
    vals = [int(v) for v in vals.split(",")]
    x = torch.relu(torch.randn((B, H, H, Cin), device="cuda")).bfloat16()
    Ho, pt = ops.same_pad(H, k, S)
    cp = (Cout + 7) // 8 * 8
    if mode == "fwd":
        w = (torch.randn((Cout, k, k, Cin), device="cuda") * 0.05).bfloat16(); b = torch.zeros(Cout, device="cuda")
        y = ops.conv2d_fwd(x, w, b, S, pt, pt, Ho, Ho, True)
        run = lambda: ops.conv2d_fwd(x, w, b, S, pt, pt, Ho, Ho, True, out=y)
        plan = lambda: L.ssd_conv2d_fwd_plan(B, H, H, Cin, Cout, k, S, pt, pt, Ho, Ho, 0, 1 << 25)
    elif mode == "head":
        per_cell = Cout // 85
        w = (torch.randn((Cout, 3, 3, Cin), device="cuda") * 0.05).bfloat16(); b = torch.zeros(Cout, device="cuda")
        loc = torch.empty((B, H * H * per_cell, 4), device="cuda", dtype=torch.bfloat16)
        conf = torch.empty((B, H * H * per_cell, 81), device="cuda", dtype=torch.bfloat16)
        run = lambda: ops.conv2d_head_fwd(x, w, b, loc, conf, per_cell, 81, 0)
        plan = lambda: L.ssd_conv2d_head_fwd_plan(B, H, H, Cin, per_cell, 81, 1 << 25)
    elif mode == "wgrad":
        dy = torch.zeros((B, Ho, Ho, cp), device="cuda").bfloat16(); dy[..., :Cout] = torch.randn((B, Ho, Ho, Cout), device="cuda").bfloat16()
        dw, db = ops.conv2d_bwd_weight(x, dy, Cout, k, S, pt, pt)
        run = lambda: ops.conv2d_bwd_weight(x, dy, Cout, k, S, pt, pt, dw=dw, dbias=db)
        plan = lambda: L.ssd_conv2d_bwd_weight_plan(B, H, H, Cin, Cout, cp, k, S, pt, pt, Ho, Ho)
    elif mode == "w0fused":                         # second layer's data gradient + first layer's weight gradient, one kernel
        dy = torch.randn((B, H, H, 64), device="cuda").bfloat16()
        w_t = (torch.randn((64, 3, 3, 64), device="cuda") * 0.05).bfloat16()
        img = torch.zeros((B, H, H, 8), device="cuda").bfloat16(); img[..., :3] = torch.randn((B, H, H, 3), device="cuda").bfloat16()
        bits = torch.randint(0, 256, (B, H, H, 8), device="cuda", dtype=torch.uint8)
        dw, db = ops.conv2d_bwd_data_wgrad_first(dy, w_t, bits, img)
        run = lambda: ops.conv2d_bwd_data_wgrad_first(dy, w_t, bits, img, dw=dw, dbias=db)
        plan = lambda: -1
    elif mode == "dgradbits":                       # the data gradient as the engine calls it: ReLU mask as sign bits
        dy = torch.zeros((B, Ho, Ho, cp), device="cuda").bfloat16(); dy[..., :Cout] = torch.randn((B, Ho, Ho, Cout), device="cuda").bfloat16()
        w_t = (torch.randn((Cin, k, k, cp), device="cuda") * 0.05).bfloat16()
        dx = torch.empty((B, H, H, Cin), device="cuda", dtype=torch.bfloat16)
        bits = torch.randint(0, 256, (B, H, H, Cin // 8), device="cuda", dtype=torch.uint8)
        run = lambda: ops.conv2d_bwd_data_bits(dy, w_t, bits, (B, H, H, Cin), S, pt, pt, accumulate=False, out=dx)
        plan = lambda: L.ssd_conv2d_bwd_data_plan(B, H, H, Cin, cp, k, S, pt, pt, Ho, Ho, 0, 1 << 25)
    else:
        dy = torch.zeros((B, Ho, Ho, cp), device="cuda").bfloat16(); dy[..., :Cout] = torch.randn((B, Ho, Ho, Cout), device="cuda").bfloat16()
        w_t = (torch.randn((Cin, k, k, cp), device="cuda") * 0.05).bfloat16()
        dx = torch.empty((B, H, H, Cin), device="cuda", dtype=torch.bfloat16)
        run = lambda: ops.conv2d_bwd_data(dy, w_t, x, (B, H, H, Cin), S, pt, pt, accumulate=False, out=dx)
        plan = lambda: L.ssd_conv2d_bwd_data_plan(B, H, H, Cin, cp, k, S, pt, pt, Ho, Ho, 0, 1 << 25)

    def timed(v, reps=10):
        L.ssd_dev_knob(knob.encode(), v)
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    t = {v: [] for v in vals}
    for rnd in range(5):
        for v in vals:
            t[v].append(timed(v))
    fl = 2.0 * B * Ho * Ho * Cout * k * k * Cin
    out = []
    for v in vals:
        L.ssd_dev_knob(knob.encode(), v)
        a = sorted(t[v])
        out.append("%s=%d [%s] %.1f us %.0f TF/s" % (knob, v, (plan_name(L, plan()) if plan() >= 0 else "fused"), a[len(a) // 2], fl / a[len(a) // 2] / 1e6))
    L.ssd_dev_knob(knob.encode(), vals[0])
    print("%-5s B%d %dx%d %d->%d k%d s%d: " % (mode, B, H, H, Cin, Cout, k, S) + " | ".join(out), flush=True)
