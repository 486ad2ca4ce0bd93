import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ssd_object_detection_amd.ops as ops
from ssd_object_detection_amd.engine import SSDEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
eng = SSDEngine(classes=81, seed=0)
torch.manual_seed(0)
img = torch.rand((B, 300, 300, 3), device="cuda")
x = ops.image_prep(img)
dloc = (torch.randn((B, 8732, 4), device="cuda") * 1e-3).bfloat16()
dconf = (torch.randn((B, 8732, 81), device="cuda") * 1e-3).bfloat16()
def run(overlap):
    eng.overlap_heads = overlap
    eng.grad.zero_()
    loc, conf = eng.forward(x)
    eng.backward(dloc, dconf)
    torch.cuda.synchronize()
    ga = eng._acts(B)["gacts"]
    return eng.grad.clone(), loc.clone(), conf.clone(), [g.clone() if g is not None else None for g in ga]
ref = run(False)
names = [t.name for t in eng.tensors]
for it in range(6):
    got = run(True)
    bad = []
    if not torch.equal(got[1], ref[1]) or not torch.equal(got[2], ref[2]): bad.append("fwd outputs")
    for t in eng.tensors:
        a = ref[0][t.offset:t.offset + t.numel]; b = got[0][t.offset:t.offset + t.numel]
        if not torch.equal(a, b): bad.append("%s (max diff %.3g of %.3g)" % (t.name, (a - b).abs().max().item(), a.abs().max().item()))
    for gi, (ga, gb) in enumerate(zip(ref[3], got[3])):
        if ga is not None and not torch.equal(ga, gb):
            bad.append("gacts[%d] %s diff elems %d of %d, max %.3g" % (gi, tuple(ga.shape), (ga != gb).sum().item(), ga.numel(), (ga.float() - gb.float()).abs().max().item()))
    print("iter", it, "MISMATCH: " + "; ".join(bad[:8]) if bad else "identical", flush=True)
ref2 = run(False)
print("overlap off twice identical:", torch.equal(ref[0], ref2[0]))
