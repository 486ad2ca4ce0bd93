"""One train step of the engine on a realistic batch (targets from the matcher, gradient rows from the loss), three times,
with event times -- the program the PMC / kernel-trace passes profile (tools_dev/collect_pmc.sh).  SSD_OVERLAP_HEADS=0 in
the environment gives the single-stream order (isolated kernel times)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ssd_object_detection_amd.ops as ops
from ssd_object_detection_amd.engine import SSDEngine
from ssd_object_detection_amd.data_loaders.synthetic import synth_batch_gt
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
eng = SSDEngine(classes=81, seed=0)
pset = ops.build_priors()
img = torch.rand((B, 300, 300, 3), device="cuda")
cls_l, box_l = synth_batch_gt(0, B)
gt = ops.pack_gt(box_l, cls_l)
tgt = ops.match_encode(*gt, pset, 0.5)
def ev(): return torch.cuda.Event(enable_timing=True)
for it in range(reps):
    e = [ev() for _ in range(5)]
    e[0].record()
    x = ops.image_prep(img)
    ploc, pconf = eng.forward(x)
    e[1].record()
    hgb = eng.head_grad_buffers(B)
    if hgb is not None:
        ops.ssd_loss_heads(pconf, ploc, *tgt, hgb)
        e[2].record()
        eng.backward(None, None, heads=hgb)
    else:
        _, dconf, dloc = ops.ssd_loss(pconf, ploc, *tgt)
        e[2].record()
        eng.backward(dloc, dconf)
    e[3].record()
    eng.clip_scales(0.01); eng.adam(1e-3, eng.grad, 1.0, True); e[4].record()
    torch.cuda.synchronize()
    print(f"B={B} fwd {e[0].elapsed_time(e[1]):.2f} ms  loss {e[1].elapsed_time(e[2]):.3f} ms  bwd {e[2].elapsed_time(e[3]):.2f} ms  opt {e[3].elapsed_time(e[4]):.2f} ms  total {e[0].elapsed_time(e[4]):.2f} ms -> {B/e[0].elapsed_time(e[4])*1e3:.0f} img/s", flush=True)
