"""Dev: the rows form of the loss (ssd_loss_fwd_bwd_heads) at batch 64 on network-like bf16 logits, graph-replayed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ssd_object_detection_amd.ops as ops
from ssd_object_detection_amd.data_loaders.synthetic import synth_batch_gt
B = 64
pset = ops.build_priors()
cls_l, box_l = synth_batch_gt(0, B)
tgt = ops.match_encode(*ops.pack_gt(box_l, cls_l), pset, 0.5)
g = torch.Generator(device="cuda").manual_seed(1)
conf = (0.05 * torch.randn((B, 8732, 81), generator=g, device="cuda")).bfloat16()      # near-uniform softmax, as a random-init network
loc = (0.05 * torch.randn((B, 8732, 4), generator=g, device="cuda")).bfloat16()
hw, npc = (1444, 361, 100, 25, 9, 1), (4, 6, 6, 6, 4, 4)
hgb = ops.HeadGradBuffers(B, hw, npc, tuple((n * 85 + 7) // 8 * 8 for n in npc))
def run():
    ops.ssd_loss_heads(conf, loc, *tgt, hgb)
run(); run(); torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
with torch.cuda.stream(s):
    run(); torch.cuda.synchronize()
    with torch.cuda.graph(gr, stream=s):
        run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
gr.replay(); torch.cuda.synchronize()
e0.record()
for _ in range(50): gr.replay()
e1.record(); torch.cuda.synchronize()
print("ssd_loss_fwd_bwd_heads B=64: %.1f us per call (graph replay)" % (e0.elapsed_time(e1) / 50 * 1e3))
