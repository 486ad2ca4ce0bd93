import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ssd_object_detection_amd.ops as ops
from ssd_object_detection_amd.data_loaders.synthetic import synth_batch_gt
pset = ops.build_priors()
for B, dt in [(32, torch.bfloat16), (32, torch.float32), (64, torch.bfloat16)]:
    cls_l, box_l = synth_batch_gt(0, B)
    gt = ops.pack_gt(box_l, cls_l)
    cls, gloc, mask = ops.match_encode(*gt, pset, 0.5)
    conf = torch.randn((B, 8732, 81), device="cuda").to(dt); loc = torch.randn((B, 8732, 4), device="cuda").to(dt)
    for _ in range(3): ops.ssd_loss(conf, loc, cls, gloc, mask)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.ssd_loss(conf, loc, cls, gloc, mask)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    es = conf.element_size()
    byts = B * 8732 * (2 * 81 * es + 2 * 4 * es + 4 * 4 + 5)
    print(f"loss B={B} {dt}: {us:.1f} us/call, algorithmic {byts/1e6:.1f} MB -> {byts/us/1e6:.2f} TB/s", flush=True)
