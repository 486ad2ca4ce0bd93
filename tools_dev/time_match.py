"""Dev helper: time ssd_match_encode on synthetic batches (HIP events on the launch stream).
usage: time_match.py [B:nt ...]   nt = 'mix' or an int"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ssd_object_detection_amd.ops as ops
from ssd_object_detection_amd.data_loaders.synthetic import synth_batch_gt

cfgs = sys.argv[1:] or ["64:mix", "64:1", "64:8", "64:32", "64:93", "256:mix", "1024:mix"]
pset = ops.build_priors()
for cfg in cfgs:
    B, nt = cfg.split(":")
    B = int(B); nt = None if nt == "mix" else int(nt)
    cls_l, box_l = synth_batch_gt(0, B, nt)
    gt_box, gt_cls, gt_off, total, max_nt = ops.pack_gt(box_l, cls_l)
    out = None
    for _ in range(5):
        out = ops.match_encode(gt_box, gt_cls, gt_off, total, max_nt, pset, 0.5, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        ops.match_encode(gt_box, gt_cls, gt_off, total, max_nt, pset, 0.5, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    byts = B * (8732 * 53) + 20 * total
    print(f"B={B} nt={nt} total_gt={total} max_nt={max_nt} pos={int(out[2].sum())}: {us:.1f} us/call  {us/B:.3f} us/img  {byts/us/1e6:.3f} TB/s algorithmic", flush=True)
