"""Dev helper: ssd_match_encode by path (SSD_MATCH_FUSED 0 = three launches, 1 = one launch, 2 = one launch with two columns per
thread), graph-replayed, interleaved rounds in one process.  usage: ab_match.py [B:nt ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ssd_object_detection_amd.ops as ops
from ssd_object_detection_amd import _lib
from ssd_object_detection_amd.data_loaders.synthetic import synth_batch_gt

L = _lib.lib()
cfgs = sys.argv[1:] or ["64:mix", "64:1", "64:8", "64:32", "64:93", "32:mix", "256:mix"]
pset = ops.build_priors()
variants = [int(v) for v in os.environ.get("VARIANTS", "0,1").split(",")]
for cfg in cfgs:
    B, nt = cfg.split(":")
    B = int(B); nt = None if nt == "mix" else int(nt)
    cls_l, box_l = synth_batch_gt(0, B, nt)
    gt = ops.pack_gt(box_l, cls_l)
    graphs, outs = {}, {}
    for v in variants:
        L.ssd_dev_knob(b"SSD_MATCH_FUSED", v)
        out = ops.match_encode(*gt, pset, 0.5)
        outs[v] = [t.clone() for t in out]
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            ops.match_encode(*gt, pset, 0.5, out=out)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                ops.match_encode(*gt, pset, 0.5, out=out)
        graphs[v] = g
    for v in [v for v in variants[1:] if v < 4]:
        for name, a, b in zip(("cls", "loc", "mask"), outs[variants[0]], outs[v]):
            if not torch.equal(a, b):
                d = (a != b)
                if d.dim() == 3:
                    d = d.any(-1)
                idx = d.nonzero()[:5].tolist()
                print("VARIANT", v, "differs in", name, "at", int(d.sum()), "anchors, first", idx,
                      [(a[i, j].tolist(), b[i, j].tolist()) for i, j in idx[:3]], flush=True)
    times = {v: [] for v in variants}
    for rnd in range(5):
        for v in variants:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            graphs[v].replay(); torch.cuda.synchronize()
            e0.record()
            for _ in range(50):
                graphs[v].replay()
            e1.record(); torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) * 1e3 / 50)
    byts = B * (8732 * 53) + 20 * gt[3]
    print(f"B={B} nt={nt} total_gt={gt[3]}:", "  ".join("v%d %.1f us (min %.1f, %.2f TB/s)" % (v, float(np.median(times[v])), min(times[v]), byts / min(times[v]) / 1e6) for v in variants), flush=True)
L.ssd_dev_knob(b"SSD_MATCH_FUSED", 0)
