"""Dev helper: run ssd_match_encode a few times for one configuration and one SSD_MATCH_FUSED value (for rocprofv3 --pmc).
usage: pmc_match.py B:nt variant"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ssd_object_detection_amd.ops as ops
from ssd_object_detection_amd import _lib
from ssd_object_detection_amd.data_loaders.synthetic import synth_batch_gt
B, nt = sys.argv[1].split(":")
B = int(B); nt = None if nt == "mix" else int(nt)
_lib.lib().ssd_dev_knob(b"SSD_MATCH_FUSED", int(sys.argv[2]))
pset = ops.build_priors()
cls_l, box_l = synth_batch_gt(0, B, nt)
gt = ops.pack_gt(box_l, cls_l)
out = None
for _ in range(6):
    out = ops.match_encode(*gt, pset, 0.5, out=out)
torch.cuda.synchronize()
