"""Dev helper: ssd_match_encode at a large batch against the oracle, for each path (SSD_MATCH_FUSED 0 / 1 / 3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ssd_object_detection_amd.ops as ops
from ssd_object_detection_amd import _lib
from ssd_object_detection_amd.data_loaders.synthetic import synth_batch_gt
from oracle import ssd_oracle as O
L = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pset = ops.build_priors()
pri = pset.priors.cpu().numpy()
cls_l, box_l = synth_batch_gt(0, B)
gt = ops.pack_gt(box_l, cls_l)
ref = [O.match_closed_form(c, b, pri, 0.5) for c, b in zip(cls_l, box_l)]
for v in (0, 1, 3):
    L.ssd_dev_knob(b"SSD_MATCH_FUSED", v)
    bad = {}
    for rep in range(5):
        owner = torch.full((B, 8732), -7, dtype=torch.int32, device="cuda")
        cls, loc, mask = ops.match_encode(*gt, pset, 0.5, owner=owner)
        cls, mask, owner = cls.cpu().numpy(), mask.cpu().numpy().astype(bool), owner.cpu().numpy()
        for i in range(B):
            rc, rb, rm = ref[i]
            if not (np.array_equal(mask[i], rm) and np.array_equal(cls[i], rc)):
                d = np.nonzero(mask[i] != rm)[0]
                bad.setdefault(i, []).append((rep, len(cls_l[i]), d[:6].tolist(), owner[i][d[:6]].tolist()))
    print("variant", v, "bad images:", len(bad), list(bad.items())[:4], flush=True)
L.ssd_dev_knob(b"SSD_MATCH_FUSED", 0)
