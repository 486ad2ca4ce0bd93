"""Dev helper: score/decode and NMS times at batch 64 on the bench's NMS input (fp32 and bf16 logits)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ssd_object_detection_amd import _lib
if os.environ.get('AB_LIB'):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ['AB_LIB'])      # dev: a -DSSD_DEV_ABLATE build for the stage times
import ssd_object_detection_amd.ops as ops
B = 64
pset = ops.build_priors()
for dt in (torch.float32, torch.bfloat16):
    conf, loc = bench.nms_inputs(torch, B, pset.A, dt)
    sd = ops.score_decode(conf, loc, pset, 0.3)
    t_sd = bench.graph_timed(torch, lambda: ops.score_decode(conf, loc, pset, 0.3), 30)
    t_nms = bench.graph_timed(torch, lambda: ops.nms(sd[0], sd[1], sd[2], sd[3], 0.45, 400), 30)
    print(dt, "score_decode %.1f us  (%.2f us/img, %.0f GB/s)  nms %.1f us" % (t_sd * 1e6, t_sd / B * 1e6,
          conf.numel() * conf.element_size() / t_sd / 1e9, t_nms * 1e6), flush=True)
    L = _lib.lib()
    for a in (1, 2, 4):
        L.ssd_dev_knob(b"SSD_ABLATE", a)
        print("   nms stop after stage %d: %.1f us" % (a, bench.graph_timed(torch, lambda: ops.nms(sd[0], sd[1], sd[2], sd[3], 0.45, 400), 30) * 1e6), flush=True)
    L.ssd_dev_knob(b"SSD_ABLATE", 0)
    dst = torch.empty_like(conf)
    t_cp = bench.graph_timed(torch, lambda: dst.copy_(conf), 30)
    t_sum = bench.graph_timed(torch, lambda: conf.amax(dim=2), 30)
    print("   copy %.1f us (%.0f GB/s rd+wr)   amax(dim=2) %.1f us (%.0f GB/s)" % (t_cp * 1e6, 2 * conf.numel() * conf.element_size() / t_cp / 1e9,
          t_sum * 1e6, conf.numel() * conf.element_size() / t_sum / 1e9), flush=True)
