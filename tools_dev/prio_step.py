"""Dev helper: train-step time with the main (data-gradient) chain on a high-priority stream vs the default stream.
usage: prio_step.py [steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ssd_object_detection_amd.ops as ops
from ssd_object_detection_amd import optimizers
from ssd_object_detection_amd.models import SSDObjectDetectionModel
from ssd_object_detection_amd.data_loaders.synthetic import synth_batch_gt
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B = 64
model = SSDObjectDetectionModel(classes=80, log_dir="gpurun_out/prio", seed=0, timestamp_dir=False)
opt = optimizers.Adam(1e-3)
gen = torch.Generator(device="cuda").manual_seed(1234)
img = torch.rand((B, 300, 300, 3), generator=gen, device="cuda")
cls_l, box_l = synth_batch_gt(0, B)
gt = ops.pack_gt(box_l, cls_l)
out = None
def step():
    global out
    out = model.match_async(gt, out=out)
    x = ops.image_prep(img, normalize=True)
    model._train_step(x, *out, opt)
def run(stream):
    with torch.cuda.stream(stream):
        for _ in range(5): step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps): step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print("priority range", lo, hi)
s_def = torch.cuda.current_stream()
s_hi = torch.cuda.Stream(priority=-1)
for rnd in range(3):
    print("round %d: default-priority main %.3f ms | high-priority main %.3f ms" % (rnd, run(s_def), run(s_hi)), flush=True)
