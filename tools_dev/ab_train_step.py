"""Dev helper: A/B of development knobs on the REAL train step (match_async + prep + _train_step: sparse heads, fused optimizer,
three streams) inside ONE process, settings interleaved round by round.
usage: ab_train_step.py "NAME=v[,NAME=v...]" "NAME=v..." ... [--rounds N] [--steps N]      ("" = defaults)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ssd_object_detection_amd.ops as ops
from ssd_object_detection_amd import _lib, optimizers
from ssd_object_detection_amd.models import SSDObjectDetectionModel
from ssd_object_detection_amd.data_loaders.synthetic import synth_batch_gt
args = sys.argv[1:]
rounds, steps = 5, 20
while "--rounds" in args:
    i = args.index("--rounds"); rounds = int(args[i + 1]); del args[i:i + 2]
while "--steps" in args:
    i = args.index("--steps"); steps = int(args[i + 1]); del args[i:i + 2]
L = _lib.lib()
B = 64
model = SSDObjectDetectionModel(classes=80, log_dir="gpurun_out/ab", seed=0, timestamp_dir=False)
opt = optimizers.Adam(1e-3)
gen = torch.Generator(device="cuda").manual_seed(1234)
img = torch.rand((B, 300, 300, 3), generator=gen, device="cuda")
gt = ops.pack_gt(*reversed(synth_batch_gt(0, B)))
out = None
xbuf = torch.empty((B, 300, 300, 8), dtype=torch.bfloat16, device="cuda")
def step():
    global out
    out = model.match_async(gt, out=out)
    x = ops.image_prep(img, normalize=True, out=xbuf)
    model._train_step(x, *out, opt)
def setting(spec, on):
    for kv in filter(None, spec.split(",")):
        k, v = kv.split("=")
        _lib.check(L.ssd_dev_knob(k.encode(), int(v) if on else -2147483648))      # INT_MIN = unset
def run():
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
res = {a: [] for a in args}
for rnd in range(rounds):
    for a in args:
        setting(a, True)
        res[a].append(run())
        setting(a, False)
for a in args:
    r = sorted(res[a])
    print("[%s] median %.3f ms (min %.3f max %.3f)" % (a, r[len(r) // 2], r[0], r[-1]), flush=True)
