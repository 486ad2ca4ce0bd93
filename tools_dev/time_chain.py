"""ssd_conv_chain at batch 64 on the SSD300 extras (nodes 17-22), forward and data gradient, timed alone (HIP events)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import ssd_object_detection_amd.ops as ops
from test_chain_gpu import EXTRAS, make_net, bits_of
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = 50
geo, x, ws, bs = make_net(ops, B, 10, EXTRAS, 3)
outs = [torch.empty((B, d["hout"], d["hout"], d["cout"]), dtype=torch.bfloat16, device="cuda") for d in geo]
bits = [torch.empty((B, d["hout"], d["hout"], d["cout"] // 8), dtype=torch.uint8, device="cuda") for d in geo]
pk = ops.chain_pack_weights([(w, None) for w in ws])
fl = [ops.chain_layer_fwd(w, p, b, o, d["s"], d["pt"], d["pt"], relu=True, relu_bits=rb) for d, w, p, b, o, rb in zip(geo, ws, pk, bs, outs, bits)]
def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("forward chain  B=%d: %.1f us" % (B, timed(lambda: ops.conv_chain(x, fl))))
acts = [x] + outs[:-1]
g_last = torch.randn(outs[-1].shape, device="cuda").bfloat16()
gouts = [torch.zeros_like(a) for a in acts]
wts = [ops.weight_transpose(w) for w in ws]
mb = [bits_of(a) for a in acts]
pkt = ops.chain_pack_weights([(w, None) for w in wts])
dl = [ops.chain_layer_dgrad(wts[i], pkt[i], gouts[i], geo[i]["s"], geo[i]["pt"], geo[i]["pt"], accumulate=(i % 2 == 0), mask_bits=mb[i])
      for i in range(len(geo) - 1, -1, -1)]
print("dgrad chain    B=%d: %.1f us" % (B, timed(lambda: ops.conv_chain(g_last, dl))))
print("pack 12 tensors: %.1f us" % timed(lambda: ops.chain_pack_weights(list(zip(ws, pk)) + list(zip(wts, pkt)))))
for i, (d, w, b, o) in enumerate(zip(geo, ws, bs, outs)):
    inp = acts[i]
    t = timed(lambda: ops.conv2d_fwd(inp, w, b, d["s"], d["pt"], d["pt"], d["hout"], d["hout"], True, out=o))
    gi = g_last if i == len(geo) - 1 else gouts[i + 1]
    t2 = timed(lambda: ops.conv2d_bwd_data(gi, wts[i], acts[i], tuple(acts[i].shape), d["s"], d["pt"], d["pt"], accumulate=(i % 2 == 0), out=gouts[i]))
    print("  node %d per-layer: fwd %.1f us, dgrad %.1f us" % (17 + i, t, t2))
print("single-layer chain launches (forward / data gradient):")
for i in range(len(geo)):
    d = geo[i]
    lf = [ops.chain_layer_fwd(ws[i], pk[i], bs[i], outs[i], d["s"], d["pt"], d["pt"], relu=True, relu_bits=bits[i])]
    tf = timed(lambda: ops.conv_chain(acts[i], lf))
    gi = g_last if i == len(geo) - 1 else gouts[i + 1]
    ld = [ops.chain_layer_dgrad(wts[i], pkt[i], gouts[i], d["s"], d["pt"], d["pt"], accumulate=(i % 2 == 0), mask_bits=mb[i])]
    td = timed(lambda: ops.conv_chain(gi, ld))
    print("  node %d: fwd %.1f us, dgrad %.1f us   (weights %d KB)" % (17 + i, tf, td, ws[i].numel() * 2 // 1024))
