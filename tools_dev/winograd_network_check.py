"""Dev (CPU): would Winograd F(2x2, 3x3) with bf16 MFMA operands stay inside this repository's tolerances?

The guardrail for any Winograd kernel: every tolerance in tests/test_conv_gpu.py and
tests/test_engine_gpu.py stays as written.  This script answers that before a kernel exists, by emulating the numerics in
the fp32 torch oracle: every 3x3 / stride 1 / SAME layer with >= 64 input channels (the layers a Winograd kernel would serve)
computes  V = B^T d B  and  U = G g G^T  in fp32 from the bf16 operands, ROUNDS THEM TO bf16 (what the MFMA would consume),
accumulates U*V over the channels in fp32 and applies A^T . A in fp32.  Compared with the direct oracle on the same weights
and input:
  * per layer: max |error| against the bound of test_conv_fwd_bwd (2^-7 x max(1, |y|max)) and the relative L2 error;
  * end to end: rel-L2 of loc / conf against the bound of test_forward_backward_vs_oracle (1e-2).
usage: winograd_network_check.py [seed [layer,layer,...|all [f16]]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import math
import numpy as np
import torch
import torch.nn.functional as F
from oracle import net_oracle as N
from ssd_object_detection_amd.engine import SSD300_TRUNK, SSD300_NUM_PRIORS

Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)
bf = lambda t: t.bfloat16().float()
stats = []


def winograd_conv(x, w, b, relu):
    """x [B,C,H,W] fp32 (bf16 values), w [N,3,3,C] (engine layout, bf16 values): SAME, stride 1."""
    Bn, C, H, W = x.shape
    He, We = H + (H & 1), W + (W & 1)
    xp = F.pad(x, (1, 1 + We - W, 1, 1 + He - H))
    tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)                     # [B, C, He/2, We/2, 4, 4]
    V = OPR(Bt @ tiles @ Bt.T)
    U = OPR(G @ w.permute(0, 3, 1, 2) @ G.T)                       # [N, C, 4, 4]
    M = torch.einsum('nckl,bcyxkl->bnyxkl', U, V)
    Y = At @ M @ At.T                                              # [B, N, He/2, We/2, 2, 2]
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(Bn, w.shape[0], He, We)[:, :, :H, :W] + b.view(1, -1, 1, 1)
    return y.relu() if relu else y


def conv(x, w, b, k, stride, same, relu, tag):
    direct = N.conv_tf(x, w, b, k, stride, same, relu)
    if k == 3 and stride == 1 and same and w.shape[3] >= 64 and (not ONLY or tag in ONLY):
        wino = winograd_conv(x, w, b, relu)
        err = (wino - direct).abs().max().item()
        bound = 2 ** -7 * max(1.0, direct.abs().max().item())
        stats.append((tag, tuple(x.shape[1:]), w.shape[0], err, bound, float((wino - direct).norm() / direct.norm())))
        return wino
    return direct


def forward(params, image_nhwc, use_winograd):
    rnd = lambda t: t.bfloat16().float()
    x = image_nhwc.permute(0, 3, 1, 2)
    feats = []
    for i, (kind, cin, cout, k, stride, mode, feat) in enumerate(SSD300_TRUNK):
        if kind == "conv":
            w, b = params["conv%d/kernel" % i], params["conv%d/bias" % i]
            x = rnd(conv(x, w, b, k, stride, mode == "same", True, "conv%d" % i) if use_winograd
                    else N.conv_tf(x, w, b, k, stride, mode == "same", True))
        else:
            if mode == "same" and x.shape[2] % 2:
                x = F.pad(x, (0, 1, 0, 1), value=float("-inf"))
            x = F.max_pool2d(x, 2, 2)
        if feat:
            feats.append(x)
    locs, confs = [], []
    for lvl, (f, n) in enumerate(zip(feats, SSD300_NUM_PRIORS)):
        w, b = params["head%d/kernel" % lvl], params["head%d/bias" % lvl]
        y = conv(f, w, b, 3, 1, True, False, "head%d" % lvl) if use_winograd else N.conv_tf(f, w, b, 3, 1, True, False)
        y = rnd(y).permute(0, 2, 3, 1)
        locs.append(y[..., :n * 4].reshape(1, -1, 4))
        confs.append(y[..., n * 4:].reshape(1, -1, 81))
    return torch.cat(locs, 1), torch.cat(confs, 1)


seed = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ONLY = set(sys.argv[2].split(",")) if len(sys.argv) > 2 and sys.argv[2] != "all" else set()
OPR = (lambda t: t.half().float()) if len(sys.argv) > 3 and sys.argv[3] == "f16" else bf   # rounding of the transformed operands
print("transformed operands rounded to", "fp16" if OPR is not bf else "bf16")      # restrict Winograd to these layers (e.g. conv4,conv7)
rng = np.random.default_rng(seed)
params = {}
for i, (kind, cin, cout, k, stride, mode, feat) in enumerate(SSD300_TRUNK):
    if kind == "conv":
        rc = 3 if i == 0 else cin
        lim = math.sqrt(6.0 / (k * k * (rc + cout)))
        w = rng.uniform(-lim, lim, (cout, k, k, cin)).astype(np.float32)
        if i == 0:
            w[..., 3:] = 0
        params["conv%d/kernel" % i] = bf(torch.from_numpy(w))
        params["conv%d/bias" % i] = torch.zeros(cout)
fms = [c for (kind, _, c, _, _, _, feat) in SSD300_TRUNK if feat]
for lvl, (c, n) in enumerate(zip(fms, SSD300_NUM_PRIORS)):
    parts = [rng.uniform(-math.sqrt(6.0 / (9 * (c + r))), math.sqrt(6.0 / (9 * (c + r))), (r, 3, 3, c)) for r in (n * 4, n * 81)]
    params["head%d/kernel" % lvl] = bf(torch.from_numpy(np.concatenate(parts, 0).astype(np.float32)))
    params["head%d/bias" % lvl] = torch.zeros(n * 85)
img = torch.zeros((1, 300, 300, 8))
img[..., :3] = bf((torch.from_numpy(rng.random((1, 300, 300, 3), dtype=np.float32)) - 0.5) * 2)
with torch.no_grad():
    loc_d, conf_d = forward(params, img, False)
    loc_w, conf_w = forward(params, img, True)
print("%-8s %-16s %5s %10s %10s %6s %10s" % ("layer", "input C,H,W", "Cout", "max|err|", "bound", "ok", "rel L2"))
for tag, shp, n, err, bound, rel in stats:
    print("%-8s %-16s %5d %10.3e %10.3e %6s %10.3e" % (tag, shp, n, err, bound, "yes" if err <= bound else "NO", rel))
rl = lambda a, b: float((a - b).norm() / b.norm())
print("end to end (Winograd in every eligible layer vs direct, both with bf16 activation storage): rel L2 loc %.3e conf %.3e"
      " (bound of test_forward_backward_vs_oracle: 1e-2, of which the direct HIP path uses ~1e-3)" % (rl(loc_w, loc_d), rl(conf_w, conf_d)))
