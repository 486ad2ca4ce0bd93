"""Plain-PyTorch (CPU, fp32) restatement of the reference network -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Follows SSDObjectDetectionModel._build (models/ssd_model.py:74-171): Keras Conv2D / MaxPool2D semantics with
TensorFlow 'SAME' padding made explicit (pad_before = pad_total // 2), NHWC reshape + concatenation of the heads
(:166-167).  PARITY UNPINNED: TensorFlow is absent from this image, so this oracle follows the published layer
semantics only.  Gradients come from torch.autograd.

`emulate_bf16=True` rounds every activation to bfloat16 after its layer (straight-through gradient), which is
what the HIP path stores; weights are whatever is passed in (tests pass the bf16-rounded copies)."""
import torch
import torch.nn.functional as F


class _RoundBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g


def _same_pad(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return out, total // 2, total - total // 2


def conv_tf(x, w, b, k, stride, same, relu):
    """x NCHW, w [Cout,k,k,Cin] (the engine's layout), TF padding."""
    if same:
        _, pt, pb = _same_pad(x.shape[2], k, stride)
        _, pl, pr = _same_pad(x.shape[3], k, stride)
        x = F.pad(x, (pl, pr, pt, pb))
    y = F.conv2d(x, w.permute(0, 3, 1, 2), b, stride=stride)
    return y.relu() if relu else y


def forward(trunk, num_priors, classes, params, image_nhwc, emulate_bf16=True):
    """trunk: engine.SSD300_TRUNK; params: dict name -> tensor ('conv{i}/kernel', 'conv{i}/bias', 'head{l}/...').
    image_nhwc: [B,S,S,Cin0] float (already normalised / channel-padded).  Returns (loc [B,A,4], conf [B,A,classes])."""
    rnd = _RoundBF16.apply if emulate_bf16 else (lambda t: t)
    x = image_nhwc.permute(0, 3, 1, 2)
    feats = []
    for i, (kind, cin, cout, k, stride, mode, feat) in enumerate(trunk):
        if kind == "conv":
            x = rnd(conv_tf(x, params["conv%d/kernel" % i], params["conv%d/bias" % i], k, stride, mode == "same", True))
        else:
            if mode == "same" and x.shape[2] % 2:
                x = F.pad(x, (0, 1, 0, 1), value=float("-inf"))
            x = F.max_pool2d(x, 2, 2)
        if feat:
            feats.append(x)
    locs, confs = [], []
    B = x.shape[0]
    for lvl, (f, n) in enumerate(zip(feats, num_priors)):
        y = conv_tf(f, params["head%d/kernel" % lvl], params["head%d/bias" % lvl], 3, 1, True, False)
        y = rnd(y).permute(0, 2, 3, 1)                                  # NHWC
        locs.append(y[..., :n * 4].reshape(B, -1, 4))                   # Reshape((-1, 4)), :166
        confs.append(y[..., n * 4:].reshape(B, -1, classes))            # Reshape((-1, classes)), :167
    return torch.cat(locs, 1), torch.cat(confs, 1)


def forward_graph(graph, num_priors, classes, params, image_nhwc, emulate_bf16=True):
    """The same for a DAG trunk (resnet_engine.resnet50_ssd512_graph: ResNet-50 v1.5 with folded batch norm + SSD extras; no
    reference counterpart, the reference hard-codes its VGG chain): nodes dict(op = conv | pool3 | add, src, k, stride, relu,
    feature), conv{i} named by node index.  Keras / TF semantics: SAME padding, MaxPooling2D(3, 2, "same"), Add + ReLU."""
    rnd = _RoundBF16.apply if emulate_bf16 else (lambda t: t)
    outs = {-1: image_nhwc.permute(0, 3, 1, 2)}
    feats = []
    for i, nd in enumerate(graph):
        if nd["op"] == "conv":
            y = conv_tf(outs[nd["src"]], params["conv%d/kernel" % i], params["conv%d/bias" % i], nd["k"], nd["stride"], True, nd["relu"])
        elif nd["op"] == "pool3":
            x = outs[nd["src"]]
            _, pt, pb = _same_pad(x.shape[2], 3, 2)
            _, pl, pr = _same_pad(x.shape[3], 3, 2)
            y = F.max_pool2d(F.pad(x, (pl, pr, pt, pb), value=float("-inf")), 3, 2)
        else:
            a, sc = nd["src"]
            y = (outs[a] + outs[sc]).relu()
        outs[i] = rnd(y)
        if nd["feature"]:
            feats.append(outs[i])
    locs, confs = [], []
    B = image_nhwc.shape[0]
    for lvl, (f, n) in enumerate(zip(feats, num_priors)):
        y = conv_tf(f, params["head%d/kernel" % lvl], params["head%d/bias" % lvl], 3, 1, True, False)
        y = rnd(y).permute(0, 2, 3, 1)
        locs.append(y[..., :n * 4].reshape(B, -1, 4))
        confs.append(y[..., n * 4:].reshape(B, -1, classes))
    return torch.cat(locs, 1), torch.cat(confs, 1)
