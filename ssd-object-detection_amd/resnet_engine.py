"""SSD512 with a ResNet-50 trunk on the gfx950 library -- BASELINE configs[4]'s network ("SSD512 ResNet-50 backbone ... 8732 ->
24564 anchors").  The reference has no counterpart: it hard-codes the 300 x 300 VGG network (models/ssd_model.py:46, 75-97), so
parity here is against this build's own plain-PyTorch restatement (oracle/net_oracle.py:forward_graph), not against the
reference.

Network: ResNet-50 v1.5 through conv4_x with frozen batch normalisation FOLDED into the convolutions (scale into the filters,
shift into the bias: every convolution is conv + bias, trainable) -- 7x7/2 stem, 3x3/2 max pooling, bottlenecks 1x1 -> 3x3
(stride on the 3x3) -> 1x1 with a projection shortcut where the shape changes, Add + ReLU -- feature maps conv3_x (64 x 64 x
512) and conv4_x (32 x 32 x 1024) at a 512 x 512 input, then the SSD recipe's extra stages (1x1 -> 3x3/2) down to 1 x 1: seven
levels 64, 32, 16, 8, 4, 2, 1 with 4, 6, 6, 6, 6, 4, 4 default boxes per cell = 24 564 anchors, heads as in the reference
(:153-162).  TF "SAME" padding everywhere (Keras semantics, as the rest of the engine).

The network is a DAG (residual adds), so forward / backward are a plain topological walk on ONE stream over the same C-ABI
entry points SSDEngine uses (1x1 layers on k_pw_gemm, 3x3 layers on the LDS-patch kernels, strided layers on the implicit-GEMM
kernels, heads' backward from the loss's compact rows) plus csrc/eltwise.hip (Add + ReLU, its gradient, 3x3/2 pooling).
Parameter storage, clip + Adam, checkpoints: inherited."""
import torch

from . import ops
from .engine import SSDEngine, SSD512_NUM_PRIORS


def resnet50_ssd512_graph():
    """Topologically ordered nodes: dict(op, src, cin, cout, k, stride, relu, feature).  src = producing node index
    (-1: the network input), for "add": (block output, shortcut)."""
    g = []

    def conv(src, cin, cout, k, stride, relu=True, feature=False):
        g.append(dict(op="conv", src=src, cin=cin, cout=cout, k=k, stride=stride, relu=relu, feature=feature))
        return len(g) - 1

    x = conv(-1, 8, 64, 7, 2)                                  # stem (image carried in 8 zero-padded channels)
    g.append(dict(op="pool3", src=x, cin=64, cout=64, k=3, stride=2, relu=False, feature=False))
    x, cin = len(g) - 1, 64
    for width, blocks, stride, feat in ((64, 3, 1, False), (128, 4, 2, True), (256, 6, 2, True)):     # conv2_x .. conv4_x
        for b in range(blocks):
            s = stride if b == 0 else 1
            a = conv(x, cin, width, 1, 1)
            a = conv(a, width, width, 3, s)                     # v1.5: the stride sits on the 3x3
            a = conv(a, width, 4 * width, 1, 1, relu=False)
            sc = conv(x, cin, 4 * width, 1, s, relu=False) if (b == 0) else x      # projection shortcut where the shape changes
            g.append(dict(op="add", src=(a, sc), cin=4 * width, cout=4 * width, k=0, stride=1, relu=True,
                          feature=feat and b == blocks - 1))
            x, cin = len(g) - 1, 4 * width
    for mid, out in ((256, 512), (128, 256), (128, 256), (128, 256), (128, 256)):   # extras: 32 -> 16 -> 8 -> 4 -> 2 -> 1
        x = conv(x, cin, mid, 1, 1)
        x = conv(x, mid, out, 3, 2, feature=True)
        cin = out
    return g


class ResNet50SSDEngine(SSDEngine):
    def __init__(self, classes=81, in_size=512, device="cuda", seed=0):
        self.graph = resnet50_ssd512_graph()
        super().__init__(classes=classes, in_size=in_size, trunk=[], num_priors=SSD512_NUM_PRIORS, device=device, seed=seed,
                         sparse_heads=True)
        self.relu_bits = None                  # ReLU masks from the bf16 activations (the sign-byte forms are a VGG-chain fusion)
        self.overlap_heads = False

    # ---------------------------------------------------------------- static planning
    def _plan_shapes(self):
        self.nodes, self.fm = [], []
        sizes = {-1: self.in_size}
        for i, nd in enumerate(self.graph):
            src = nd["src"][0] if nd["op"] == "add" else nd["src"]
            hin = sizes[src]
            if nd["op"] == "add":
                ho, pt = hin, 0
            else:
                ho, pt = ops.same_pad(hin, nd["k"], nd["stride"])
            sizes[i] = ho
            kind = "conv" if nd["op"] == "conv" else nd["op"]
            self.nodes.append(dict(kind=kind, cin=nd["cin"], cout=nd["cout"], k=nd["k"], stride=nd["stride"], pt=pt, pl=pt, hin=hin,
                                   hout=ho, feature=nd["feature"], same=True, src=nd["src"], relu=nd["relu"]))
            if nd["feature"]:
                self.fm.append((i, ho, nd["cout"]))
        assert len(self.fm) == len(self.num_priors), (len(self.fm), len(self.num_priors))
        self.level_off = [0]
        for (_, h, _), n in zip(self.fm, self.num_priors):
            self.level_off.append(self.level_off[-1] + h * h * n)
        self.A = self.level_off[-1]
        self.grids = tuple((h, h) for _, h, _ in self.fm)

    def _acts(self, B):
        c = self._act_cache.get(B)
        if c is None:
            c = super()._acts(B)
            dev = self.device
            c["pool3_code"] = {i: torch.empty((B, nd["hout"], nd["hout"], nd["cout"] // 8), dtype=torch.int32, device=dev)
                               for i, nd in enumerate(self.nodes) if nd["kind"] == "pool3"}
        return c

    def _in(self, acts, src):
        return acts[src + 1]                   # acts[0] = network input, acts[i + 1] = output of node i

    # ---------------------------------------------------------------- forward / backward
    def forward(self, x):
        B = x.shape[0]
        c = self._acts(B)
        acts = c["acts"]
        acts[0] = x
        self.bits_valid = set()
        for i, nd in enumerate(self.nodes):
            if nd["kind"] == "conv":
                wt, bt = self.conv_params[i]
                ops.conv2d_fwd(self._in(acts, nd["src"]), self.view(wt, self.param_bf16), self.view(bt, self.param), nd["stride"],
                               nd["pt"], nd["pl"], nd["hout"], nd["hout"], nd["relu"], out=acts[i + 1], ws=self._ws)
            elif nd["kind"] == "pool3":
                ops.maxpool3x3s2_fwd(self._in(acts, nd["src"]), out=acts[i + 1], code=c["pool3_code"][i])
            else:
                a, sc = nd["src"]
                ops.add_relu_fwd(acts[a + 1], acts[sc + 1], out=acts[i + 1])
        for lvl, (ni, _, _) in enumerate(self.fm):
            wt, bt = self.head_params[lvl]
            ops.conv2d_head_fwd(acts[ni + 1], self.view(wt, self.param_bf16), self.view(bt, self.param), c["loc"], c["conf"],
                                self.num_priors[lvl], self.classes, self.level_off[lvl], ws=self._ws)
        return c["loc"], c["conf"]

    def backward(self, dloc, dconf, on_ready=None, fused_adam=None, heads=None, on_dgrad=None):
        """Gradients of all parameters into self.grad from d(loss)/d(loc), d(loss)/d(conf) (or the loss's compact rows)."""
        assert fused_adam is None and on_dgrad is None, "the per-bucket optimizer schedule belongs to the VGG chain engine"
        if heads is None:
            heads = self.heads_from_dense(dloc, dconf)
        c = self._acts(heads.B)
        acts, gacts = c["acts"], list(c["gacts"])
        n = len(self.nodes)
        written = [False] * (n + 1)
        # heads: every feature-map gradient is written (masked by the map's own ReLU), then the trunk accumulates onto it
        hl, keep = self._head_layers(c)
        ops.heads_bwd_weight_sparse(heads, hl, ws=self._ws_hw)
        ops.heads_bwd_data_sparse(heads, hl, ws=self._ws_hz)
        del keep
        for ni, _, _ in self.fm:
            written[ni + 1] = True

        def relu_of(idx):                      # does activation acts[idx] carry its own ReLU?
            return idx > 0 and self.nodes[idx - 1]["relu"]

        for i in range(n - 1, -1, -1):
            nd = self.nodes[i]
            g = gacts[i + 1]
            assert written[i + 1], (i, nd)
            if nd["kind"] == "add":
                a, sc = nd["src"]
                # g = d loss / d relu(a + sc), already masked by this node's ReLU (its consumers' data gradients did that).
                # The block output (a linear 1x1 convolution) takes it as it is: alias, no copy.
                assert not written[a + 1] and not self.nodes[a]["relu"]
                gacts[a + 1] = g
                written[a + 1] = True
                if self.nodes[sc]["kind"] == "conv" and not self.nodes[sc]["relu"]:      # projection shortcut: linear too
                    assert not written[sc + 1]
                    gacts[sc + 1] = g
                else:                           # identity shortcut: masked by the source's own ReLU, summed with its other uses
                    ops.relu_mask_bwd(g, acts[sc + 1], out=gacts[sc + 1], accumulate=written[sc + 1])
                written[sc + 1] = True
                continue
            src = nd["src"]
            if nd["kind"] == "pool3":
                assert not written[src + 1]
                ops.maxpool3x3s2_bwd(c["pool3_code"][i], g, acts[src + 1].shape, out=gacts[src + 1])
                written[src + 1] = True
                continue
            wt, bt = self.conv_params[i]
            ops.conv2d_bwd_weight(acts[src + 1], g, nd["cout"], nd["k"], nd["stride"], nd["pt"], nd["pl"], dw=self.view(wt, self.grad),
                                  dbias=self.view(bt, self.grad), ws=self._ws)
            if on_ready:
                on_ready([wt.index, bt.index])
            if src < 0:
                continue                        # no gradient w.r.t. the image
            ops.conv2d_bwd_data(g, self.w_t[i], acts[src + 1] if relu_of(src + 1) else None, acts[src + 1].shape, nd["stride"],
                                nd["pt"], nd["pl"], accumulate=written[src + 1], out=gacts[src + 1], ws=self._ws)
            written[src + 1] = True
        if on_ready:
            on_ready([i for wt, bt in self.head_params for t in (wt, bt) for i in t.indices])
