"""Execution engine of the SSD300 network on the gfx950 library: parameter storage, forward, backward and
the optimizer step.  Python only sequences C-ABI calls (ops.py / _lib.py); every tensor operation runs in
libssd_hip.so.  torch provides device memory, streams and (for data parallelism) torch.distributed.

Network = SSDObjectDetectionModel._build of the reference (models/ssd_model.py:74-171): Keras VGG16 up to
block3_conv3, a SAME max-pool, three 38x38 convolutions, five extra stages, and per-level 3x3 loc/conf heads.
Activations are NHWC bf16, weights [Cout][k][k][Cin] bf16 (fp32 masters), accumulation fp32 on MFMA.
The 3-channel image is carried in 8 zero-padded channels (first-layer weights of channels 3..7 are and stay 0).
"""
import ctypes
import math
import os

import numpy as np
import torch

from . import _lib, ops

# (kind, cin, cout, k, stride, mode, feature_map)   -- mode: TF padding; every conv has ReLU
SSD300_TRUNK = [
    ("conv", 8, 64, 3, 1, "same", False),        # block1_conv1   (Cin 3 padded to 8)
    ("conv", 64, 64, 3, 1, "same", False),       # block1_conv2
    ("pool", 64, 64, 2, 2, "valid", False),      # block1_pool 300 -> 150
    ("conv", 64, 128, 3, 1, "same", False),      # block2_conv1
    ("conv", 128, 128, 3, 1, "same", False),     # block2_conv2
    ("pool", 128, 128, 2, 2, "valid", False),    # block2_pool 150 -> 75
    ("conv", 128, 256, 3, 1, "same", False),     # block3_conv1
    ("conv", 256, 256, 3, 1, "same", False),     # block3_conv2
    ("conv", 256, 256, 3, 1, "same", False),     # block3_conv3                     models/ssd_model.py:77-82
    ("pool", 256, 256, 2, 2, "same", False),     # MaxPool2D SAME 75 -> 38           :84
    ("conv", 256, 512, 3, 1, "same", False),     # :86
    ("conv", 512, 512, 3, 1, "same", False),     # :90
    ("conv", 512, 512, 1, 1, "same", True),      # :94   -> feature map 0 (38x38x512)
    ("conv", 512, 1024, 3, 2, "same", False),    # :102
    ("conv", 1024, 1024, 1, 1, "same", True),    # :107  -> feature map 1 (19x19x1024)
    ("conv", 1024, 256, 1, 1, "same", False),    # :113
    ("conv", 256, 512, 3, 2, "same", True),      # :117  -> feature map 2 (10x10x512)
    ("conv", 512, 128, 1, 1, "same", False),     # :124
    ("conv", 128, 256, 3, 2, "same", True),      # :128  -> feature map 3 (5x5x256)
    ("conv", 256, 128, 1, 1, "same", False),     # :135
    ("conv", 128, 256, 3, 1, "valid", True),     # :139  -> feature map 4 (3x3x256)
    ("conv", 256, 128, 1, 1, "same", False),     # :144
    ("conv", 128, 256, 3, 1, "valid", True),     # :148  -> feature map 5 (1x1x256)
]
SSD300_NUM_PRIORS = (4, 6, 6, 6, 4, 4)           # models/ssd_model.py:153
IMAGE_CHANNELS = 3

# BASELINE.json configs[4] stress geometry (no reference counterpart: the reference hard-codes 300 / 8732): the same
# network recipe at 512 x 512 with one more stride-2 stage, 7 feature levels 64, 32, 16, 8, 4, 2, 1 and 4, 6, 6, 6, 6, 4, 4
# default boxes per cell = 24 564 anchors.  (VGG-style trunk in bf16 -- not the ResNet-50 / fp8 of that config's title.)
SSD512_TRUNK = SSD300_TRUNK[:13] + [
    ("conv", 512, 1024, 3, 2, "same", False),    # 64 -> 32
    ("conv", 1024, 1024, 1, 1, "same", True),    # feature map 1 (32x32x1024)
    ("conv", 1024, 256, 1, 1, "same", False),
    ("conv", 256, 512, 3, 2, "same", True),      # feature map 2 (16x16x512)
    ("conv", 512, 128, 1, 1, "same", False),
    ("conv", 128, 256, 3, 2, "same", True),      # feature map 3 (8x8x256)
    ("conv", 256, 128, 1, 1, "same", False),
    ("conv", 128, 256, 3, 2, "same", True),      # feature map 4 (4x4x256)
    ("conv", 256, 128, 1, 1, "same", False),
    ("conv", 128, 256, 3, 2, "same", True),      # feature map 5 (2x2x256)
    ("conv", 256, 128, 1, 1, "same", False),
    ("conv", 128, 256, 3, 2, "same", True),      # feature map 6 (1x1x256)
]
SSD512_NUM_PRIORS = (4, 6, 6, 6, 6, 4, 4)


class ParamTensor:
    """One trainable variable of the reference (= one tf.clip_by_norm unit, models/ssd_model.py:249): `numel` elements at
    `offset` of the flat buffers, alone in optimizer blocks block0 .. block0 + nblocks - 1 (the rest of them is zero)."""

    def __init__(self, name, shape, offset, index, block0, nblocks):
        self.name, self.shape, self.offset, self.index = name, tuple(shape), offset, index
        self.numel = int(np.prod(shape))
        self.block0, self.nblocks = block0, nblocks


class FusedView:
    """Two adjacent ParamTensors read as one array: the loc and conf filters of a level are separate Keras layers
    (models/ssd_model.py:155-162: separate variables, separate clip norms) but one GEMM here.  The first part ends on a
    block boundary and the second starts on it, so the pair is contiguous without sharing an optimizer block."""

    def __init__(self, name, shape, parts):
        self.name, self.shape, self.parts = name, tuple(shape), tuple(parts)
        self.offset, self.index = parts[0].offset, parts[0].index
        self.numel = int(np.prod(shape))
        assert parts[1].offset == parts[0].offset + parts[0].numel and self.numel == parts[0].numel + parts[1].numel
        self.indices = [p.index for p in parts]


class SSDEngine:
    def __init__(self, classes=81, in_size=300, trunk=SSD300_TRUNK, num_priors=SSD300_NUM_PRIORS, device="cuda",
                 seed=0, sparse_heads=None):
        self.L = _lib.lib()
        self.classes, self.in_size, self.device = classes, in_size, torch.device(device)
        self.trunk, self.num_priors = list(trunk), tuple(num_priors)
        self.block = self.L.ssd_opt_block_elems()
        # the heads' backward pass from the loss's compact gradient rows (csrc/sparse.hip); SSD_SPARSE_HEADS=0: the dense
        # kernels on the scattered gradient (tests compare the two)
        self.sparse_heads = (os.environ.get("SSD_SPARSE_HEADS", "1") == "1") if sparse_heads is None else bool(sparse_heads)
        # the trailing convolutions on LDS-sized maps (the extras behind the 19x19 map) as ONE launch, forward and data gradient
        # (ops.conv_chain); a set: {"fwd", "bwd"}, emptied by a refusal of the kernel
        self.chain = {"1": {"fwd", "bwd"}, "fwd": {"fwd"}, "bwd": {"bwd"}}.get(os.environ.get("SSD_CHAIN", "1"), set())
        self._plan_shapes()
        self._plan_params()
        self._alloc_params()
        self.init_params(seed)
        self._act_cache = {}
        self._ws = ops.MatchWorkspace()
        self._ws_hz = ops.MatchWorkspace()     # Z of the sparse head data gradient
        self._ws_hw = ops.MatchWorkspace()     # slabs of the sparse head weight gradient
        self._side = None
        self.overlap_heads = os.environ.get("SSD_OVERLAP_HEADS", "1") != "0" and self.device.type == "cuda"
        self.step_count = 0
        self.skip_fullres = os.environ.get("SSD_SKIP_FULLRES", "1") == "1"   # pooled convs store the pooled map only
        # schedule switches of the host program (read once, here): third stream for the small heads of the forward pass; dense
        # head path only: the two large heads' backward on the side stream, their gradient packing there too
        self.tail_stream = os.environ.get("SSD_TAIL_STREAM", "1") == "1"
        self.big_heads_side = os.environ.get("SSD_BIG_HEADS_SIDE", "1") == "1"
        self.pack_side = os.environ.get("SSD_PACK_SIDE", "1") == "1"
        # 2 = weight gradients alternate between two side streams: measured 8.99 -> 9.36 ms (three MFMA kernels share the CUs'
        # LDS and registers badly); 1 = one side stream
        self.side_streams = int(os.environ.get("SSD_SIDE_STREAMS", "1"))
        # the split reductions behind the weight-gradient kernels (0.55 ms of small HBM-bound launches per step, each in front
        # of the next layer's kernel on the side stream) on a stream of their own (ssd_set_wgrad_reduce_stream): measured
        # 9.02 -> 9.34 ms -- beside TWO MFMA kernels the small launches starve, and the next-but-one layer waits for them: off
        self.reduce_stream = os.environ.get("SSD_REDUCE_STREAM", "0") == "1"
        self.batch_chain_wgrads = os.environ.get("SSD_BATCH_CHAIN_WGRADS", "1") == "1"
        self.batch_chain_front = os.environ.get("SSD_BATCH_CHAIN_FRONT", "0") == "1"   # measured 9.015 -> 9.037 ms: off
        self.wgrad_group = int(os.environ.get("SSD_WGRAD_GROUP", "3"))
        self.wgrad_on_main = set(int(v) for v in os.environ.get("SSD_WGRAD_ON_MAIN", "").split(",") if v.strip())   # trunk nodes      # weight-gradient launches per cross-stream wait
        self.split_heads_dgrad = int(os.environ.get("SSD_SPLIT_HEADS_DGRAD", "2"))   # 0 one call, 1 small | large levels, 2 ... and one call per large level
        # the heads of the maps the forward chain produces (all available at once, behind one launch): every other one on the
        # main stream instead of queueing all of them on the third
        self.chain_heads_split = os.environ.get("SSD_CHAIN_HEADS_SPLIT", "1") == "1"
        self.chain_prefetch = os.environ.get("SSD_CHAIN_PREFETCH", "1") == "1"
        # the large levels' gradient maps (38x38, 19x19: 140 MB at batch 64) are cleared during the FORWARD pass, on the third
        # stream under a compute-bound layer, and the heads' data gradient then writes only the ~5 % of pixels a gradient row
        # reaches: the 140 MB of zero stores leave the window behind the loss, where nothing large can run yet
        self.prezero_maps = os.environ.get("SSD_PREZERO_MAPS", "0") == "1"   # measured neutral (9.046 vs 9.060 ms): off
        self._prezeroed = set()
        # fused-optimizer buckets that run at the END of the main stream instead of in the side stream's queue: the side stream (weight
        # gradients) is the longer chain, the main stream finishes ~0.5 ms earlier (round 4, same-box A/B: 1 -> 4 buckets -0.06 ms)
        self.opt_defer = int(os.environ.get("SSD_OPT_DEFER", "4"))
        # the heads' bucket (45 % of the parameters) is complete right after the loss, where the side stream would run its
        # HBM-bound update next to the extras' latency-bound data-gradient chain: 1 = run it at the main stream's tail instead
        self.opt_defer_heads = int(os.environ.get("SSD_OPT_DEFER_HEADS", "0"))
        self.pool_only = {}                    # node -> whether a pool-only kernel serves it (learned at the first call)
        self.fuse_unpool = {} if os.environ.get("SSD_FUSE_UNPOOL", "1") == "1" else None    # node -> data gradient un-pools itself
        # activation index -> its sign bits are written by the forward kernel (learned at the first call); data-gradient
        # key (str) -> that kernel reads them
        self.relu_bits = {} if os.environ.get("SSD_RELU_BITS", "1") == "1" else None
        self.bits_valid = set()
        # second layer's data gradient and first layer's weight gradient in one kernel (the gradient w.r.t. the first layer's
        # output has no other consumer and is never stored); False after a refusal
        self.fuse_first = os.environ.get("SSD_FUSE_FIRST", "1") == "1"
        self.wgrad_probe = None                # dict(nodes={...}, events=[]): time those layers' weight-gradient launches in the step

    # ---------------------------------------------------------------- static planning
    def _plan_shapes(self):
        s = self.in_size
        self.nodes = []                       # dicts: kind, cin, cout, k, stride, pt, pl, hin, hout, feature
        self.fm = []                          # (node index, h, c)
        for kind, cin, cout, k, stride, mode, feat in self.trunk:
            if mode == "same":
                ho, pt = ops.same_pad(s, k, stride)
            else:
                ho, pt = ops.valid_out(s, k, stride), 0
            self.nodes.append(dict(kind=kind, cin=cin, cout=cout, k=k, stride=stride, pt=pt, pl=pt, hin=s, hout=ho,
                                   feature=feat, same=(mode == "same")))
            s = ho
            if feat:
                self.fm.append((len(self.nodes) - 1, ho, cout))
        assert len(self.fm) == len(self.num_priors)
        # the chain: the longest run of trailing convolutions whose input and output maps have <= 112 pixels and whose channel
        # counts are multiples of 128 (ssd_conv_chain's limits), each behind another convolution (its ReLU is the mask)
        j = len(self.nodes)
        while (j > 1 and len(self.nodes) - j < _lib.SSD_CHAIN_MAX_LAYERS and self.nodes[j - 1]["kind"] == "conv"
               and self.nodes[j - 2]["kind"] == "conv" and self.nodes[j - 1]["hin"] ** 2 <= 112 and self.nodes[j - 1]["hout"] ** 2 <= 112
               and self.nodes[j - 1]["cin"] % 128 == 0 and self.nodes[j - 1]["cout"] % 128 == 0):
            j -= 1
        self.chain_start = j if len(self.nodes) - j >= 2 else None
        if any(c % 128 for _, _, c in self.fm) or len(self.fm) > _lib.SSD_MAX_LEVELS:
            self.sparse_heads = False
        self.level_off = [0]
        for (_, h, _), n in zip(self.fm, self.num_priors):
            self.level_off.append(self.level_off[-1] + h * h * n)
        self.A = self.level_off[-1]            # 8732 for SSD300 (models/ssd_model.py:221)
        self.grids = tuple((h, h) for _, h, _ in self.fm)

    def _plan_params(self):
        self.tensors = []
        off = 0

        def add(name, shape, end_aligned=False):
            nonlocal off
            numel = int(np.prod(shape))
            nb = (numel + self.block - 1) // self.block
            # end-aligned: the tensor ENDS on a block boundary (the next one starts there: one contiguous GEMM operand over two
            # optimizer variables); its start must stay 16-byte aligned in the bf16 copy too (DMA, float4)
            start = off + (nb * self.block - numel if end_aligned else 0)
            if start % 8:
                raise ValueError("%s: %d elements do not end-align on a 16-byte boundary (per-cell anchor counts must make "
                                 "n*4 a multiple of 8, i.e. even: the reference's are 4 and 6)" % (name, numel))
            t = ParamTensor(name, shape, start, len(self.tensors), off // self.block, nb)
            self.tensors.append(t)
            off += nb * self.block
            return t

        self.conv_params = {}
        for i, nd in enumerate(self.nodes):
            if nd["kind"] != "conv":
                continue
            self.conv_params[i] = (add("conv%d/kernel" % i, (nd["cout"], nd["k"], nd["k"], nd["cin"])),
                                   add("conv%d/bias" % i, (nd["cout"],)))
        self.head_params = []
        for lvl, ((_, h, c), n) in enumerate(zip(self.fm, self.num_priors)):
            # loc filters (n*4) then conf filters (n*classes): one fused GEMM over two variables each (kernel, bias)
            nl, nc = n * 4, n * self.classes
            lk, ck = add("head%d/loc_kernel" % lvl, (nl, 3, 3, c), True), add("head%d/conf_kernel" % lvl, (nc, 3, 3, c))
            lb, cb = add("head%d/loc_bias" % lvl, (nl,), True), add("head%d/conf_bias" % lvl, (nc,))
            self.head_params.append((FusedView("head%d/kernel" % lvl, (nl + nc, 3, 3, c), (lk, ck)),
                                     FusedView("head%d/bias" % lvl, (nl + nc,), (lb, cb))))
        self.n_flat = off
        self.n_params = sum(t.numel for t in self.tensors)

    def _alloc_params(self):
        dev, n = self.device, self.n_flat
        self.param = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad_acc = None
        self.adam_m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.adam_v = torch.zeros(n, dtype=torch.float32, device=dev)
        self.param_bf16 = torch.zeros(n, dtype=torch.bfloat16, device=dev)
        nb = n // self.block
        tbo = np.zeros(len(self.tensors) + 1, np.int32)
        bt = np.zeros(nb, np.int32)
        for i, t in enumerate(self.tensors):
            b0, b1 = t.block0, t.block0 + t.nblocks
            tbo[i], tbo[i + 1] = b0, b1
            bt[b0:b1] = i
        self.tensor_block_off = torch.from_numpy(tbo).to(dev)
        self.block_tensor = torch.from_numpy(bt).to(dev)
        self.sq_partial = torch.empty(nb, dtype=torch.float64, device=dev)
        self.clip_scale = torch.empty(len(self.tensors), dtype=torch.float32, device=dev)
        self.grad_norms = torch.empty(len(self.tensors), dtype=torch.float32, device=dev)
        # transposed weights for the data gradient (own buffers)
        self.w_t = {}
        for i, nd in enumerate(self.nodes):
            if nd["kind"] == "conv" and i > 0:
                self.w_t[i] = torch.empty((nd["cin"], nd["k"], nd["k"], nd["cout"]), dtype=torch.bfloat16, device=dev)
        # fragment-packed copies of the chain layers' filters (ops.conv_chain's operands), forward and data gradient
        self.chain_pk_fwd, self.chain_pk_bwd = {}, {}
        if getattr(self, "chain_start", None) is not None and self.chain:
            for i in range(self.chain_start, len(self.nodes)):
                nd = self.nodes[i]
                self.chain_pk_fwd[i] = torch.empty((nd["cout"], nd["k"], nd["k"], nd["cin"]), dtype=torch.bfloat16, device=dev)
                self.chain_pk_bwd[i] = torch.empty_like(self.w_t[i])
        self.head_npad = [(n * (4 + self.classes) + 7) // 8 * 8 for n in self.num_priors]
        # transposed head filters: [Cin][3][3][npad] flipped for the dense data gradient, or tap-major [3][3][Cin][npad] for
        # the sparse one
        self.head_w_t = [torch.empty((3, 3, c, npad) if self.sparse_heads else (c, 3, 3, npad), dtype=torch.bfloat16, device=dev)
                         for (_, _, c), npad in zip(self.fm, self.head_npad)]

    # ---------------------------------------------------------------- parameter views
    def view(self, t, buf):
        return buf[t.offset:t.offset + t.numel].view(t.shape)

    def init_params(self, seed=0):
        """Keras defaults: glorot_uniform kernels, zero biases (the reference's VGG part loads ImageNet weights
        from the network, which is unavailable offline -- SURVEY.md F9)."""
        rng = np.random.default_rng(seed)
        host = np.zeros(self.n_flat, np.float32)
        for t in self.tensors:
            if not t.name.endswith("kernel"):
                continue
            cout, k, _, cin = t.shape
            if t.name.startswith("head"):                       # loc and conf: two separate Keras layers (:155-162)
                lim = math.sqrt(6.0 / (k * k * cin + k * k * cout))
                w = rng.uniform(-lim, lim, (cout, k, k, cin))
            else:
                real_cin = IMAGE_CHANNELS if t.name == "conv0/kernel" else cin
                lim = math.sqrt(6.0 / (k * k * real_cin + k * k * cout))
                w = rng.uniform(-lim, lim, (cout, k, k, cin))
                if real_cin != cin:
                    w[..., real_cin:] = 0.0
            host[t.offset:t.offset + t.numel] = w.astype(np.float32).reshape(-1)
        self.param.copy_(torch.from_numpy(host))
        self.adam_m.zero_()
        self.adam_v.zero_()
        self.step_count = 0
        self.refresh_weights(cast=True)

    def refresh_weights(self, cast=False, tensors=None):
        """bf16 copy (if not already written by the optimizer kernel) + transposed copies for the data gradient.
        tensors=(t0, t1): only the copies of parameter tensors t0..t1-1 (the per-bucket optimizer step)."""
        if cast:
            ops.cast_bf16(self.param, self.param_bf16)
        if getattr(self, "_tr_desc", None) is None:         # {src, dst, Cout, k, Cin, Cout_pad} per transposed copy
            rows, tiles = [], 0
            pairs = [(self.conv_params[i][0], wt) for i, wt in self.w_t.items()]
            n_trunk = len(pairs)
            pairs += [(self.head_params[lvl][0], wt) for lvl, wt in enumerate(self.head_w_t)]
            for r, (pt, wt) in enumerate(pairs):
                cout, k, _, cin = pt.shape
                cpad = wt.shape[-1]
                kfield = k | (0x100 if (self.sparse_heads and r >= n_trunk) else 0)     # bit 8: tap-major, not flipped
                rows.append([self.view(pt, self.param_bf16).data_ptr(), wt.data_ptr(), cout, kfield, cin, cpad])
                tiles = max(tiles, ((cpad + 31) // 32) * ((cin + 31) // 32) * k * k)
            self._tr_desc = torch.tensor(rows, dtype=torch.int64, device=self.device)
            self._tr_tiles = tiles
            self._tr_tensor = [pt.index for pt, _ in pairs]      # ascending: trunk kernels, then head kernels
        r0, r1 = 0, self._tr_desc.shape[0]
        if tensors is not None:
            inside = [r for r, t in enumerate(self._tr_tensor) if tensors[0] <= t < tensors[1]]
            if not inside:
                return
            r0, r1 = inside[0], inside[-1] + 1
            assert inside == list(range(r0, r1))
        _lib.check(self.L.ssd_weight_transpose_batched(ops._ptr(self._tr_desc[r0:]), r1 - r0, self._tr_tiles,
                                                       ops._stream()))
        if self.chain_pk_fwd:                                # ... and the chain's fragment-packed copies of the same tensors
            items = []
            for i in self.chain_pk_fwd:
                wt = self.conv_params[i][0]
                if tensors is None or tensors[0] <= wt.index < tensors[1]:
                    items += [(self.view(wt, self.param_bf16), self.chain_pk_fwd[i]), (self.w_t[i], self.chain_pk_bwd[i])]
            if items:
                ops.chain_pack_weights(items)

    # ---------------------------------------------------------------- activations
    def _acts(self, B):
        c = self._act_cache.get(B)
        if c is None:
            dev = self.device
            acts = [None]                         # acts[0] = network input, set per call
            gacts = [None]
            for nd in self.nodes:
                shape = (B, nd["hout"], nd["hout"], nd["cout"])
                acts.append(torch.empty(shape, dtype=torch.bfloat16, device=dev))
                gacts.append(torch.empty(shape, dtype=torch.bfloat16, device=dev))
            loc = torch.empty((B, self.A, 4), dtype=torch.bfloat16, device=dev)
            conf = torch.empty((B, self.A, self.classes), dtype=torch.bfloat16, device=dev)
            packed = [torch.empty((B, h * h, npad), dtype=torch.bfloat16, device=dev)
                      for (_, h, _), npad in zip(self.fm, self.head_npad)]
            pool_code = {i: torch.empty((B, nd["hout"], nd["hout"], nd["cout"] // 8), dtype=torch.int32, device=dev)
                         for i, nd in enumerate(self.nodes) if nd["kind"] == "pool"}
            # ReLU sign bits of every convolution output (one byte per pixel and 8 channels): what the data gradients read
            # instead of the bf16 activation
            rbits = {i + 1: torch.empty((B, nd["hout"], nd["hout"], nd["cout"] // 8), dtype=torch.uint8, device=dev)
                     for i, nd in enumerate(self.nodes) if nd["kind"] == "conv" and nd["cout"] % 8 == 0}
            hgb = None
            if self.sparse_heads:
                hgb = ops.HeadGradBuffers(B, [h * h for _, h, _ in self.fm], self.num_priors, self.head_npad, device=dev)
            c = dict(acts=acts, gacts=gacts, loc=loc, conf=conf, packed=packed, pool_code=pool_code, rbits=rbits, hgb=hgb)
            self._act_cache = {B: c}              # keep one batch size resident
        return c

    # ---------------------------------------------------------------- forward / backward
    # Two HIP streams.  The two large heads (38x38 and 19x19 maps) are independent of the small tail of the network
    # (conv 12-19 and heads 2-5: ~45 short, latency-bound launches that leave most CUs idle), so they run on a side
    # stream next to it, forward and backward; results do not depend on this (disjoint outputs, the one shared
    # accumulation target is ordered by an event).
    SIDE_HEADS = (0, 1)

    def _side_stream(self):
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.device, priority=int(os.environ.get("SSD_SIDE_PRIO", "0")))
            self._ws_side = ops.MatchWorkspace()
        return self._side

    def forward(self, x):
        """x: bf16 [B, S, S, 8] (ops.image_prep).  Returns (loc bf16 [B,A,4], conf bf16 [B,A,classes])."""
        B = x.shape[0]
        c = self._acts(B)
        acts = c["acts"]
        acts[0] = x
        main = torch.cuda.current_stream()
        side = self._side_stream() if self.overlap_heads else None
        tail = None
        if side is not None and self.tail_stream:
            if getattr(self, "_tail", None) is None:
                self._tail = torch.cuda.Stream(device=self.device)
                self._ws_tail = ops.MatchWorkspace()
            tail = self._tail
        fm_level = {ni: lvl for lvl, (ni, _, _) in enumerate(self.fm)}

        def head(lvl, ws):
            ni = self.fm[lvl][0]
            wt, bt = self.head_params[lvl]
            ops.conv2d_head_fwd(acts[ni + 1], self.view(wt, self.param_bf16), self.view(bt, self.param), c["loc"],
                                c["conf"], self.num_priors[lvl], self.classes, self.level_off[lvl], ws=ws)

        def after_node(i):
            """Launch the head of the feature map node i produced, where it runs next to the trunk."""
            lvl = fm_level.get(i)
            if side is not None and lvl in self.SIDE_HEADS:
                ev = torch.cuda.Event()
                ev.record(main)
                with torch.cuda.stream(side):
                    side.wait_event(ev)
                    head(lvl, self._ws_side)
            elif tail is not None and lvl is not None:
                # the small levels' heads (10x10 and below: a few workgroups each) on a third stream, as soon as their map
                # exists: next to the extras' chain on the main stream and the 19x19 head on the side stream they cost
                # nothing, behind them they were 190 us of a nearly idle GPU
                ev = torch.cuda.Event()
                ev.record(main)
                with torch.cuda.stream(tail):
                    tail.wait_event(ev)
                    head(lvl, self._ws_tail)

        self.bits_valid = set()
        self._prezeroed = set()
        big_maps = [ni + 1 for lvl, (ni, h, ch) in enumerate(self.fm) if B * h * h >= 16384]
        clear_at = min(9, len(self.nodes) - 1)            # under block 4 (compute-bound 3x3 layers on 38x38 maps)
        for i, nd in enumerate(self.nodes):
            if (i == clear_at and tail is not None and self.prezero_maps and self.sparse_heads and self.split_heads_dgrad
                    and big_maps and len(big_maps) < len(self.fm)):
                ev = torch.cuda.Event()
                ev.record(main)
                with torch.cuda.stream(tail):
                    tail.wait_event(ev)
                    for a in big_maps:
                        c["gacts"][a].zero_()
                self._prezeroed = set(big_maps)
            if (self.chain_start is not None and i == self.chain_start - 1 and "fwd" in self.chain and tail is not None
                    and self.chain_prefetch):
                ev = torch.cuda.Event()                   # the chain's packed filters into L2 while the layer in front of it runs
                ev.record(main)
                with torch.cuda.stream(tail):
                    tail.wait_event(ev)
                    ops.chain_prefetch([self.chain_pk_fwd[j] for j in range(self.chain_start, len(self.nodes))])
            if i == self.chain_start and "fwd" in self.chain:
                # nodes i .. end in one launch, one workgroup per image (ops.conv_chain)
                want_bits = self.relu_bits is not None
                layers = []
                for j in range(i, len(self.nodes)):
                    ndj = self.nodes[j]
                    wt, bt = self.conv_params[j]
                    rb = c["rbits"].get(j + 1) if want_bits else None
                    layers.append(ops.chain_layer_fwd(self.view(wt, self.param_bf16), self.chain_pk_fwd[j], self.view(bt, self.param), acts[j + 1],
                                                      ndj["stride"], ndj["pt"], ndj["pl"], relu=True, relu_bits=rb))
                try:
                    ops.conv_chain(acts[i], layers)
                    nth = 0
                    for j in range(i, len(self.nodes)):
                        if layers[j - i]["relu_bits"] is not None:
                            self.bits_valid.add(j + 1)
                        lvl = fm_level.get(j)
                        if lvl is not None and tail is not None and self.chain_heads_split and lvl not in self.SIDE_HEADS:
                            nth += 1
                            if nth % 2 == 0:
                                head(lvl, self._ws)           # (the main stream has nothing else left to do)
                                continue
                        after_node(j)
                    break
                except NotImplementedError:           # SSD_ERR_UNSUPPORTED: nothing launched
                    self.chain = set()
            if nd["kind"] == "conv":
                wt, bt = self.conv_params[i]
                nxt = self.nodes[i + 1] if i + 1 < len(self.nodes) else None
                if nxt is not None and nxt["kind"] == "pool":      # conv + the pooling behind it in one call
                    # nothing but the pooling reads this conv's full-resolution output (the backward pass works from the
                    # pooled map and the winner codes): ask for the pooled map only, where a fused kernel serves the layer
                    pool_only = self.pool_only.get(i, self.skip_fullres)
                    args = (acts[i], self.view(wt, self.param_bf16), self.view(bt, self.param), nd["stride"], nd["pt"],
                            nd["pl"], nd["hout"], nd["hout"], True, nxt["hout"] * 2 != nxt["hin"])
                    kw = dict(out=acts[i + 1], pool_out=acts[i + 2], code=c["pool_code"][i + 1], ws=self._ws)
                    if pool_only:
                        try:
                            ops.conv2d_fwd_pool(*args, pool_only=True, **kw)
                        except ValueError:            # SSD_ERR_VALUE: no pooling kernel for this shape, nothing launched
                            pool_only = False
                    self.pool_only[i] = pool_only
                    if not pool_only:
                        ops.conv2d_fwd_pool(*args, **kw)
                else:
                    args = (acts[i], self.view(wt, self.param_bf16), self.view(bt, self.param), nd["stride"], nd["pt"], nd["pl"],
                            nd["hout"], nd["hout"])
                    done = False
                    if self.relu_bits is not None and self.relu_bits.get(i + 1, True) and (i + 1) in c["rbits"]:
                        try:
                            ops.conv2d_fwd_relubits(*args, c["rbits"][i + 1], out=acts[i + 1], ws=self._ws)
                            done = True
                            self.bits_valid.add(i + 1)
                        except NotImplementedError:   # SSD_ERR_UNSUPPORTED: nothing launched
                            pass
                        self.relu_bits[i + 1] = done
                    if not done:
                        ops.conv2d_fwd(*args, True, out=acts[i + 1], ws=self._ws)
            elif i == 0 or self.nodes[i - 1]["kind"] != "conv":
                ops.maxpool2x2_fwd_argmax(acts[i], out=acts[i + 1], code=c["pool_code"][i],
                                          same=nd["hout"] * 2 != nd["hin"])
            after_node(i)
        for lvl in range(len(self.fm)):
            if side is None or (lvl not in self.SIDE_HEADS and tail is None):
                head(lvl, self._ws)
        if side is not None:
            main.wait_stream(side)
            if tail is not None:
                main.wait_stream(tail)
        return c["loc"], c["conf"]

    def head_grad_buffers(self, B):
        """The ops.HeadGradBuffers the loss writes for a batch of B (None when the dense head path is selected)."""
        return self._acts(B)["hgb"]

    def heads_from_dense(self, dloc, dconf):
        """Compact rows from a dense (dloc, dconf) pair -- for callers that hold an arbitrary gradient (tests, the oracle
        comparison); the training step gets the rows from the loss directly.  Synchronises the host."""
        B = dloc.shape[0]
        hgb = self.head_grad_buffers(B)
        counts = []
        for lvl, (_, h, _) in enumerate(self.fm):
            npad = self.head_npad[lvl]
            packed = ops.head_grad_pack(dloc, dconf, h * h, self.num_priors[lvl], self.classes, npad,
                                        self.level_off[lvl]).view(B * h * h, npad)
            idx = (packed != 0).any(dim=1).nonzero().squeeze(1)
            k = int(idx.numel())
            hgb.rows[lvl][:k] = packed[idx]
            hgb.pixel_of_row[lvl][:k] = idx.int()
            hgb.row_of_pixel[lvl].fill_(-1)
            hgb.row_of_pixel[lvl][idx] = torch.arange(k, dtype=torch.int32, device=idx.device)
            counts.append(k)
        hgb.count[:len(counts)] = torch.tensor(counts, dtype=torch.int32, device=hgb.count.device)
        return hgb

    def _head_layers(self, c):
        acts, gacts = c["acts"], c["gacts"]
        idx = [ni + 1 for ni, _, _ in self.fm]
        use_bits = [self.relu_bits is not None and a in self.bits_valid for a in idx]
        hl, keep = ops.head_layers(
            [acts[a] for a in idx], self.head_w_t, [gacts[a] for a in idx],
            [self.view(wt, self.grad) for wt, _ in self.head_params], [self.view(bt, self.grad) for _, bt in self.head_params],
            [n * (4 + self.classes) for n in self.num_priors],
            relu_bits=[c["rbits"][a] if ub else None for a, ub in zip(idx, use_bits)],
            relu_src=[None if ub else acts[a] for a, ub in zip(idx, use_bits)])
        return hl, keep

    def bucket_gates(self, buckets):
        """For tensor ranges [(t0, t1)] (the gradient exchange's buckets): the trunk node whose data gradient is the last
        reader of the range's transposed weights (the lowest convolution in it), None for a range of head tensors only."""
        gates = []
        for t0, t1 in buckets:
            nodes = [i for i, (wt, bt) in self.conv_params.items() if t0 <= wt.index < t1 or t0 <= bt.index < t1]
            gates.append(min(nodes) if nodes else None)
        return gates

    def backward(self, dloc, dconf, on_ready=None, fused_adam=None, heads=None, on_dgrad=None):
        try:
            return self._backward(dloc, dconf, on_ready, fused_adam, heads, on_dgrad)
        finally:
            if getattr(self, "_red_active", None) is not None:   # (an exception between set and reset: do not leave the
                self.L.ssd_set_wgrad_reduce_stream(None)          #  library sending other callers' reductions to our stream)
                self._red_active = None

    def _backward(self, dloc, dconf, on_ready=None, fused_adam=None, heads=None, on_dgrad=None):
        """Gradients of all parameters into self.grad (flat fp32) from d(loss)/d(loc), d(loss)/d(conf).
        on_ready([tensor indices]) is called right after the launches that complete those tensors' gradients (on the
        stream that runs them: an event recorded there covers them).
        fused_adam = dict(lr, beta1, beta2, eps, clip): clip_by_norm + Adam + weight copies run per bucket of tensors
        (opt_buckets) on the side stream as soon as the bucket's gradients exist and the last data gradient that reads
        its transposed weights has been enqueued -- the optimizer disappears under the rest of the backward pass.

        The data-gradient chain (the critical path) runs on the current stream, every weight gradient on the side
        stream as soon as its input gradient exists: the split reductions and round tails of one overlap the MFMA
        work of the other.  Each launch still sums in a fixed order, so results do not depend on the overlap."""
        if heads is None and self.sparse_heads:
            heads = self.heads_from_dense(dloc, dconf)
        B = heads.B if heads is not None else dloc.shape[0]
        c = self._acts(B)
        acts, gacts = c["acts"], c["gacts"]
        written = [False] * len(acts)
        main = torch.cuda.current_stream()
        side = self._side_stream() if self.overlap_heads else None
        opt_at = {}
        deferred, defer_nodes = [], set()
        if fused_adam is not None:
            ndefer = self.opt_defer
            trunk_nodes = [node for _, _, node in self.opt_buckets() if node is not None]
            # the last bucket (lowest node) follows its own weight gradients on the side stream; the `ndefer` before it go to the
            # END of the main stream, which finishes its chain ~0.3 ms before the side stream does
            defer_nodes = set(trunk_nodes[max(0, len(trunk_nodes) - 1 - ndefer):len(trunk_nodes) - 1]) if ndefer > 0 else set()
        if fused_adam is not None:
            self.step_count += 1
            t = self.step_count
            hp = fused_adam
            lr_t = hp["lr"] * math.sqrt(1.0 - hp["beta2"] ** t) / (1.0 - hp["beta1"] ** t)
            opt_at = {node: (t0, t1) for t0, t1, node in self.opt_buckets()}

        def opt_bucket(node):
            """Called once the data gradient of `node` (None: of every head) is enqueued on the main stream."""
            if on_dgrad is not None:
                on_dgrad(node)
            if node in opt_at:
                t0, t1 = opt_at.pop(node)
                if side is not None and ((node is not None and node in defer_nodes) or (node is None and self.opt_defer_heads)):
                    flush_side()
                    ev = torch.cuda.Event()
                    ev.record(side)                    # the bucket's weight gradients are all enqueued there by now
                    ev2 = None
                    if getattr(self, "_side2", None) is not None and self.side_streams == 2 and on_ready is None and on_dgrad is None:
                        ev2 = torch.cuda.Event()
                        ev2.record(self._side2)
                    if getattr(self, "_red_active", None) is not None:
                        ev2 = torch.cuda.Event()
                        ev2.record(self._red_active)
                    deferred.append((t0, t1, ev, ev2))
                    return
                on_side(lambda ws: self.adam_range(t0, t1, lr_t, hp["beta1"], hp["beta2"], hp["eps"], hp["clip"]), [], join=True)

        side2 = None
        # (not with a gradient exchange attached: its bucket launch records ONE event on the stream of the bucket's last tensor,
        #  which covers the bucket only if all of its weight gradients ran on that stream)
        if side is not None and self.side_streams == 2 and on_ready is None and on_dgrad is None:
            if getattr(self, "_side2", None) is None:
                self._side2 = torch.cuda.Stream(device=self.device)
                self._ws_side2 = ops.MatchWorkspace()
            side2 = self._side2
        turn = [0]
        red, red_events = None, []
        if side is not None and self.reduce_stream and side2 is None and on_ready is None and on_dgrad is None:
            if getattr(self, "_red", None) is None:
                self._red = torch.cuda.Stream(device=self.device)
                self._ws_side_b = ops.MatchWorkspace()
            red = self._red
            ev0 = torch.cuda.Event()
            ev0.record(main)
            red.wait_event(ev0)                                # (joins the launch sequence / a graph capture here)
            _lib.check(self.L.ssd_set_wgrad_reduce_stream(ctypes.c_void_p(red.cuda_stream)))
        self._red_active = red

        def on_side(fn, tensors, join=False, now=False):
            """Run fn (a weight-gradient launch) after everything enqueued so far on the main stream.
            Reduction stream: the slab sums of the calls go to `red` (the library orders each behind its slab kernel); calls
            alternate between two slab workspaces and call k waits for the sum of call k - 2, whose workspace it reuses.
            join: fn reads gradients (the optimizer): behind every sum enqueued so far.
            Two side streams (SSD_SIDE_STREAMS=2): the launches alternate between them instead."""
            if side is None:
                fn(self._ws)
                if on_ready:
                    on_ready(tensors)
                return
            if join or now or self.wgrad_group <= 1 or side2 is not None or red is not None:
                flush_side()
                run_on_side(fn, tensors, join, None)
                return
            # grouped: the side stream is hundreds of microseconds behind the main stream for most of the backward pass, yet every
            # cross-stream wait costs it ~6 us of idle time (30 of them per step).  `wgrad_group` launches share ONE wait -- on the
            # event of the LAST of them, which a stream that is behind anyway has long passed
            pending.append((fn, tensors))
            if len(pending) >= self.wgrad_group:
                flush_side()

        pending = []

        def flush_side():
            if not pending:
                return
            ev = torch.cuda.Event()
            ev.record(main)
            first = True
            for fn_, tensors_ in pending:
                run_on_side(fn_, tensors_, False, ev if first else False)
                first = False
            pending.clear()

        def run_on_side(fn, tensors, join, ev):
            """ev: None = wait for the main stream as it is now; an event = wait for it; False = no wait (grouped behind one)."""
            if ev is None:
                ev = torch.cuda.Event()
                ev.record(main)
            s_, ws_ = side, self._ws_side
            if side2 is not None and not join:
                turn[0] ^= 1
                if turn[0] == 0:
                    s_, ws_ = side2, self._ws_side2
            with torch.cuda.stream(s_):
                if ev is not False:
                    s_.wait_event(ev)
                if join and side2 is not None:
                    e2 = torch.cuda.Event()
                    e2.record(side2)
                    s_.wait_event(e2)
                if red is not None:
                    if join:
                        if red_events:
                            s_.wait_event(red_events[-1])
                    else:
                        k = len(red_events)
                        if k & 1:
                            ws_ = self._ws_side_b
                        if k >= 2:
                            s_.wait_event(red_events[k - 2])
                fn(ws_)
                if red is not None and not join:
                    e = torch.cuda.Event()
                    e.record(red)
                    red_events.append(e)
                if on_ready:
                    on_ready(tensors)

        # heads.  The two large levels (38x38, 19x19: ~1.1 ms of work) go to the side stream whole, data gradient first, so
        # that the main stream can walk the small levels and the extras' data-gradient chain (a dozen launches that
        # each fill a fraction of the chip) underneath them.  The accumulation order into a feature-map gradient is
        # still "head first, trunk second": the trunk launch waits for the head's event.
        if side is not None and self.chain_start is not None and "bwd" in self.chain and self.chain_prefetch:
            if getattr(self, "_tail", None) is None:
                self._tail = torch.cuda.Stream(device=self.device)
                self._ws_tail = ops.MatchWorkspace()
            ev = torch.cuda.Event()                       # the data-gradient chain's packed filters into L2, ~50 us ahead of it
            ev.record(main)
            with torch.cuda.stream(self._tail):
                self._tail.wait_event(ev)
                ops.chain_prefetch([self.chain_pk_bwd[j] for j in range(self.chain_start, len(self.nodes))])
            prefetch_done = torch.cuda.Event()
            prefetch_done.record(self._tail)
        else:
            prefetch_done = None
        sparse_head_done = {}                              # activation index -> event after a large level's sparse data gradient
        if heads is not None:
            # all levels at once from the compact rows: data gradient on the main stream (every feature-map gradient is
            # written before the trunk chain accumulates into it), weight gradient next to it on the side stream
            hl, keep = self._head_layers(c)
            on_side(lambda ws: ops.heads_bwd_weight_sparse(heads, hl, ws=self._ws_hw),
                    [i for wt, bt in self.head_params for t in (wt, bt) for i in t.indices], now=True)
            big_lv = [lvl for lvl, (ni, h, ch) in enumerate(self.fm) if B * h * h >= 16384]
            small_lv = [lvl for lvl in range(len(self.fm)) if lvl not in big_lv]
            if side is not None and self.split_heads_dgrad and big_lv and small_lv:
                # the small maps' gradients head the extras' chain; the 38x38 / 19x19 maps' (most of the launch's time: their
                # dense maps are ~140 MB of stores) are not read until the chain reaches those maps -- third stream, the
                # chain's accumulation waits for its event (sparse_head_done)
                ops.heads_bwd_data_sparse(heads, hl, ws=self._ws_hz, levels=small_lv)
                if getattr(self, "_tail", None) is None:
                    self._tail = torch.cuda.Stream(device=self.device)
                    self._ws_tail = ops.MatchWorkspace()
                ev = torch.cuda.Event()
                ev.record(main)
                with torch.cuda.stream(self._tail):
                    self._tail.wait_event(ev)
                    pz = all(self.fm[lvl][0] + 1 in self._prezeroed for lvl in big_lv)
                    self._prezeroed = set()                # (consumed: the trunk chain accumulates into the maps from here on)
                    if self.split_heads_dgrad == 2:
                        # one call per level, the level the chain reaches first (19x19) first: its event does not wait for
                        # the 38x38 level's 94 MB of stores
                        for lvl in reversed(big_lv):
                            ops.heads_bwd_data_sparse(heads, hl, ws=self._ws_hz, levels=[lvl], prezeroed=pz)
                            done = torch.cuda.Event()
                            done.record(self._tail)
                            sparse_head_done[self.fm[lvl][0] + 1] = done
                    else:
                        ops.heads_bwd_data_sparse(heads, hl, ws=self._ws_hz, levels=big_lv, prezeroed=pz)
                        done = torch.cuda.Event()
                        done.record(self._tail)
                        for lvl in big_lv:
                            sparse_head_done[self.fm[lvl][0] + 1] = done
            else:
                ops.heads_bwd_data_sparse(heads, hl, ws=self._ws_hz)
            for ni, _, _ in self.fm:
                written[ni + 1] = True
            del keep
        big = [lvl for lvl, (ni, h, ch) in enumerate(self.fm) if heads is None and side is not None and B * h * h >= 16384
               and self.big_heads_side]
        head_done = {}                                     # activation index -> event after the head's data gradient
        packed = [None] * len(self.fm)

        def pack(lvl):                                     # loc + conf gradients of one level in the head's channel order
            h = self.fm[lvl][1]
            packed[lvl] = ops.head_grad_pack(dloc, dconf, h * h, self.num_priors[lvl], self.classes, self.head_npad[lvl],
                                             self.level_off[lvl], out=c["packed"][lvl]).view(B, h, h, self.head_npad[lvl])

        pack_side = self.pack_side
        for lvl in range(len(self.fm) if heads is None else 0):   # the large levels are packed where they are consumed (side stream)
            if lvl not in big or not pack_side:
                pack(lvl)

        def masked_dgrad(key, dy, w_t, a, stride, pt, pl, accumulate, ws):
            """Data gradient w.r.t. activation a (index), masked by its ReLU sign: from the sign bits where this step's forward
            pass wrote them and the kernel reads them (learned per call site), else from the bf16 activation."""
            if self.relu_bits is not None and a in self.bits_valid and self.relu_bits.get(key, True):
                try:
                    ops.conv2d_bwd_data_bits(dy, w_t, c["rbits"][a], acts[a].shape, stride, pt, pl, accumulate=accumulate,
                                             out=gacts[a], ws=ws)
                    self.relu_bits[key] = True
                    return
                except NotImplementedError:
                    self.relu_bits[key] = False
            ops.conv2d_bwd_data(dy, w_t, acts[a], acts[a].shape, stride, pt, pl, accumulate=accumulate, out=gacts[a], ws=ws)

        def head_dgrad(lvl, ws):
            ni = self.fm[lvl][0]
            masked_dgrad("head%d" % lvl, packed[lvl], self.head_w_t[lvl], ni + 1, 1, 1, 1, False, ws)
            written[ni + 1] = True

        def head_wgrad(lvl, ws):
            ni = self.fm[lvl][0]
            wt, bt = self.head_params[lvl]
            ops.conv2d_bwd_weight(acts[ni + 1], packed[lvl], wt.shape[0], 3, 1, 1, 1, dw=self.view(wt, self.grad),
                                  dbias=self.view(bt, self.grad), ws=ws)

        if big:
            ev = torch.cuda.Event()
            ev.record(main)
            with torch.cuda.stream(side):
                side.wait_event(ev)
                for lvl in reversed(big):                  # 19x19 first: the trunk chain reaches it first
                    if pack_side:
                        pack(lvl)
                    head_dgrad(lvl, self._ws_side)
                    done = torch.cuda.Event()
                    done.record(side)
                    head_done[self.fm[lvl][0] + 1] = done
                for lvl in reversed(big):
                    head_wgrad(lvl, self._ws_side)
                    if on_ready:
                        on_ready([i for t in self.head_params[lvl] for i in t.indices])
        for lvl in range(len(self.fm) if heads is None else 0):
            if lvl in big:
                continue
            on_side(lambda ws, lvl=lvl: head_wgrad(lvl, ws), [i for t in self.head_params[lvl] for i in t.indices], now=True)
            head_dgrad(lvl, self._ws)
        opt_bucket(None)
        # trunk, last layer first
        unpooled = set()                          # pooling nodes whose backward pass ran inside the next convolution's data gradient
        first_fused = False
        chained = set()
        if self.chain_start is not None and "bwd" in self.chain:
            # data gradients of nodes end .. chain_start in one launch (ops.conv_chain): the head of the backward pass's
            # critical path -- nothing large can start before this chain reaches the 19x19 map
            last = len(self.nodes) - 1
            assert written[last + 1]
            layers = []
            for j in range(last, self.chain_start - 1, -1):
                ndj = self.nodes[j]
                use_bits = self.relu_bits is not None and j in self.bits_valid
                layers.append(ops.chain_layer_dgrad(self.w_t[j], self.chain_pk_bwd[j], gacts[j], ndj["stride"], ndj["pt"], ndj["pl"], accumulate=written[j],
                                                    mask_bits=c["rbits"][j] if use_bits else None,
                                                    mask_src=None if use_bits else acts[j]))
            try:
                for j in range(last, self.chain_start - 1, -1):
                    for done in (head_done, sparse_head_done):
                        if j in done:
                            main.wait_event(done.pop(j))
                ops.conv_chain(gacts[last + 1], layers)
                chained = set(range(self.chain_start, last + 1))
                for j in chained:
                    written[j] = True
            except NotImplementedError:               # SSD_ERR_UNSUPPORTED: nothing launched
                self.chain = self.chain - {"bwd"}
        batched_w = set()
        if chained and side is not None and self.batch_chain_wgrads and side2 is None and red is None and self.wgrad_probe is None:
            # ... and their weight gradients in two launches (slab kernel + slab sums) instead of twelve
            # (the layer in FRONT of the chain too, where the small-layer kernel serves it: its output gradient is the chain's
            #  last result, so its weight gradient is ready at the same moment)
            extra = [self.chain_start - 1] if (self.batch_chain_front and self.nodes[self.chain_start - 1]["kind"] == "conv") else []
            for front in (extra, []):
                order = sorted(chained, reverse=True) + front
                layers, tens = [], []
                for j in order:
                    ndj = self.nodes[j]
                    wt, bt = self.conv_params[j]
                    layers.append((acts[j], gacts[j + 1], ndj["cout"], ndj["k"], ndj["stride"], ndj["pt"], ndj["pl"],
                                   self.view(wt, self.grad), self.view(bt, self.grad)))
                    tens += [wt.index, bt.index]
                try:
                    on_side(lambda ws: ops.conv2d_bwd_weight_batched(layers, ws=ws), tens, now=True)
                    batched_w = set(order)
                    break
                except NotImplementedError:           # SSD_ERR_UNSUPPORTED: nothing launched
                    if front:
                        self.batch_chain_front = False
                    else:
                        self.batch_chain_wgrads = False
        for i in range(len(self.nodes) - 1, -1, -1):
            nd = self.nodes[i]
            g_out = gacts[i + 1]
            assert written[i + 1]
            if nd["kind"] == "pool":
                assert not written[i], "a pooled activation cannot also feed a head (the pool gradient overwrites)"
                if i not in unpooled:                 # (else the convolution behind the pooling already wrote gacts[i])
                    ops.maxpool2x2_bwd_argmax(c["pool_code"][i], g_out, acts[i].shape, out=gacts[i])
                written[i] = True
                continue
            wt, bt = self.conv_params[i]
            if i == 0 and first_fused:            # its weight gradient came out of the second layer's data-gradient kernel
                on_side(lambda ws: None, [wt.index, bt.index])
                opt_bucket(i)
                continue
            def wgrad(ws, i=i, a=acts[i], go=g_out, nd=nd, wt=wt, bt=bt):
                probe = self.wgrad_probe          # measurement only (bench.py): HIP events around the launches of chosen layers
                timed = probe is not None and i in probe["nodes"]
                if timed:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                ops.conv2d_bwd_weight(a, go, nd["cout"], nd["k"], nd["stride"], nd["pt"], nd["pl"], dw=self.view(wt, self.grad),
                                      dbias=self.view(bt, self.grad), ws=ws)
                if timed:
                    e1.record()
                    probe["events"].append((i, e0, e1))
            if i in batched_w:
                pass                                  # (its weight gradient left with the batched launch above)
            elif side is not None and i in self.wgrad_on_main:
                # the side stream (every weight gradient + the optimizer) ends ~0.45 ms after the main stream: this layer's weight
                # gradient runs on the MAIN stream, in front of its data gradient, and both chains end closer together
                wgrad(self._ws)
                if on_ready:
                    on_ready([wt.index, bt.index])
            else:
                on_side(wgrad, [wt.index, bt.index])
            if i == 0 or i in chained:            # no gradient w.r.t. the image / its data gradient came out of the chain launch
                opt_bucket(i)
                continue
            if (i == 1 and self.fuse_first and not written[1] and self.nodes[0]["kind"] == "conv" and self.nodes[0]["cin"] == 8
                    and self.nodes[0]["cout"] == 64 and nd["cin"] == 64 and nd["cout"] == 64 and nd["k"] == 3 and nd["stride"] == 1
                    and nd["pt"] == 1 and nd["pl"] == 1 and self.nodes[0]["k"] == 3 and self.nodes[0]["stride"] == 1
                    and self.relu_bits is not None and 1 in self.bits_valid):
                wt0, bt0 = self.conv_params[0]
                try:
                    ops.conv2d_bwd_data_wgrad_first(g_out, self.w_t[1], c["rbits"][1], acts[0], dw=self.view(wt0, self.grad),
                                                    dbias=self.view(bt0, self.grad), ws=self._ws)
                    first_fused = True
                    written[1] = True             # (consumed in the kernel: gacts[1] is not written)
                    opt_bucket(i)
                    continue
                except NotImplementedError:       # SSD_ERR_UNSUPPORTED: nothing was launched
                    self.fuse_first = False
            prev_is_relu_conv = self.nodes[i - 1]["kind"] == "conv"
            if i in head_done:                    # a large head wrote gacts[i] on the side stream: accumulate after it
                main.wait_event(head_done.pop(i))
            if i in sparse_head_done:
                main.wait_event(sparse_head_done.pop(i))
            # a 3x3 / stride-1 convolution right behind a pooling: its data gradient is carried through the pooling in the
            # convolution's own store stage (no pooled gradient in HBM, no pooling-backward launch) where an LDS-patch kernel
            # serves the layer; learned at the first call, like pool_only
            fused = False
            if (self.fuse_unpool is not None and self.fuse_unpool.get(i, True) and self.nodes[i - 1]["kind"] == "pool" and not written[i] and not written[i - 1]
                    and nd["k"] == 3 and nd["stride"] == 1 and nd["pt"] == 1 and nd["pl"] == 1):
                try:
                    ops.conv2d_bwd_data_unpool(g_out, self.w_t[i], None, c["pool_code"][i - 1], tuple(gacts[i - 1].shape),
                                               out=gacts[i - 1], ws=self._ws)
                    fused = True
                    unpooled.add(i - 1)
                except NotImplementedError:       # SSD_ERR_UNSUPPORTED: nothing was launched
                    pass
                self.fuse_unpool[i] = fused
            if not fused and prev_is_relu_conv:
                masked_dgrad("conv%d" % i, g_out, self.w_t[i], i, nd["stride"], nd["pt"], nd["pl"], written[i], self._ws)
            elif not fused:
                ops.conv2d_bwd_data(g_out, self.w_t[i], None, acts[i].shape, nd["stride"], nd["pt"], nd["pl"],
                                    accumulate=written[i], out=gacts[i], ws=self._ws)
            written[i] = True
            opt_bucket(i)
        assert not opt_at
        flush_side()
        if prefetch_done is not None:
            main.wait_event(prefetch_done)                # (joins the third stream even where nothing else ran on it)
        for ev in sparse_head_done.values():      # (a large level whose map no trunk node accumulated into)
            main.wait_event(ev)
        for t0, t1, ev, ev2 in deferred:
            main.wait_event(ev)
            if ev2 is not None:
                main.wait_event(ev2)
            self.adam_range(t0, t1, lr_t, hp["beta1"], hp["beta2"], hp["eps"], hp["clip"])
        if side is not None:
            main.wait_stream(side)
            if side2 is not None:
                main.wait_stream(side2)
            if red is not None:
                main.wait_stream(red)
                _lib.check(self.L.ssd_set_wgrad_reduce_stream(None))
                self._red_active = None

    # ---------------------------------------------------------------- optimizer
    def clip_scales(self, clip=0.01):
        """Per-tensor scale = clip / max(||g||, clip)  (tf.clip_by_norm, models/ssd_model.py:249)."""
        _lib.check(self.L.ssd_grad_clip_scales(ops._ptr(self.grad), self.n_flat, ops._ptr(self.tensor_block_off),
                                               len(self.tensors), float(clip), ops._ptr(self.sq_partial),
                                               ops._ptr(self.clip_scale), ops._ptr(self.grad_norms), ops._stream()))

    def apply_clip_in_place(self):
        _lib.check(self.L.ssd_grad_apply_scale(ops._ptr(self.grad), self.n_flat, ops._ptr(self.block_tensor),
                                               ops._ptr(self.clip_scale), ops._stream()))

    def _range_table(self, t0, t1):
        """(first block, tensor->block offsets, block->tensor map), both relative to the range, for tensors t0..t1-1."""
        key = (t0, t1)
        if not hasattr(self, "_range_tables"):
            self._range_tables = {}
        tab = self._range_tables.get(key)
        if tab is None:
            b0 = self.tensors[t0].block0
            tbo = (self.tensor_block_off[t0:t1 + 1] - b0).contiguous()
            bt = (self.block_tensor[b0:int(self.tensor_block_off[t1].item())] - t0).contiguous()
            tab = (b0, tbo, bt)
            self._range_tables[key] = tab
        return tab

    def opt_buckets(self, min_elems=2_000_000):
        """Contiguous tensor ranges for the per-bucket optimizer step, in the order the backward pass completes them:
        [(t0, t1, node)] -- node = lowest trunk node of the range (its data gradient is the last reader of the
        range's transposed weights), None for the heads (all of them, first)."""
        if getattr(self, "_opt_buckets", None) is None:
            first_head = self.head_params[0][0].index
            out = [(first_head, len(self.tensors), None)]
            hi, acc = first_head, 0
            conv_nodes = sorted(self.conv_params)
            for i in reversed(conv_nodes):
                wt, bt = self.conv_params[i]
                acc += wt.numel + bt.numel
                if acc >= min_elems or i == conv_nodes[0]:
                    out.append((wt.index, hi, i))
                    hi, acc = wt.index, 0
            self._opt_buckets = out
        return self._opt_buckets

    def adam_range(self, t0, t1, lr_t, beta1, beta2, eps, clip, grad_scale=1.0):
        """clip_by_norm + Adam + bf16 / transposed copies for parameter tensors t0..t1-1 on the current stream.
        Per tensor the arithmetic is that of clip_scales() + adam() over the whole flat buffer, bit for bit.
        clip=None: the gradient is already clipped (and summed over ranks): only grad_scale (1 / world) applies."""
        b0, tbo, bt = self._range_table(t0, t1)
        start, n = b0 * self.block, bt.numel() * self.block
        sl = slice(start, start + n)
        if clip is not None:
            _lib.check(self.L.ssd_grad_clip_scales(ops._ptr(self.grad[sl]), n, ops._ptr(tbo), t1 - t0, float(clip),
                                                   ops._ptr(self.sq_partial[b0:]), ops._ptr(self.clip_scale[t0:]),
                                                   ops._ptr(self.grad_norms[t0:]), ops._stream()))
        _lib.check(self.L.ssd_adam_step(ops._ptr(self.param[sl]), ops._ptr(self.grad[sl]), ops._ptr(self.adam_m[sl]),
                                        ops._ptr(self.adam_v[sl]), ops._ptr(self.param_bf16[sl]), n, ops._ptr(bt),
                                        ops._ptr(self.clip_scale[t0:]) if clip is not None else None, float(grad_scale),
                                        float(lr_t), float(beta1), float(beta2), float(eps), ops._stream()))
        self.refresh_weights(tensors=(t0, t1))

    def clip_range_in_place(self, t0, t1, clip=0.01):
        """clip_by_norm of tensors t0..t1-1 (a contiguous range of the flat gradient) in place, on the current stream."""
        b0, tbo, bt = self._range_table(t0, t1)
        start = b0 * self.block
        n = bt.numel() * self.block
        g = self.grad[start:start + n]
        _lib.check(self.L.ssd_grad_clip_scales(ops._ptr(g), n, ops._ptr(tbo), t1 - t0, float(clip),
                                               ops._ptr(self.sq_partial[b0:]), ops._ptr(self.clip_scale[t0:]),
                                               ops._ptr(self.grad_norms[t0:]), ops._stream()))
        _lib.check(self.L.ssd_grad_apply_scale(ops._ptr(g), n, ops._ptr(bt), ops._ptr(self.clip_scale[t0:]), ops._stream()))

    def accumulate_clipped(self, first):
        if self.grad_acc is None:
            self.grad_acc = torch.empty_like(self.grad)
        _lib.check(self.L.ssd_grad_accumulate(ops._ptr(self.grad_acc), ops._ptr(self.grad), self.n_flat,
                                              ops._ptr(self.block_tensor), ops._ptr(self.clip_scale), 1 if first else 0,
                                              ops._stream()))

    def adam(self, lr, grad, grad_scale=1.0, use_clip_scale=False, beta1=0.9, beta2=0.999, eps=1e-7):
        self.step_count += 1
        t = self.step_count
        lr_t = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
        _lib.check(self.L.ssd_adam_step(ops._ptr(self.param), ops._ptr(grad), ops._ptr(self.adam_m), ops._ptr(self.adam_v),
                                        ops._ptr(self.param_bf16), self.n_flat, ops._ptr(self.block_tensor),
                                        ops._ptr(self.clip_scale) if use_clip_scale else None, float(grad_scale),
                                        float(lr_t), float(beta1), float(beta2), float(eps), ops._stream()))
        self.refresh_weights()

    def sgd(self, lr, grad, grad_scale=1.0, use_clip_scale=False):
        self.step_count += 1
        _lib.check(self.L.ssd_sgd_step(ops._ptr(self.param), ops._ptr(grad), ops._ptr(self.param_bf16), self.n_flat,
                                       ops._ptr(self.block_tensor), ops._ptr(self.clip_scale) if use_clip_scale else None,
                                       float(grad_scale), float(lr), ops._stream()))
        self.refresh_weights()

    # ---------------------------------------------------------------- state
    LAYOUT_VERSION = 2                         # 2: a level's loc / conf filters are separate variables (64 tensors for SSD300)

    def state_dict(self):
        return dict(param=self.param.cpu(), adam_m=self.adam_m.cpu(), adam_v=self.adam_v.cpu(), step=self.step_count,
                    names=[t.name for t in self.tensors], shapes=[t.shape for t in self.tensors],
                    offsets=[t.offset for t in self.tensors], layout_version=self.LAYOUT_VERSION)

    def load_state_dict(self, sd):
        """The flat buffers are only meaningful together with the layout they were saved under: names, shapes and offsets of
        every variable must match this engine's (another class count, block size or an older fused-head layout would
        otherwise load misaligned weights silently where the sizes happen to coincide)."""
        mine = ([t.name for t in self.tensors], [tuple(t.shape) for t in self.tensors], [t.offset for t in self.tensors])
        theirs = (list(sd.get("names", [])), [tuple(x) for x in sd.get("shapes", [])], list(sd.get("offsets", [])))
        if sd.get("layout_version", 1) != self.LAYOUT_VERSION or mine != theirs or sd["param"].numel() != self.n_flat:
            bad = [n for n in mine[0] if n not in theirs[0]][:3] + [n for n in theirs[0] if n not in mine[0]][:3]
            raise ValueError("checkpoint parameter layout (version %s, %d variables, %d elements) does not match this engine's "
                             "(version %d, %d variables, %d elements)%s" % (
                                 sd.get("layout_version", 1), len(theirs[0]), sd["param"].numel(), self.LAYOUT_VERSION,
                                 len(mine[0]), self.n_flat, "; e.g. " + ", ".join(bad) if bad else ""))
        self.param.copy_(sd["param"])
        self.adam_m.copy_(sd["adam_m"])
        self.adam_v.copy_(sd["adam_v"])
        self.step_count = int(sd["step"])
        self.refresh_weights(cast=True)
