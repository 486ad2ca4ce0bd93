"""SSDObjectDetectionModel with the reference's surface (models/ssd_model.py of the reference: constructor
:50-52, TrainConfig :20-40, train :326, get_train_set :209, _train_step :229, _ssd_loss :341, get_prior_box
:416, save/load :405-411, get_log_dir :419), executed on MI355X by the HIP engine.

What differs from the reference, by design:
  * get_train_set batches samples first and runs target assignment for the whole batch on the device.
  * _ssd_loss / _train_step return device tensors; nothing synchronises the host unless the caller reads a
    scalar (the reference forces >= 4 device->host copies per micro-batch, :389-394).
  * `distributed=True` shards the batch by image over torch.distributed ranks: every rank is one micro-batch
    (loss, mining and per-tensor clipping are per rank, as the reference does per micro-batch :240-256), the
    clipped gradients are summed with one RCCL all-reduce and divided by the rank count.
  * inference adds a real NMS pass (`detect`), which the reference lacks.
"""
import logging
import math
import os
import time

import numpy as np
import torch

from .. import ops
from ..engine import SSDEngine
from ..parallel import GradReducer, shard_range
from .. import optimizers as _opt
from ..utils.scalar_log import ScalarLog

logger = logging.getLogger(__name__)
_RESTORED = object()                            # _slot_owner marker: Adam moments were loaded from a checkpoint


class SSDObjectDetectionModel:
    class TrainConfig:
        def __init__(self, epoch, batch_size, optimizer, warmup=True, warmup_optimizer=None, warmup_step=1000,
                     visualization_log_interval=10, split_batch=False, split_batch_size=4, start_epoch=0):
            if warmup_optimizer is None:
                warmup_optimizer = _opt.Adam(_opt.PolynomialDecay(1e-6, 1000, 0.001))
            self.epoch = epoch
            self.batch_size = batch_size
            self.optimizer = optimizer
            self.warmup = warmup
            self.warmup_optimizer = warmup_optimizer
            self.warmup_step = warmup_step
            self.visualization_log_interval = visualization_log_interval
            self.split_batch = split_batch
            self.split_batch_size = split_batch_size
            self.start_epoch = start_epoch                 # > 0: resumed run (no warm-up, epochs start_epoch..epoch-1)

    class Config:
        def __init__(self, classes, log_dir):
            self.log_dir = log_dir
            self.input_shape = (300, 300, 3)
            self.classes = classes + 1               # background is the LAST class index
            self.thresh = 0.5

    def __init__(self, classes, log_dir, device="cuda", seed=0, distributed=False, timestamp_dir=True):
        self.distributed = bool(distributed)
        if timestamp_dir:
            stamp = [time.strftime("%Y-%m-%d-%H%M%S", time.localtime())]
            if self.distributed and torch.distributed.is_initialized():
                torch.distributed.broadcast_object_list(stamp, src=0)     # one run directory for all ranks (rank 0's clock)
            log_dir = os.path.join(log_dir, stamp[0])
        self.cfg = SSDObjectDetectionModel.Config(classes, log_dir)
        self.device = torch.device(device)
        self._engine = SSDEngine(classes=self.cfg.classes, in_size=self.cfg.input_shape[0], device=device, seed=seed)
        self._pset = ops.build_priors(grids=self._engine.grids, device=device)
        assert self._pset.A == self._engine.A
        self._prior_box = None
        self._comm_stream = None
        self._slot_owner = None                      # optimizer object whose Adam moments the engine holds
        self._reducer = None
        self.last_info = None
        self._targets_event = None
        self._match_ws = ops.MatchWorkspace()
        self.fused_optimizer = os.environ.get("SSD_FUSED_OPTIMIZER", "1") == "1"   # Adam per bucket inside backward
        # the step's main chain (forward, loss, data gradients) runs on a HIGH-priority stream: its kernels are the critical
        # path, the weight gradients / optimizer on the engine's side stream fill in around them (measured: -0.04 ms per step;
        # the other way round, side stream high, +0.13 ms)
        self.high_priority_main = os.environ.get("SSD_MAIN_PRIO", "-1") == "-1"
        self._main_hi = None

    # ------------------------------------------------------------------ accessors
    def get_prior_box(self):
        """float64 [8732, 4] numpy array, as the reference returns."""
        if self._prior_box is None:
            self._prior_box = self._pset.priors.cpu().numpy()
        return self._prior_box

    def get_log_dir(self):
        return self.cfg.log_dir

    def get_engine(self):
        return self._engine

    # ------------------------------------------------------------------ input pipeline (A8)
    def _rank_world(self):
        if self.distributed and torch.distributed.is_initialized():
            return torch.distributed.get_rank(), torch.distributed.get_world_size()
        return 0, 1

    def get_train_set(self, dataset, batch_size=1, shard=None):
        """Iterable of (image f32[B,300,300,3] in [-1,1], (cls i32[B,A], loc f32[B,A,4], mask u8[B,A])) device
        batches; remainder dropped (reference :209-227: match_bbox -> apply_anchor_box -> (x-0.5)*2 -> batch).

        Data parallel (`distributed=True`, or an explicit shard=(rank, world)): batch_size is the GLOBAL batch and the
        iterable yields this rank's images of every global batch -- parallel.shard_range, positions
        [rank*b/world, (rank+1)*b/world) -- so all ranks see the same number of batches (the same remainder is dropped)
        and disjoint samples.  Samples of other ranks are skipped without being produced when the dataset offers
        lazy() (an iterable of zero-argument callables, one per sample, in iteration order)."""
        model = self
        rank, world = shard if shard is not None else self._rank_world()
        assert batch_size % world == 0, "the global batch must divide evenly over the ranks"
        lo, hi = shard_range(batch_size, rank, world)
        raw = bool(getattr(dataset, "raw", False))     # reader-contract samples: /255, resize, box normalisation on the device

        class _Batches:
            def __iter__(self_inner):
                thunks = dataset.lazy() if hasattr(dataset, "lazy") else ((lambda s=s: s) for s in dataset)
                mine, pos = [], 0
                for thunk in thunks:
                    if lo <= pos < hi:
                        mine.append(thunk)
                    pos += 1
                    if pos == batch_size:
                        imgs, clss, boxes = [], [], []
                        for t in mine:
                            image, cls, box = t()
                            imgs.append(np.asarray(image) if raw else np.asarray(image, np.float32))
                            clss.append(np.asarray(cls, np.float32))
                            boxes.append(np.asarray(box, np.float32))
                        yield model.make_batch_raw(imgs, clss, boxes) if raw else model.make_batch(imgs, clss, boxes)
                        mine, pos = [], 0

        return _Batches()

    def match_async(self, gt, out=None):
        """Target assignment (A3-A5) for a packed ground-truth batch on the engine's side stream: it depends on nothing
        the network computes, so it runs underneath the forward pass; _train_step waits for it before the loss.
        gt = ops.pack_gt(...) tuple; out = optional (cls, loc, mask) buffers to reuse."""
        B, A, dev = gt[2].numel() - 1, self._pset.A, self.device
        if out is None:
            out = (torch.empty((B, A), dtype=torch.int32, device=dev),
                   torch.empty((B, A, 4), dtype=torch.float32, device=dev),
                   torch.empty((B, A), dtype=torch.uint8, device=dev))
        side = self._engine._side_stream() if self._engine.overlap_heads else None
        if side is None:
            return ops.match_encode(*gt, self._pset, self.cfg.thresh, out=out, ws=self._match_ws)
        start = torch.cuda.Event()
        start.record()                                 # everything that still reads `out` / gt is ahead of this point
        with torch.cuda.stream(side):
            side.wait_event(start)
            ops.match_encode(*gt, self._pset, self.cfg.thresh, out=out, ws=self._match_ws)
            self._targets_event = torch.cuda.Event()
            self._targets_event.record(side)
        return out

    def make_batch(self, images, cls_list, box_list):
        img = torch.from_numpy(np.stack(images, 0)).to(self.device, non_blocking=True)
        img = (img - 0.5) * 2                          # reference :214 (exact in fp32); get_train_set's contract is f32
        gt = ops.pack_gt(box_list, cls_list, device=self.device)
        cls, loc, mask = ops.match_encode(*gt, self._pset, self.cfg.thresh)
        return img, (cls, loc, mask)

    def make_batch_raw(self, images_u8, cls_list, box_tlwh_list):
        """The same from what the COCO reader yields before any preprocessing: decoded uint8 RGB images
        of arbitrary sizes and COCO [x, y, w, h] pixel boxes.  '/255', cv2.resize to the network size, '(x-0.5)*2'
        (reference data_loaders/coco/make_dataset.py:117, data_loaders/ssd/make_dataset.py:40-44, models/ssd_model.py:214)
        and the box conversion run on the device; returns the prepared bf16 network input instead of the f32 image."""
        hw = np.array([im.shape[:2] for im in images_u8], np.int32)
        sizes = [int(im.size) for im in images_u8]
        off = np.zeros(len(sizes), np.int64)
        off[1:] = np.cumsum(sizes[:-1])
        flat = np.concatenate([np.ascontiguousarray(im, np.uint8).reshape(-1) for im in images_u8])
        dev = self.device
        hw_d = torch.from_numpy(hw).to(dev)
        x = ops.image_resize_prep(torch.from_numpy(flat).to(dev), torch.from_numpy(off).to(dev), hw_d,
                                  int(self.cfg.input_shape[0]), True)
        gt_box, gt_cls, gt_off, total, max_nt = ops.pack_gt(box_tlwh_list, cls_list, device=dev)
        gt_box = ops.box_prep(gt_box, gt_off, hw_d) if total else gt_box
        cls, loc, mask = ops.match_encode(gt_box, gt_cls, gt_off, total, max_nt, self._pset, self.cfg.thresh)
        return x, (cls, loc, mask)

    # ------------------------------------------------------------------ loss (A6)
    @staticmethod
    def _ssd_loss(y_true, y_pred, heads=None):
        """Returns (total loss tensor, info) where info maps the reference's three names to device scalars and
        carries the gradients w.r.t. (pred_box, pred_cls) under 'dloc' / 'dconf' -- or, with `heads` (the engine's
        ops.HeadGradBuffers), as the compact per-level rows the heads' backward pass consumes ('heads')."""
        gt_cls, gt_box, gt_mask = y_true
        pred_box, pred_cls = y_pred
        assert gt_cls.shape[0] == gt_box.shape[0] == gt_mask.shape[0] == pred_box.shape[0] == pred_cls.shape[0]
        if heads is not None:
            out = ops.ssd_loss_heads(pred_cls, pred_box, gt_cls, gt_box, gt_mask, heads)
            dconf = dloc = None
        else:
            out, dconf, dloc = ops.ssd_loss(pred_cls, pred_box, gt_cls, gt_box, gt_mask)
        info = {"cls loss pos": out[1], "cls loss neg": out[2], "loc loss": out[0], "status": out[7],
                "dconf": dconf, "dloc": dloc, "heads": heads, "raw": out}
        return out[3], info

    # ------------------------------------------------------------------ train step (A7)
    def main_stream(self):
        """The high-priority stream the step's main chain runs on.  A training loop that issues its own per-step work (input
        preparation, target assignment, logging reads) inside `with torch.cuda.stream(model.main_stream())` saves the two
        cross-stream hand-overs per step (~15-25 us each) that `_train_step` otherwise makes from and back to the caller's
        stream.  None where the step runs on the caller's stream anyway."""
        if not self.high_priority_main or not torch.cuda.is_available():
            return None
        if self._main_hi is None:
            self._main_hi = torch.cuda.Stream(priority=-1)
        return self._main_hi

    def _train_step(self, *args, **kwargs):
        if not self.high_priority_main or not torch.cuda.is_available():
            return self._train_step_on_current(*args, **kwargs)
        if self._main_hi is None:
            self._main_hi = torch.cuda.Stream(priority=-1)
        cur = torch.cuda.current_stream()
        if cur == self._main_hi:                       # the caller already works on it (main_stream())
            return self._train_step_on_current(*args, **kwargs)
        self._main_hi.wait_stream(cur)
        with torch.cuda.stream(self._main_hi):
            out = self._train_step_on_current(*args, **kwargs)
        cur.wait_stream(self._main_hi)             # the caller's stream sees the step complete, as before
        return out

    def _train_step_on_current(self, image, gt_cls, gt_bbox, gt_mask, ssd_optimizer, stage="train", set_names=None,
                               set_colors=None, step=0, cfg=None):
        eng = self._engine
        batch_size = image.shape[0]
        batch_step = batch_size if (cfg is None or not cfg.split_batch) else cfg.split_batch_size
        n_micro = 0
        info = None
        world = torch.distributed.get_world_size() if self.distributed else 1
        single = world == 1 and batch_step >= batch_size
        overlap = world > 1 and batch_step >= batch_size           # one micro-batch per rank: bucketed, overlapped reduce
        if overlap and self._reducer is None:
            blocks = [t.nblocks for t in eng.tensors]
            self._reducer = GradReducer(eng.grad, [t.block0 * eng.block for t in eng.tensors], blocks, eng.block,
                                        eng.clip_range_in_place)
        fused = single and isinstance(ssd_optimizer, _opt.Adam) and self.fused_optimizer
        fused_dp = overlap and isinstance(ssd_optimizer, _opt.Adam) and self.fused_optimizer
        if fused_dp:
            self._adopt_slots(ssd_optimizer)
        if fused:                                      # the optimizer runs per bucket inside the backward pass
            self._adopt_slots(ssd_optimizer)
            eng.step_count = ssd_optimizer.iterations
            lr = ssd_optimizer.lr()
            fused_adam = dict(lr=lr, beta1=ssd_optimizer.beta_1, beta2=ssd_optimizer.beta_2,
                              eps=ssd_optimizer.epsilon, clip=0.01)          # tf.clip_by_norm(x, 0.01), reference :249
        for i in range(0, batch_size, batch_step):
            if image.dtype == torch.bfloat16:          # already prepared on the device (make_batch_raw)
                x = image[i:i + batch_step]
            else:
                x = ops.image_prep(image[i:i + batch_step].contiguous(), normalize=False)
            pred_loc, pred_conf = eng.forward(x)
            if self._targets_event is not None:        # targets assigned on the side stream (match_async)
                torch.cuda.current_stream().wait_event(self._targets_event)
                self._targets_event = None
            heads = eng.head_grad_buffers(pred_loc.shape[0]) if pred_conf.dtype == torch.bfloat16 else None
            _, info = self._ssd_loss((gt_cls[i:i + batch_step], gt_bbox[i:i + batch_step], gt_mask[i:i + batch_step]),
                                     (pred_loc, pred_conf), heads)
            if overlap:
                post = gates = None
                if fused_dp:
                    # Adam of a bucket on the communication stream right behind its all-reduce, as soon as the last data
                    # gradient that reads the bucket's transposed weights is enqueued (engine.bucket_gates): the update of
                    # all but the last bucket runs underneath the rest of the backward pass
                    eng.step_count = ssd_optimizer.iterations + 1
                    t = eng.step_count
                    lr = ssd_optimizer.lr()
                    b1, b2 = ssd_optimizer.beta_1, ssd_optimizer.beta_2
                    lr_t = lr * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
                    post = lambda t0, t1: eng.adam_range(t0, t1, lr_t, b1, b2, ssd_optimizer.epsilon, None, 1.0 / world)
                    gates = eng.bucket_gates(self._reducer.buckets)
                self._reducer.begin(post, gates)
                eng.backward(info["dloc"], info["dconf"], on_ready=self._reducer.tensor_ready, heads=info["heads"],
                             on_dgrad=self._reducer.dgrad_done if fused_dp else None)
                self._reducer.finish()                 # clipped per bucket, summed over ranks (RCCL over xGMI)
            elif fused:
                eng.backward(info["dloc"], info["dconf"], fused_adam=fused_adam, heads=info["heads"])
            else:
                eng.backward(info["dloc"], info["dconf"], heads=info["heads"])
                eng.clip_scales(0.01)                  # tf.clip_by_norm(x, 0.01) per tensor, reference :249
                if not single:
                    eng.accumulate_clipped(first=(n_micro == 0))
            n_micro += 1
        if fused or fused_dp:
            ssd_optimizer.iterations += 1
            return self._finish_step(info, lr, pred_conf, pred_loc)
        self._adopt_slots(ssd_optimizer)
        lr = ssd_optimizer.lr()
        if single:
            grad, gscale, use_clip = eng.grad, 1.0, True
        elif overlap:
            grad, gscale, use_clip = eng.grad, 1.0 / world, False
        else:
            grad, use_clip = eng.grad_acc, False
            if world > 1:
                torch.distributed.all_reduce(grad)     # RCCL over xGMI: sum of the ranks' clipped gradients
            gscale = 1.0 / (n_micro * world)           # reference :256
        if isinstance(ssd_optimizer, _opt.SGD):
            eng.sgd(lr, grad, gscale, use_clip)
        else:
            eng.step_count = ssd_optimizer.iterations
            eng.adam(lr, grad, gscale, use_clip, ssd_optimizer.beta_1, ssd_optimizer.beta_2, ssd_optimizer.epsilon)
        ssd_optimizer.iterations += 1
        return self._finish_step(info, lr, pred_conf, pred_loc)

    def _adopt_slots(self, ssd_optimizer):
        eng = self._engine
        if self._slot_owner is _RESTORED:             # moments came from a checkpoint: the first optimizer adopts them
            self._slot_owner = ssd_optimizer
        elif self._slot_owner is not ssd_optimizer:   # Keras keeps separate slots per optimizer (warm-up vs train)
            eng.adam_m.zero_()
            eng.adam_v.zero_()
            self._slot_owner = ssd_optimizer

    def _finish_step(self, info, lr, pred_conf, pred_loc):
        self._last_raw = info["raw"]                  # device f32[8] row for the sync-free scalar log (N4)
        info = {k: v for k, v in info.items() if k in ("cls loss pos", "cls loss neg", "loc loss", "status")}
        info["lr"] = lr
        self.last_info = info
        return pred_conf, pred_loc, info

    # ------------------------------------------------------------------ trainer shell
    def _train(self, data_loader, cfg):
        train_set, _val_set = data_loader.get_dataset()
        set_names, set_colors = data_loader.get_names_and_colors()
        batches = self.get_train_set(train_set, batch_size=cfg.batch_size)
        self._assert_replicas_identical()
        self._scalars = ScalarLog(self.cfg.log_dir, self.device, distributed=self.distributed,
                                  console_interval=cfg.visualization_log_interval, logger=logger)
        try:
            self._train_loop(batches, set_names, set_colors, cfg)
        except BaseException:
            self._scalars.close(collective=False)      # unwinding: the other ranks may not reach a collective
            raise
        self._scalars.close()

    def _assert_replicas_identical(self):
        """Data parallelism assumes every rank starts from the same weights and optimizer state (same seed / same
        checkpoint): compare a checksum of the flat buffers across ranks before the first step."""
        rank, world = self._rank_world()
        if world == 1:
            return
        eng = self._engine
        sums = torch.stack([eng.param.double().sum(), eng.param.double().abs().sum(), eng.adam_m.double().abs().sum(),
                            eng.adam_v.double().sum(), torch.tensor(float(eng.step_count), dtype=torch.float64,
                                                                    device=eng.param.device)])
        lo, hi = sums.clone(), sums.clone()
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        if not torch.equal(lo, hi):
            raise RuntimeError("data-parallel replicas differ before training (checksums %s vs %s): same seed / "
                               "checkpoint on every rank?" % (lo.tolist(), hi.tolist()))

    def _train_loop(self, batches, set_names, set_colors, cfg):
        ms = self.main_stream()
        if ms is None or torch.cuda.current_stream() == ms:
            return self._train_loop_on_current(batches, set_names, set_colors, cfg)
        cur = torch.cuda.current_stream()              # the whole loop (batch assembly, steps, scalar ring) on the step's stream
        ms.wait_stream(cur)
        try:
            with torch.cuda.stream(ms):
                return self._train_loop_on_current(batches, set_names, set_colors, cfg)
        finally:
            cur.wait_stream(ms)

    def _train_loop_on_current(self, batches, set_names, set_colors, cfg):
        if cfg.warmup and getattr(cfg, "start_epoch", 0) == 0:
            logger.info("Warm up for %s steps", cfg.warmup_step)
            step = 0
            while step < cfg.warmup_step:
                got = False
                for image, (gt_cls, gt_bbox, gt_mask) in batches:
                    got = True
                    step += 1
                    _, _, info = self._train_step(image, gt_cls, gt_bbox, gt_mask, cfg.warmup_optimizer, "warmup",
                                                  set_names, set_colors, step, cfg)
                    self._log(step, info, cfg, "warmup")
                    if step >= cfg.warmup_step:
                        break
                if not got:
                    break
        step = cfg.optimizer.iterations if getattr(cfg, "start_epoch", 0) else 0     # resumed runs keep counting
        for epoch in range(getattr(cfg, "start_epoch", 0), cfg.epoch):
            logger.info("Epoch %s/%s", epoch + 1, cfg.epoch)
            for image, (gt_cls, gt_bbox, gt_mask) in batches:
                step += 1
                _, _, info = self._train_step(image, gt_cls, gt_bbox, gt_mask, cfg.optimizer, "train", set_names,
                                              set_colors, step, cfg)
                self._log(step, info, cfg, "train")
            self._scalars.flush()                      # one device->host read per epoch (or per full ring)
            self.save(os.path.join(self.cfg.log_dir, "model_weight", "model_weight_epoch_%d.pt" % epoch),
                      extra=dict(epoch=epoch + 1, iterations=cfg.optimizer.iterations,
                                 warmup_iterations=cfg.warmup_optimizer.iterations if cfg.warmup_optimizer else 0))

    def _log(self, step, info, cfg, stage):
        """Reference :281-285 writes five scalars per step, each a host read; here a step enqueues one 32-byte device
        copy and utils/scalar_log.py reads the ring back once per epoch / 512 steps (the reference's status assert,
        :375, fires at that read)."""
        self._scalars.record(stage, step, self._last_raw, info["lr"])

    def train(self, data_loader, cfg):
        if cfg.warmup is True:
            assert cfg.warmup_optimizer is not None, "Define a warmup optimizer if you want to enable warmup!"
        try:
            self._train(data_loader, cfg)
        except Exception:
            self.save("error_exit_save.pt", collective=False)
            logger.critical("Error occurred while training, last model weight is saved to 'error_exit_save.pt'")
            raise

    # ------------------------------------------------------------------ inference (A9 + A9')
    def detect(self, image, score_thresh=0.3, iou_thresh=0.45, max_cand=400):
        """image f32 [B,300,300,3] in [-1,1] -> (score, cls, box_px, keep) device tensors [B,A(,4)].
        Scoring/decoding as the reference's visualize(); `keep` adds per-class NMS (no reference counterpart)."""
        x = ops.image_prep(image.contiguous(), normalize=False)
        loc, conf = self._engine.forward(x)
        score, cls, box, cand = ops.score_decode(conf, loc, self._pset, score_thresh, float(self.cfg.input_shape[0]))
        keep = ops.nms(score, cls, box, cand, iou_thresh, max_cand)
        return score, cls, box, keep

    def evaluate(self, samples, batch_size=32, score_thresh=0.05, iou_thresh=0.45, max_dets=100, return_detections=False):
        """Evaluation pass (the reference fetches its val split at models/ssd_model.py:291 and drops it):
        samples = iterable of (image f32 [S,S,3] in [0,1], cls [n], box [n,4] relative cx,cy,w,h) as the loaders yield
        them.  Network forward, scoring/decoding and per-class NMS run on the device; the kept detections go to
        utils.metrics.coco_map on the host.  Returns its dict (mAP = AP@[.5:.95], AP50, AP75, per_class); with
        return_detections also the per-image (score, cls, box_px) arrays that were scored."""
        from ..utils.metrics import coco_map
        size = float(self.cfg.input_shape[0])
        dets, gts, buf = [], [], []

        def flush():
            if not buf:
                return
            img = torch.from_numpy(np.stack([b[0] for b in buf], 0)).to(self.device)
            score, cls, box, keep = self.detect((img - 0.5) * 2, score_thresh, iou_thresh)
            score, cls, box, keep = score.cpu().numpy(), cls.cpu().numpy(), box.cpu().numpy(), keep.cpu().numpy().astype(bool)
            for i, (_, gcls, gbox) in enumerate(buf):
                k = keep[i]
                dets.append((score[i][k], cls[i][k], box[i][k]))
                gts.append((np.asarray(gcls), np.asarray(gbox, np.float64) * size))     # pixels, like the decoded boxes
            buf.clear()

        for sample in samples:
            buf.append(sample)
            if len(buf) == batch_size:
                flush()
        flush()
        result = coco_map(dets, gts, max_dets=max_dets)
        return (result, dets) if return_detections else result

    # ------------------------------------------------------------------ checkpoint
    def save(self, path="model_weight.pt", extra=None, collective=True):
        """Weights + Adam moments + step count (the reference's Keras .h5 has weights only, models/ssd_model.py:405-407);
        `extra` (e.g. optimizer iteration counters, epoch) is stored alongside for resume.
        Data parallel: the replicas are identical, rank 0 writes and the others wait for the file (collective=False on
        error paths, where the other ranks may never arrive)."""
        rank, world = self._rank_world()
        if rank == 0:
            d = os.path.dirname(path)
            if d:
                os.makedirs(d, exist_ok=True)
            sd = self._engine.state_dict()
            if extra:
                sd["extra"] = dict(extra)
            torch.save(sd, path)
            logger.info("Model is saved to %s", path)
        if world > 1 and collective:
            torch.distributed.barrier()

    def load(self, path="model_weight.pt"):
        sd = torch.load(path, map_location="cpu", weights_only=True)     # tensors, lists, ints and strings only
        self._engine.load_state_dict(sd)
        self._slot_owner = _RESTORED
        logger.info("Model is loaded from %s", path)
        return sd.get("extra", {})
