"""Per-step training scalars without per-step host syncs.

The reference writes five TensorBoard scalars after EVERY step (`models/ssd_model.py:281-285`:
`<stage>/loc loss`, `<stage>/cls loss pos`, `<stage>/cls loss neg`, `<stage>/loss`, `<stage>/lr`), each one a
device->host read.  Here a step only enqueues one 32-byte device-to-device copy of the loss kernel's result row into
a ring in HBM; the ring is read back in ONE transfer when it fills, at the end of an epoch, or on close, and the same
five tags are appended to `<log_dir>/scalars.jsonl` (one JSON object per scalar: tag, step, value).  TensorBoard is not
in the image, so the event-file encoding is out of scope; tags, steps and values are the reference's.

Row layout = the loss kernel's `out8`: loc, pos, neg, total, P, N, tau, status (ops.ssd_loss).
"""
import json
import os

import torch

TAGS = ("loc loss", "cls loss pos", "cls loss neg", "loss", "lr")


class HardNegativeThresholdError(AssertionError):
    """status == 2 in a logged row: the hard-negative threshold reached 0 (reference assert, models/ssd_model.py:375)."""


class NoPositivesError(ValueError):
    """status == 1 in a logged row: no positive anchor in the (micro-)batch, or 3P exceeds the anchor count -- the
    reference's tf.math.top_k / division would raise at that step (models/ssd_model.py:359,368)."""


class NonFiniteLossError(FloatingPointError):
    """status == 3 in a logged row: a logit row or a positive's predicted offsets were NaN / Inf -- the run has diverged
    (the reference would go on logging NaN losses; the convolution epilogues here are compiled without NaN semantics, so
    the loss kernel is where a diverged run is caught)."""


class ScalarLog:
    def __init__(self, log_dir, device, capacity=512, distributed=False, console_interval=10, logger=None):
        self.path = os.path.join(log_dir, "scalars.jsonl")
        self.ring = torch.zeros((capacity, 8), dtype=torch.float32, device=device)
        self.meta = []                                   # (stage, step, lr) per filled slot, host side
        self.capacity = capacity
        self.distributed = bool(distributed)
        self.console_interval = max(1, int(console_interval))
        self.logger = logger
        self.rows_written = 0
        self._file = None

    def record(self, stage, step, raw8, lr):
        """Enqueue one step's scalars.  `raw8` = device f32[8] from ops.ssd_loss (stream-ordered copy, no sync)."""
        self.ring[len(self.meta)].copy_(raw8, non_blocking=True)
        self.meta.append((stage, int(step), float(lr)))
        if len(self.meta) == self.capacity:
            self.flush()

    def flush(self, collective=True):
        """One device->host transfer for everything recorded since the last flush; returns the rows written.
        A row whose status word is not 0 raises here -- up to `capacity` steps (or one epoch) after the step itself: the
        price of not reading the device every step; the updates in between have been applied (and a step with status 1
        moved the weights on stale Adam momentum only: its gradient is zero)."""
        n = len(self.meta)
        if n == 0:
            return []
        block = self.ring[:n]
        if collective and self.distributed and torch.distributed.is_initialized():
            # losses: mean over ranks (each rank logged its own image shard); status: the worst of any rank counts
            block = block.clone()
            status = block[:, 7].clone()
            block[:, 7] = 0
            torch.distributed.all_reduce(block)
            torch.distributed.all_reduce(status, op=torch.distributed.ReduceOp.MAX)
            block[:, :7] /= torch.distributed.get_world_size()
            block[:, 7] = status
        host = block.cpu().numpy()                       # the only host sync
        rows = []
        write = (not self.distributed) or (not torch.distributed.is_initialized()) or torch.distributed.get_rank() == 0
        if write and self._file is None:
            os.makedirs(os.path.dirname(self.path) or ".", exist_ok=True)
            self._file = open(self.path, "a")
        for (stage, step, lr), r in zip(self.meta, host):
            loc, pos, neg = float(r[0]), float(r[1]), float(r[2])
            vals = (loc, pos, neg, loc + pos + neg, lr)  # reference :284 sums the three on the host
            rows.append((stage, step, vals))
            if write:
                for tag, v in zip(TAGS, vals):
                    self._file.write(json.dumps({"tag": stage + "/" + tag, "step": step, "value": v}) + "\n")
                if self.logger is not None and step % self.console_interval == 0:
                    self.logger.info("%s step %d: %s", stage, step, dict(zip(TAGS, vals)))
        if write:
            self._file.flush()
        self.rows_written += n
        self.meta.clear()
        bad = [(m, int(r[7])) for m, r in zip(rows, host) if r[7] != 0]
        if bad:
            (stage, step, _), status = bad[0]
            if status == 3:
                raise NonFiniteLossError("non-finite logits / offsets at %s step %d: the run has diverged" % (stage, step))
            if status == 2:
                raise HardNegativeThresholdError("hard-negative threshold reached 0 at %s step %d "
                                                 "(reference assert, models/ssd_model.py:375)" % (stage, step))
            raise NoPositivesError("no positive anchors (or 3P > anchors) at %s step %d: the reference's top_k / division "
                                   "raises there (models/ssd_model.py:359,368)" % (stage, step))
        return rows

    def close(self, collective=True):
        """collective=False while unwinding from an exception: a rank that failed alone must not enter an all-reduce the
        others never reach (its rows are written from local values)."""
        try:
            self.flush(collective=collective)
        finally:
            if self._file is not None:
                self._file.close()
                self._file = None


def read_scalars(path):
    """scalars.jsonl -> {tag: [(step, value), ...]} in file order."""
    out = {}
    with open(path) as f:
        for line in f:
            d = json.loads(line)
            out.setdefault(d["tag"], []).append((d["step"], d["value"]))
    return out
