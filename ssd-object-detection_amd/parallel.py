"""Data-parallel gradient exchange: image-sharded DP with ONE logical all-reduce per step, issued as a few large
buckets that overlap the rest of the backward pass.

Semantics: every rank is one micro-batch of the reference's split_batch loop
(models/ssd_model.py:240-256 of the reference): gradients are clipped per tensor on the rank that produced them
(tf.clip_by_norm, :249), summed over ranks (all-reduce) and divided by the rank count (:256, folded into the
optimizer kernel's grad_scale).

The flat gradient buffer is laid out in forward order and the backward pass completes it from the back (heads,
then conv22 ... conv0), so buckets are contiguous ranges taken from the end of the buffer.  As soon as the last
tensor of a bucket has its weight gradient enqueued, the bucket is clipped in place and all-reduced on a side
stream while the compute stream keeps running the (FLOP-heavy, parameter-light) VGG layers.  xGMI is point to
point (7 links x ~153 GB/s per GPU): few, large messages; the default 6 buckets: ~17-20 MB each for SSD300, and what is
left for the last one -- the only exchange that cannot overlap the backward pass -- is block1..block3 (4.6 MB).
"""
import torch
import torch.distributed as dist


def make_buckets(sizes_blocks, target_blocks):
    """Partition tensors 0..n-1 (sizes in optimizer blocks, flat order) into contiguous buckets filled from the
    END of the buffer.  Returns a list of (first_tensor, last_tensor_exclusive) in completion order."""
    buckets = []
    hi = len(sizes_blocks)
    acc = 0
    end = hi
    for t in range(hi - 1, -1, -1):
        acc += sizes_blocks[t]
        if acc >= target_blocks or t == 0:
            buckets.append((t, end))
            end = t
            acc = 0
    return buckets


class GradReducer:
    """Bucketed, overlapped sum-all-reduce of a flat gradient tensor.

    clip_fn(t0, t1) must clip tensors t0..t1-1 of the flat buffer in place (on the current stream / synchronously on
    CPU); it is how the engine's HIP kernels are plugged in, and how CPU tests plug in a numpy reference."""

    def __init__(self, flat_grad, tensor_offsets, tensor_blocks, block_elems, clip_fn, n_buckets=6, group=None):
        self.flat = flat_grad
        self.offsets = list(tensor_offsets)
        self.blocks = list(tensor_blocks)
        self.block_elems = block_elems
        self.clip_fn = clip_fn
        self.group = group
        total = sum(self.blocks)
        self.buckets = make_buckets(self.blocks, max(1, (total + n_buckets - 1) // n_buckets))
        self.use_streams = flat_grad.is_cuda
        # high priority: the collective's few workgroups must not queue behind compute kernels that fill every CU (the
        # exchange of a bucket is what its optimizer step -- and the end of the step -- waits for)
        self.comm = torch.cuda.Stream(priority=-1) if self.use_streams else None
        self.profile = False                  # True: events around every bucket's exchange (bucket_times_ms)
        self._events = []
        self.begin()

    def begin(self, post_fn=None, gates=None):
        """Start a step.  post_fn(t0, t1) (optional): the optimizer step of a bucket, run on the communication stream right
        behind the bucket's all-reduce.  gates[k]: the bucket may only be updated once the data gradient of trunk node
        gates[k] has been enqueued (it is the last reader of the bucket's transposed weights; None: the heads' data
        gradient); the backward pass reports those through dgrad_done().  Without gates the updates run in finish()."""
        self.ready = [False] * len(self.offsets)
        self.next_bucket = 0
        self.handles = []
        self._events = []
        self.post_fn = post_fn
        self.gates = list(gates) if gates is not None else None
        assert self.gates is None or len(self.gates) == len(self.buckets)
        self.next_post = 0
        self._last_dgrad = "none"              # "none": nothing reported yet; None: heads; int: lowest trunk node so far

    def _mark_done(self, k):
        if self.profile and k < len(self._events):
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record(self.comm)
            self._events[k][1] = e1

    def bucket_times_ms(self):
        """Per bucket of the last step (profile=True): time from 'clipped, all-reduce enqueued' to 'all-reduce complete'
        on the communication stream -- includes waiting behind the previous bucket's exchange."""
        torch.cuda.synchronize()
        return [e0.elapsed_time(e1) for e0, e1 in self._events if e1 is not None]

    def _range(self, t0, t1):
        start = self.offsets[t0]
        end = self.offsets[t1 - 1] + self.blocks[t1 - 1] * self.block_elems
        return start, end

    def tensor_ready(self, idxs):
        for i in idxs:
            self.ready[i] = True
        while self.next_bucket < len(self.buckets):
            t0, t1 = self.buckets[self.next_bucket]
            if not all(self.ready[t0:t1]):
                break
            self._launch(t0, t1)
            self.next_bucket += 1

    def _launch(self, t0, t1):
        start, end = self._range(t0, t1)
        view = self.flat[start:end]
        if self.use_streams:
            ev = torch.cuda.Event()
            ev.record()                                   # the bucket's weight gradients are enqueued
            with torch.cuda.stream(self.comm):
                self.comm.wait_event(ev)
                self.clip_fn(t0, t1)
                if self.profile:
                    e0 = torch.cuda.Event(enable_timing=True)
                    e0.record(self.comm)
                    self._events.append([e0, None])
                self.handles.append(dist.all_reduce(view, group=self.group, async_op=True))
        else:
            self.clip_fn(t0, t1)
            self.handles.append(dist.all_reduce(view, group=self.group, async_op=True))

    def _gate_open(self, gate):
        if self._last_dgrad == "none":
            return False
        if gate is None:
            return True
        return self._last_dgrad is not None and self._last_dgrad <= gate

    def dgrad_done(self, node):
        """The backward pass has just enqueued (on the current stream) the data gradient of trunk node `node` (None: of the
        heads).  Every launched bucket whose gate this opens gets its update enqueued on the communication stream: behind
        its own all-reduce and behind that data gradient -- while the rest of the backward pass is still running."""
        self._last_dgrad = node if (node is None or self._last_dgrad in ("none", None)) else min(self._last_dgrad, node)
        self._post_ready()

    def _post_ready(self):
        if self.post_fn is None or self.gates is None:
            return
        ev = None
        while self.next_post < self.next_bucket and self._gate_open(self.gates[self.next_post]):
            k = self.next_post
            t0, t1 = self.buckets[k]
            if self.use_streams:
                if ev is None:
                    ev = torch.cuda.Event()
                    ev.record()                               # the gating data gradient is enqueued on the caller's stream
                with torch.cuda.stream(self.comm):
                    self.comm.wait_event(ev)
                    self.handles[k].wait()
                    self._mark_done(k)
                    self.post_fn(t0, t1)
            else:
                self.handles[k].wait()
                self.post_fn(t0, t1)
            self.next_post += 1

    def finish(self, post_fn=None):
        """Wait for the exchange.  post_fn(t0, t1), if given, is run for every bucket right behind its all-reduce (on the
        communication stream, once everything the caller has enqueued so far is done): the optimizer step of a bucket
        then overlaps the all-reduce of the next ones instead of waiting for the last."""
        assert self.next_bucket == len(self.buckets), "backward did not report every tensor"
        if self.post_fn is not None and self.gates is not None:
            # per-bucket updates: whatever the gates have not released yet (nothing, when the backward pass reported every
            # data gradient) runs now, behind the whole backward pass
            self._last_dgrad = -1
            self._post_ready()
            assert self.next_post == len(self.buckets)
            if self.use_streams:
                torch.cuda.current_stream().wait_stream(self.comm)
            self.handles = []
            return
        if post_fn is not None and self.use_streams:
            ev = torch.cuda.Event()
            ev.record()                                   # the caller's stream: the whole backward pass is enqueued
            with torch.cuda.stream(self.comm):
                self.comm.wait_event(ev)
                for k, (h, (t0, t1)) in enumerate(zip(self.handles, self.buckets)):
                    h.wait()
                    self._mark_done(k)
                    post_fn(t0, t1)
            torch.cuda.current_stream().wait_stream(self.comm)
            self.handles = []
            return
        for k, (h, (t0, t1)) in enumerate(zip(self.handles, self.buckets)):
            if self.use_streams:
                with torch.cuda.stream(self.comm):
                    h.wait()
                    self._mark_done(k)
            else:
                h.wait()
            if post_fn is not None:
                post_fn(t0, t1)
        if self.use_streams:
            torch.cuda.current_stream().wait_stream(self.comm)
        self.handles = []


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of n_items for `rank` (image sharding of a global batch)."""
    per = n_items // world
    return rank * per, (rank + 1) * per
