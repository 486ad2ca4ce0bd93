"""CPU suite, world_size 2 over gloo: the data-parallel gradient exchange (parallel.GradReducer) -- bucket plan,
completion-order launching, per-rank clipping before the sum -- against the reference's split_batch semantics
(models/ssd_model.py:240-256 of the reference) evaluated with the numpy oracle."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ssd_oracle as O

BLOCK = 16
SIZES = [40, 7, 300, 16, 64, 5, 129, 33]          # elements per tensor (flat order = forward order)


def layout():
    offs, blocks, off = [], [], 0
    for n in SIZES:
        offs.append(off)
        b = (n + BLOCK - 1) // BLOCK
        blocks.append(b)
        off += b * BLOCK
    return offs, blocks, off


def rank_grads(rank):
    rng = np.random.default_rng(100 + rank)
    offs, blocks, total = layout()
    g = np.zeros(total, np.float32)
    for i, n in enumerate(SIZES):
        g[offs[i]:offs[i] + n] = rng.normal(0, 10.0 ** (-(i % 4)), n) / np.sqrt(n)
    return g


def _worker(rank, world, port, out):
    from ssd_object_detection_amd.parallel import GradReducer, make_buckets, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    offs, blocks, total = layout()
    flat = torch.from_numpy(rank_grads(rank))

    def clip_fn(t0, t1):                                   # numpy stand-in for the HIP clip kernels
        for t in range(t0, t1):
            seg = flat[offs[t]:offs[t] + SIZES[t]]
            seg.copy_(torch.from_numpy(O.clip_by_norm(seg.numpy(), 0.01).astype(np.float32)))

    red = GradReducer(flat, offs, blocks, BLOCK, clip_fn, n_buckets=3)
    order = []
    # backward completes tensors from the back; report them one at a time
    for t in range(len(SIZES) - 1, -1, -1):
        before = red.next_bucket
        red.tensor_ready([t])
        order.append((t, red.next_bucket - before))
    red.finish()
    assert shard_range(64, rank, world) == (rank * 32, rank * 32 + 32)
    if rank == 0:
        out.put((flat.numpy().copy(), red.buckets, order))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_world2():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, buckets, order = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    offs, blocks, total = layout()
    # expected: sum over ranks of per-rank clipped gradients (the division by world is the optimizer's grad_scale)
    want = np.zeros(total, np.float64)
    for r in range(2):
        g = rank_grads(r)
        for i, n in enumerate(SIZES):
            want[offs[i]:offs[i] + n] += O.clip_by_norm(g[offs[i]:offs[i] + n], 0.01)
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-9)
    # bucket plan: contiguous, from the end, covering every tensor exactly once
    flat_ids = [t for (t0, t1) in buckets for t in range(t0, t1)]
    assert sorted(flat_ids) == list(range(len(SIZES))) and buckets[0][1] == len(SIZES)
    assert all(b[0] == nxt[1] for b, nxt in zip(buckets, buckets[1:]))
    # a bucket is launched exactly when its first (lowest) tensor reports ready
    launched_at = [t for t, n in order if n > 0]
    assert launched_at == [b[0] for b in buckets]


# ---------------------------------------------------------------------------------------------------------------
# world size 8 (BASELINE configs[3]: 8 ranks): the same exchange, plus the per-bucket optimizer step gated on the data
# gradients (GradReducer.begin(post_fn, gates) / dgrad_done) and the image shards of a global batch of 512
def _worker8(rank, world, port, out):
    from ssd_object_detection_amd.parallel import GradReducer, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    offs, blocks, total = layout()
    flat = torch.from_numpy(rank_grads(rank))

    def clip_fn(t0, t1):
        for t in range(t0, t1):
            seg = flat[offs[t]:offs[t] + SIZES[t]]
            seg.copy_(torch.from_numpy(O.clip_by_norm(seg.numpy(), 0.01).astype(np.float32)))

    red = GradReducer(flat, offs, blocks, BLOCK, clip_fn, n_buckets=3)
    # tensors 6, 7 play the heads; tensor t < 6 is the kernel of trunk node t: gate of a bucket = its lowest trunk node
    gates = [min([t for t in range(t0, t1) if t < 6], default=None) for t0, t1 in red.buckets]
    log = []
    snap = {}

    def post(t0, t1):                                       # the "optimizer": sees the bucket summed over all ranks
        log.append(("post", t0, t1))
        s, e = red._range(t0, t1)
        snap[(t0, t1)] = flat[s:e].clone()

    red.begin(post, gates)
    red.tensor_ready([6, 7]); log.append(("dgrad", None)); red.dgrad_done(None)
    for node in range(5, -1, -1):                           # weight gradient of a node first, then its data gradient
        red.tensor_ready([node])
        log.append(("dgrad", node))
        red.dgrad_done(node)
    red.finish()
    lo, hi = shard_range(512, rank, world)
    assert (lo, hi) == (rank * 64, rank * 64 + 64)
    if rank == 0:
        out.put((flat.numpy().copy(), red.buckets, gates, log, {k: v.numpy() for k, v in snap.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_and_gated_updates_world8():
    world = 8
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, buckets, gates, log, snap = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    offs, blocks, total = layout()
    want = np.zeros(total, np.float64)
    for r in range(world):
        g = rank_grads(r)
        for i, n in enumerate(SIZES):
            want[offs[i]:offs[i] + n] += O.clip_by_norm(g[offs[i]:offs[i] + n], 0.01)
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-9)
    # every bucket was updated exactly once, in bucket order, with the fully reduced values ...
    posts = [(e[1], e[2]) for e in log if e[0] == "post"]
    assert posts == list(buckets)
    for (t0, t1), v in snap.items():
        s0 = offs[t0]
        np.testing.assert_allclose(v, want[s0:s0 + v.size], rtol=1e-6, atol=1e-9)
    # ... and never before the data gradient of its gate node had been reported (nor, for a heads-only bucket, before the
    # heads' data gradient), but right behind it: not deferred to the end of the backward pass
    for (t0, t1), gate in zip(buckets, gates):
        at = log.index(("post", t0, t1))
        seen = [e[1] for e in log[:at] if e[0] == "dgrad"]
        assert (None in seen) if gate is None else (gate in seen)
        later = [e[1] for e in log[at:] if e[0] == "dgrad"]
        assert all(n is not None and (gate is None or n < gate) for n in later)
