"""GPU: data-parallel train step.  Two ranks (gloo backend, both on the one visible GPU) each run their image shard
as one micro-batch with the bucketed, overlapped gradient exchange; the result must equal ONE process running the
whole batch with the reference's split_batch semantics (models/ssd_model.py:240-256: loss + per-tensor clip per
micro-batch, then the mean) -- SURVEY.md section 8(e)."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

PER_RANK = 2
WORLD = 2


def _make_inputs(model):
    from ssd_object_detection_amd.data_loaders.synthetic import synth_batch_gt, synth_image
    n = PER_RANK * WORLD
    cls_l, box_l = synth_batch_gt(300, n)
    imgs = [synth_image(300 + i) for i in range(n)]
    return imgs, cls_l, box_l


def _rank_main(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ssd_object_detection_amd import optimizers
    from ssd_object_detection_amd.models import SSDObjectDetectionModel
    model = SSDObjectDetectionModel(classes=80, log_dir="gpurun_out/dp_test", seed=5, distributed=True, timestamp_dir=False)
    imgs, cls_l, box_l = _make_inputs(model)
    lo, hi = rank * PER_RANK, (rank + 1) * PER_RANK
    image, (cls, loc, mask) = model.make_batch(imgs[lo:hi], cls_l[lo:hi], box_l[lo:hi])
    opt = optimizers.Adam(1e-3)
    model._train_step(image, cls, loc, mask, opt)
    torch.cuda.synchronize()
    if rank == 0:
        q.put((model.get_engine().grad.cpu().numpy(), model.get_engine().param.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_split_batch():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, WORLD, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    dp_grad, dp_param = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0

    from ssd_object_detection_amd import optimizers
    from ssd_object_detection_amd.models import SSDObjectDetectionModel
    model = SSDObjectDetectionModel(classes=80, log_dir="gpurun_out/dp_test", seed=5, timestamp_dir=False)
    p0 = model.get_engine().param.cpu().numpy().copy()
    imgs, cls_l, box_l = _make_inputs(model)
    image, (cls, loc, mask) = model.make_batch(imgs, cls_l, box_l)
    cfg = SSDObjectDetectionModel.TrainConfig(epoch=1, batch_size=PER_RANK * WORLD, optimizer=None, warmup=False,
                                              split_batch=True, split_batch_size=PER_RANK)
    opt = optimizers.Adam(1e-3)
    model._train_step(image, cls, loc, mask, opt, cfg=cfg)
    ref_grad = model.get_engine().grad_acc.cpu().numpy()        # sum over micro-batches of the clipped gradients
    ref = model.get_engine().param.cpu().numpy()
    moved = np.abs(ref - p0).max()
    assert moved > 1e-4                                         # the step did something
    # the exchanged quantity: sum over ranks of per-rank clipped gradients == sum over micro-batches (identical
    # mathematics; only the grouping of the final additions differs: fma in the accumulate kernel, add in the all-reduce)
    scale = np.abs(ref_grad).max()
    assert scale > 0
    assert np.abs(dp_grad - ref_grad).max() <= 1e-6 * scale
    # parameters after Adam: the bulk agrees to fp32 rounding; Adam divides by sqrt(v) + 1e-7, so last-bit differences
    # of near-zero gradients are amplified on a few elements
    diff = np.abs(dp_param - ref)
    assert float(diff.mean()) < 1e-7 and float((diff > 1e-6).mean()) < 0.02 and diff.max() <= 0.5 * moved


def _train_cfg(log_dir, split):
    from ssd_object_detection_amd.tools import train as T
    cfg = T.load_config(os.path.join(os.path.dirname(T.__file__), "..", "config", "default.yml"))
    cfg["data"]["mini_batch"]["num_data"] = 2 * PER_RANK * WORLD + 1          # two global batches + a dropped remainder
    cfg["data"]["shuffle"] = True
    cfg["model"]["log_dir"] = log_dir
    cfg["model"]["train"]["batch_size"] = PER_RANK * WORLD                     # the GLOBAL batch
    cfg["model"]["train"]["epoch"] = 1
    cfg["model"]["warmup"]["enable"] = False
    cfg["model"]["split_train"]["enable"] = split
    cfg["model"]["split_train"]["batch_size"] = PER_RANK
    cfg["model"]["log_interval"] = 100
    return cfg


def _train_rank_main(rank, world, port, log_dir, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", SSD_DIST_BACKEND="gloo")
    import torch.distributed as dist
    from ssd_object_detection_amd.tools import train as T
    model = T.train(_train_cfg(log_dir, split=False))
    torch.cuda.synchronize()
    q.put((rank, model.get_log_dir(), model.get_engine().param.cpu().numpy() if rank == 0 else None,
           model.get_engine().step_count))
    dist.barrier()
    dist.destroy_process_group()


def test_train_cli_two_ranks_equal_split_batch(tmp_path):
    """The product entry point (tools/train.py -> SSDObjectDetectionModel.train) under two ranks: each rank trains on its
    image shard of every global batch, one log directory, rank 0 alone writes config.json / checkpoints / scalars, and the
    weights after the epoch equal ONE process running the same global batches with split_batch (reference
    models/ssd_model.py:240-256)."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    dp_dir = str(tmp_path / "dp")
    procs = [ctx.Process(target=_train_rank_main, args=(r, WORLD, port, dp_dir, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(WORLD):
        r, log_dir, param, steps = q.get(timeout=600)
        res[r] = (log_dir, param, steps)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0][0] == res[1][0] and res[0][2] == res[1][2] == 2              # one run directory, two optimizer steps
    run_dir = res[0][0]
    assert sorted(os.listdir(dp_dir)) == [os.path.basename(run_dir)]           # no second timestamp directory
    for f in ("config.json", "model_last.pt", "scalars.jsonl", os.path.join("model_weight", "model_weight_epoch_0.pt")):
        assert os.path.exists(os.path.join(run_dir, f)), f
    from ssd_object_detection_amd.utils.scalar_log import read_scalars
    assert [s for s, _ in read_scalars(os.path.join(run_dir, "scalars.jsonl"))["train/loss"]] == [1, 2]   # written once

    from ssd_object_detection_amd.tools import train as T
    single = T.train(_train_cfg(str(tmp_path / "single"), split=True))
    ref = single.get_engine().param.cpu().numpy()
    p0 = type(single)(classes=80, log_dir=str(tmp_path / "p0"), timestamp_dir=False).get_engine().param.cpu().numpy()
    moved = np.abs(ref - p0).max()
    diff = np.abs(res[0][1] - ref)
    assert moved > 1e-4
    # Adam's first steps move EVERY element by ~lr whatever its gradient's size: an element whose near-zero gradient changes sign
    # between the two summation orders (bucketed clip + all-reduce vs accumulate) ends up to 2 lr = `moved` apart after one
    # flipped step.  The mean and the fraction of elements that differ at all are the check; the maximum is bounded by one flip.
    assert float(diff.mean()) < 2e-7 and float((diff > 2e-6).mean()) < 0.03 and diff.max() <= 1.0 * moved


def _nccl_rank_main(port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from ssd_object_detection_amd import optimizers
    from ssd_object_detection_amd.models import SSDObjectDetectionModel
    model = SSDObjectDetectionModel(classes=80, log_dir="gpurun_out/dp_test", seed=5, distributed=True, timestamp_dir=False)
    imgs, cls_l, box_l = _make_inputs(model)
    image, (cls, loc, mask) = model.make_batch(imgs, cls_l, box_l)
    opt = optimizers.Adam(1e-3)
    for _ in range(2):
        model._train_step(image, cls, loc, mask, opt)
    torch.cuda.synchronize()
    q.put(model.get_engine().param.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_backend_single_rank_equals_the_local_step():
    """The distributed train step on the `nccl` backend (RCCL): the one-GPU pool cannot hold two RCCL ranks, but a
    communicator of ONE rank runs every call the N-rank job makes -- bucketed asynchronous all-reduce of the flat fp32
    gradient on the exchange stream, the per-bucket optimizer behind it, the events between the three streams -- and must
    reproduce the non-distributed step (sum over one rank, scale 1 / world = 1): two steps, parameters compared."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_rank_main, args=(port, q))
    p.start()
    dp_param = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0

    from ssd_object_detection_amd import optimizers
    from ssd_object_detection_amd.models import SSDObjectDetectionModel
    model = SSDObjectDetectionModel(classes=80, log_dir="gpurun_out/dp_test", seed=5, timestamp_dir=False)
    p0 = model.get_engine().param.cpu().numpy().copy()
    imgs, cls_l, box_l = _make_inputs(model)
    image, (cls, loc, mask) = model.make_batch(imgs, cls_l, box_l)
    opt = optimizers.Adam(1e-3)
    for _ in range(2):
        model._train_step(image, cls, loc, mask, opt)
    ref = model.get_engine().param.cpu().numpy()
    moved = np.abs(ref - p0).max()
    assert moved > 1e-4
    diff = np.abs(dp_param - ref)
    # (the distributed step clips per tensor before the exchange and applies Adam per bucket: same arithmetic per element)
    assert float(diff.mean()) < 1e-7 and float((diff > 1e-6).mean()) < 0.02 and diff.max() <= 0.5 * moved
