"""CPU suite: the sync-free scalar log (reference models/ssd_model.py:281-285 writes the same five tags
per step) -- ring bookkeeping, tags/values, the deferred status assert, and the rank-mean over a world-2 gloo group."""
import importlib
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

SL = importlib.import_module("ssd_object_detection_amd.utils.scalar_log")


def row(loc, pos, neg, status=0.0):
    return torch.tensor([loc, pos, neg, loc + pos + neg, 3.0, 9.0, 0.5, status], dtype=torch.float32)


def test_tags_values_and_ring_autoflush(tmp_path):
    log = SL.ScalarLog(str(tmp_path), "cpu", capacity=4)
    want = []
    for step in range(1, 11):
        r = row(0.1 * step, 0.2 * step, 0.3 * step)
        log.record("train" if step > 3 else "warmup", step, r, 1e-3 / step)
        want.append(r.numpy())
    assert log.rows_written == 8 and len(log.meta) == 2           # two full rings flushed by themselves
    log.close()
    got = SL.read_scalars(log.path)
    assert set(got) == {s + "/" + t for s in ("warmup", "train") for t in SL.TAGS}
    assert [s for s, _ in got["warmup/loss"]] == [1, 2, 3] and [s for s, _ in got["train/loss"]] == list(range(4, 11))
    for step, v in got["train/loc loss"]:
        assert v == float(want[step - 1][0])                       # fp32 value, bit for bit
    for step, v in got["train/loss"]:
        w = want[step - 1]
        assert v == float(w[0]) + float(w[1]) + float(w[2])        # summed on the host like reference :284
    for step, v in got["train/lr"]:
        assert v == 1e-3 / step


def test_status_assert_fires_at_the_read(tmp_path):
    log = SL.ScalarLog(str(tmp_path), "cpu", capacity=8)
    log.record("train", 1, row(1, 1, 1), 1e-3)
    log.record("train", 2, row(1, 1, 1, status=2.0), 1e-3)
    log.record("train", 3, row(1, 1, 1), 1e-3)
    with pytest.raises(SL.HardNegativeThresholdError, match="step 2"):
        log.flush()
    assert len(SL.read_scalars(log.path)["train/loss"]) == 3       # the rows are on disk before the assert
    log.close()


def _worker(rank, world, port, log_dir, q):
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    log = SL.ScalarLog(log_dir, "cpu", capacity=8, distributed=True)
    for step in (1, 2):
        log.record("train", step, row(1.0 + rank, 2.0 * step, 0.5 + rank * step), 1e-3)
    rows = log.flush()
    log.close()
    q.put((rank, rows))
    dist.barrier()
    dist.destroy_process_group()


def test_rank_mean_world2(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == res[1]                                        # every rank sees the same averaged rows
    for (stage, step, vals) in res[0]:
        np.testing.assert_allclose(vals[:3], (1.5, 2.0 * step, 0.5 + 0.5 * step), rtol=1e-6)
    got = SL.read_scalars(str(tmp_path / "scalars.jsonl"))         # rank 0 alone writes the file
    assert len(got["train/loss"]) == 2


def test_status_1_is_an_error_too(tmp_path):
    """No positives in a step (status 1): the reference's top_k / division raises at that step (models/ssd_model.py:359,368);
    here it surfaces at the read, with its own exception type, after the rows are on disk."""
    log = SL.ScalarLog(str(tmp_path), "cpu", capacity=8)
    log.record("warmup", 1, row(1, 1, 1), 1e-3)
    log.record("warmup", 2, row(0, 0, 0, status=1.0), 1e-3)
    with pytest.raises(SL.NoPositivesError, match="step 2"):
        log.flush()
    assert len(SL.read_scalars(log.path)["warmup/loss"]) == 2
    log.close()


def test_status_3_non_finite_logits(tmp_path):
    """status 3 (a NaN / Inf logit row or offset reached the loss kernel): its own exception type, and it outranks the others."""
    log = SL.ScalarLog(str(tmp_path), "cpu", capacity=8)
    log.record("train", 1, row(1, 1, 1), 1e-3)
    log.record("train", 2, row(float("nan"), 1, 1, status=3.0), 1e-3)
    with pytest.raises(SL.NonFiniteLossError, match="step 2"):
        log.flush()
    assert issubclass(SL.NonFiniteLossError, FloatingPointError)
    log.close()


def test_close_without_collective_does_not_touch_the_process_group(tmp_path, monkeypatch):
    """Unwinding from an exception on one rank: close(collective=False) must write local rows without an all-reduce."""
    log = SL.ScalarLog(str(tmp_path), "cpu", capacity=8, distributed=True)
    log.record("train", 1, row(1, 2, 3), 1e-3)
    monkeypatch.setattr(torch.distributed, "is_initialized", lambda: True)
    monkeypatch.setattr(torch.distributed, "get_rank", lambda: 0)

    def boom(*a, **k):
        raise AssertionError("collective entered")

    monkeypatch.setattr(torch.distributed, "all_reduce", boom)
    log.close(collective=False)
    assert SL.read_scalars(log.path)["train/loss"] == [(1, 6.0)]
