#!/usr/bin/env python3
"""Benchmark of the SSD300 data-parallel hot path on MI355X (contract: see the task prompt / DESIGN.md).

One "step" = one full training step of the reference's path on one batch of synthetic input that is already
resident in HBM: batched anchor matching + encoding (A3-A5), image normalisation to bf16, conv stack forward
(A1), loss forward+backward (A6), conv stack backward, per-tensor clip + Adam (A7); with N > 1 ranks, one RCCL
all-reduce of the clipped gradients.  Workload = BASELINE.json configs[2]: SSD300, per-GPU batch 64.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernels (the conv implicit GEMMs, MFMA-bound);
`roofline_match` is the HBM-bound anchor-matching kernel; `cpu_baseline` is the oracle port of the same step
timed on this host's cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_FWD_PER_IMAGE = 57.554e9
FLOP_TRAIN_PER_IMAGE = 172.66e9 - 0.31e9   # fwd + dgrad + wgrad, first-layer dgrad not needed
PEAK_BF16_TFLOPS = 2500.0               # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0
MATCH_BYTES_PER_IMAGE = 8732 * 53       # + 20 * n_t  


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch (BASELINE config: 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        print("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()))   # (modulo: lets a 1-GPU box rehearse N ranks)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SSD_DIST_BACKEND", "nccl")
        try:                                    # bind the communicator to this rank's GPU up front (no lazy-init surprises)
            dist.init_process_group(backend, device_id=torch.device("cuda", torch.cuda.current_device())
                                    if backend == "nccl" else None)
        except TypeError:
            dist.init_process_group(backend)

    import ssd_object_detection_amd.ops as ops
    from ssd_object_detection_amd import optimizers
    from ssd_object_detection_amd.models import SSDObjectDetectionModel
    from ssd_object_detection_amd.data_loaders.synthetic import synth_batch_gt

    B = args.batch
    model = SSDObjectDetectionModel(classes=80, log_dir="gpurun_out/bench", seed=0, distributed=world > 1,
                                    timestamp_dir=False)
    eng = model.get_engine()
    opt = optimizers.Adam(optimizers.ExponentialDecay(1e-3, 100, 0.99), beta_1=0.9, beta_2=0.999, epsilon=1e-7)

    # synthetic input, resident in HBM: NBATCH distinct batches cycled (images uniform[0,1), COCO-shaped boxes)
    NBATCH = 2
    gen = torch.Generator(device="cuda").manual_seed(1234 + rank)
    batches = []
    for i in range(NBATCH):
        img = torch.rand((B, 300, 300, 3), generator=gen, device="cuda", dtype=torch.float32)
        cls_l, box_l = synth_batch_gt((rank * NBATCH + i) * B, B)
        img = (img - 0.5) * 2            # get_train_set delivers normalised images (reference :214); done once, untimed
        batches.append((img, ops.pack_gt(box_l, cls_l)))
    pset = model._pset
    match_out = None

    def step(i):
        nonlocal match_out
        img, gt = batches[i % NBATCH]
        match_out = model.match_async(gt, out=match_out)     # A3-A5 on the device, side stream, under the forward pass
        cls, loc, mask = match_out
        model._train_step(img, cls, loc, mask, opt)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    sync_all()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    status = float(model.last_info["status"])
    loss_vals = {k: float(model.last_info[k]) for k in ("loc loss", "cls loss pos", "cls loss neg")}

    images = B * world * args.steps
    value = images / elapsed
    result = {
        "metric": "images/sec SSD300 train step, batch 64",
        "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "BASELINE configs[2]: SSD300 full train step (anchor match+encode, conv fwd, loss, conv bwd, "
                               "per-tensor clip + Adam), per-GPU batch %d, 80 classes, synthetic COCO-shaped boxes" % B,
                   "per_gpu_batch": B, "global_batch": B * world,
                   "parallelism": "dp%d" % world if world > 1 else "single"},
        "loss_check": dict(loss_vals, status=status),
        "frac_of_conv_gemm_roofline": round(FLOP_TRAIN_PER_IMAGE * value / (world * PEAK_BF16_TFLOPS * 1e12), 4),
    }

    if rank == 0 and not args.no_kernel_timing:
        # ---- per-kernel timing with HIP events on the launch stream (outside the timed region) ----
        def timed(fn, reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fn()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps * 1e-3

        img, gt = batches[0]
        x = ops.image_prep(img, normalize=False)
        cls, loc, mask = match_out
        t_fwd = timed(lambda: eng.forward(x), 3)
        ploc, pconf = eng.forward(x)
        _, dconf, dloc = ops.ssd_loss(pconf, ploc, cls, loc, mask)
        t_bwd = timed(lambda: eng.backward(dloc, dconf), 3)
        conv_flops = FLOP_TRAIN_PER_IMAGE * B
        conv_time = t_fwd + t_bwd
        # memory-side traffic of the convolution launches of one step, from rocprofv3 PMC passes (separate --pmc FETCH_SIZE /
        # WRITE_SIZE runs of tools_dev/time_step.py, summed by tools_dev/pmc_conv_traffic.py; FETCH_SIZE doubled: gfx950
        # tallies 128-byte requests at 64 B, MI355X_MICROARCH.md "HBM").  Only valid for the batch it was collected at.
        conv_traffic = None
        conv_pmc = os.path.join(ROOT, "profiles", "r01d_conv_pmc.json")
        if B == 64 and os.path.exists(conv_pmc):
            conv_traffic = int(json.load(open(conv_pmc))["conv_hbm_bytes_per_step"])
        result["roofline"] = {
            "bound": "mfma", "kernel": "all convolution launches of one step (k_conv3x3_patch32, k_conv3x3_wgrad_patch, k_conv_igemm_*, k_conv_wgrad_*, k_conv0_*)",
            "achieved": round(conv_flops / conv_time / 1e12, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(conv_flops / conv_time / 1e12 / PEAK_BF16_TFLOPS, 4), "traffic": conv_traffic,
            "fwd_ms": round(t_fwd * 1e3, 3), "bwd_ms": round(t_bwd * 1e3, 3),
            "fwd_tflops": round(FLOP_FWD_PER_IMAGE * B / t_fwd / 1e12, 2)}
        # anchor matching: graph-replayed so that host launch overhead is not in the number
        total_gt = gt[3]
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            ops.match_encode(*gt, pset, 0.5, out=match_out)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                ops.match_encode(*gt, pset, 0.5, out=match_out)
        t_match = timed(lambda: g.replay(), 50)
        mbytes = B * MATCH_BYTES_PER_IMAGE + 20 * total_gt
        # HBM traffic of the dominant matching kernel from rocprofv3 PMC passes (separate --pmc FETCH_SIZE / WRITE_SIZE
        # runs of tools_dev/time_match.py 64:mix; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950)
        traffic = None
        pmc_file = os.path.join(ROOT, "profiles", "r01_match_pmc.json")
        if B == 64 and os.path.exists(pmc_file):
            pmc = json.load(open(pmc_file))
            traffic = int((2 * pmc["k_match_pairs.FETCH_SIZE"] + pmc["k_match_pairs.WRITE_SIZE"]) * 1024)
        result["roofline_match"] = {
            "bound": "hbm", "kernel": "ssd_match_encode (k_match_rows + k_match_pairs + k_match_phase1)",
            "achieved": round(mbytes / t_match / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": round(mbytes / t_match / 1e9 / PEAK_HBM_GBS, 4), "traffic": traffic,
            "algorithmic_bytes": mbytes, "us_per_image": round(t_match / B * 1e6, 4), "total_gt": total_gt}

        # N1: device-side input preprocessing of a batch of COCO-sized uint8 images, HBM-bound:
        # algorithmic bytes = the source bytes read once + the bf16 [B,300,300,8] network input written once
        from ssd_object_detection_amd.data_loaders.synthetic import synth_raw_sample
        raw = [synth_raw_sample(i)[0] for i in range(B)]
        hw = np.array([r.shape[:2] for r in raw], np.int32)
        sizes = [int(r.size) for r in raw]
        off = np.zeros(B, np.int64)
        off[1:] = np.cumsum(sizes[:-1])
        flat = torch.from_numpy(np.concatenate([r.reshape(-1) for r in raw])).cuda()
        off_d, hw_d = torch.from_numpy(off).cuda(), torch.from_numpy(hw).cuda()
        xprep = ops.image_resize_prep(flat, off_d, hw_d, 300, True)
        t_prep = timed(lambda: ops.image_resize_prep(flat, off_d, hw_d, 300, True, out=xprep), 20)
        pbytes = int(sum(sizes)) + B * 300 * 300 * 8 * 2
        result["roofline_prep"] = {
            "bound": "hbm", "kernel": "ssd_image_resize_prep (k_image_resize_prep)", "achieved": round(pbytes / t_prep / 1e9, 1),
            "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(pbytes / t_prep / 1e9 / PEAK_HBM_GBS, 4), "traffic": None,
            "algorithmic_bytes": pbytes, "us_per_image": round(t_prep / B * 1e6, 3)}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(np, torch)

    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()                          # rank 0 was still timing kernels: leave together
        dist.destroy_process_group()



def _usable_cores():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota (a GPU box exposes all of
    the host's CPUs but grants a share of them; more threads than that only thrash)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, n)

def cpu_baseline(np, torch):
    """The oracle port of the same step on the host cores, on a bounded sample: anchor matching by the literal C
    restatement of utils/bbox.py (one thread, as the reference's generator is), network forward/backward by the
    plain-PyTorch fp32 restatement on all cores, loss by the numpy float64 restatement."""
    from oracle import c_oracle, net_oracle, ssd_oracle as O
    from ssd_object_detection_amd.engine import SSD300_TRUNK, SSD300_NUM_PRIORS
    from ssd_object_detection_amd.data_loaders.synthetic import synth_batch_gt
    import math
    cores = _usable_cores()
    torch.set_num_threads(cores)
    Bc = 4
    pri = O.priors()
    rng = np.random.default_rng(0)
    params = {}
    for i, (kind, cin, cout, k, stride, mode, feat) in enumerate(SSD300_TRUNK):
        if kind != "conv":
            continue
        lim = math.sqrt(6.0 / (k * k * (cin + cout)))
        params["conv%d/kernel" % i] = torch.from_numpy(rng.uniform(-lim, lim, (cout, k, k, cin)).astype(np.float32)).requires_grad_(True)
        params["conv%d/bias" % i] = torch.zeros(cout, requires_grad=True)
    fms = [c for (kind, _, c, _, _, _, feat) in SSD300_TRUNK if feat]
    for lvl, (c, n) in enumerate(zip(fms, SSD300_NUM_PRIORS)):
        params["head%d/kernel" % lvl] = torch.from_numpy(rng.uniform(-0.02, 0.02, (n * 85, 3, 3, c)).astype(np.float32)).requires_grad_(True)
        params["head%d/bias" % lvl] = torch.zeros(n * 85, requires_grad=True)
    cls_l, box_l = synth_batch_gt(0, Bc)
    img = torch.from_numpy(rng.random((Bc, 300, 300, 8), dtype=np.float32))
    img[..., 3:] = 0

    def one_step():
        t = {}
        t0 = time.perf_counter()
        tc, tl, tm = [], [], []
        for c, b in zip(cls_l, box_l):
            mc, _, mm, enc, _ = c_oracle.match_encode(c, b, pri, 0.5)
            tc.append(mc); tl.append(enc); tm.append(mm)
        t["match"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        loc, conf = net_oracle.forward(SSD300_TRUNK, SSD300_NUM_PRIORS, 81, params, img, emulate_bf16=False)
        res = O.ssd_loss(np.stack(tc), np.stack(tl), np.stack(tm), loc.detach().numpy(), conf.detach().numpy(), want_grad=True)
        (loc * torch.from_numpy(res["dbox"].astype(np.float32))).sum().add(
            (conf * torch.from_numpy(res["dcls"].astype(np.float32))).sum()).backward()
        t["net+loss"] = time.perf_counter() - t0
        return t

    one_step()                                  # warm
    reps, tot, tmatch = 0, 0.0, 0.0
    t_begin = time.perf_counter()
    while time.perf_counter() - t_begin < 12.0 and reps < 40:
        t = one_step()
        tot += t["match"] + t["net+loss"]
        tmatch += t["match"]
        reps += 1
    return {"value": round(Bc * reps / tot, 3), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "%d steps of batch %d (same step: C port of utils/bbox.py matching single-threaded, torch-CPU fp32 "
                      "network fwd+bwd on all cores, numpy f64 loss); TensorFlow itself is not installable here" % (reps, Bc),
            "match_us_per_image_1core": round(tmatch / (Bc * reps) * 1e6, 1)}


if __name__ == "__main__":
    main()
