#!/usr/bin/env python3
"""Benchmark of the SSD300 data-parallel hot path on MI355X (contract: see the task prompt / DESIGN.md section 8).

One "step" = one full training step of the reference's path on one batch of synthetic input that is already
resident in HBM: batched anchor matching + encoding (A3-A5), image normalisation (x-0.5)*2 to bf16 (A8), conv stack
forward (A1), loss forward+backward (A6), conv stack backward, per-variable clip + Adam (A7); with N > 1 ranks, one
bucketed RCCL all-reduce of the clipped gradients.  Workload = BASELINE.json configs[2]: SSD300, per-GPU batch 64.

`python bench.py --gpus N` starts the N rank processes itself (children are spawned before the parent touches a GPU;
nothing is exec'ed); under torch.distributed.run (WORLD_SIZE set) it is one of the ranks.

Prints ONE JSON line (rank 0).  `roofline` is for the convolution launches (MFMA-bound); `roofline_match`,
`roofline_loss`, `roofline_detect` (score + decode + NMS) and `roofline_prep` are the HBM-bound stages; `m2` is the
BASELINE metric's second half (anchor-match + NMS microseconds per image); `cfg5_anchors` runs the anchor-side kernels at
BASELINE configs[4]'s 24 564 anchors; `cpu_baseline` is the oracle port timed on this host's cores.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_FWD_PER_IMAGE = 57.554e9
FLOP_TRAIN_PER_IMAGE = 172.66e9 - 0.31e9   # fwd + dgrad + wgrad, first-layer dgrad not needed (every head gradient row counted)
FLOP_HEADS_FWD_PER_IMAGE = 2 * 4.231e9  # the twelve head convolutions
# The heads' data + weight gradients are computed from the rows the loss selects (positives + mined negatives) only; the
# rest of the dense 2 x FLOP_HEADS_FWD is multiplication by exact zeros and is NOT executed.  Rooflines below count the
# FLOPs that run: the dense network minus that part, plus what the sparse kernels do for the measured batch's rows.
FLOP_TRUNK_TRAIN_PER_IMAGE = FLOP_TRAIN_PER_IMAGE - 2 * FLOP_HEADS_FWD_PER_IMAGE
PEAK_BF16_TFLOPS = 2500.0               # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0
MATCH_BYTES_PER_ANCHOR = 53             # priors f64 in (32) + cls i32 + loc f32x4 + mask u8 out; + 20 B per gt box
CFG5_GRIDS = ((64, 64), (32, 32), (16, 16), (8, 8), (4, 4), (2, 2), (1, 1))      # A = 24 564
CFG5_RATIOS = ((2,), (2, 3), (2, 3), (2, 3), (2, 3), (2,), (2,))
CFG5_S_REF = (20, 51, 133, 215, 297, 379, 461, 543)      # SSD512-style scales / 512 (no reference counterpart)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch (BASELINE config: 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    return ap.parse_args()


# --------------------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: the parent only spawns and waits
# --------------------------------------------------------------------------------------------------------------------
def spawn_ranks(args):
    import torch                                     # device_count() does not initialise the GPU
    n_dev = torch.cuda.device_count()
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in env:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            env["MASTER_PORT"] = str(s.getsockname()[1])
    if n_dev < args.gpus:                            # rehearsal on a smaller box: ranks share GPUs, RCCL cannot
        env.setdefault("SSD_DIST_BACKEND", "gloo")
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(args.gpus):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0, _ = procs[0].communicate()
    rc = procs[0].returncode
    deadline = time.time() + 120
    for p in procs[1:]:
        try:
            p.wait(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()                                 # exactly the PID we started
            p.wait()
        rc = rc or p.returncode
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    return rc


# --------------------------------------------------------------------------------------------------------------------
def timed(torch, fn, reps, warm=1):
    """Average seconds per call from HIP events on the current stream (the stream the C ABI launches on)."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def graph_timed(torch, fn, reps):
    """The same with the call captured in a hipGraph (host launch overhead is not in the number)."""
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            fn()
    return timed(torch, g.replay, reps)


def hbm_entry(kernel, nbytes, seconds, per_image, **extra):
    d = {"bound": "hbm", "kernel": kernel, "achieved": round(nbytes / seconds / 1e9, 1), "peak": PEAK_HBM_GBS,
         "unit": "GB/s", "frac": round(nbytes / seconds / 1e9 / PEAK_HBM_GBS, 4), "traffic": None,
         "algorithmic_bytes": int(nbytes), "us_per_image": round(seconds / per_image * 1e6, 4)}
    d.update(extra)
    return d


def pmc_profile(name):
    """Newest committed PMC summary profiles/r*_<name>.json (collected by separate rocprofv3 --pmc passes, see
    tools_dev/collect_pmc.sh) or None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s.json" % name)))
    return (json.load(open(files[-1])), os.path.basename(files[-1])) if files else (None, None)


def nms_inputs(torch, B, A, dtype, seed=7, n_hot=320):
    """SURVEY.md 8(d) NMS input: logits N(0,1) with the background logit +4 (background-dominated, as a trained detector's
    are) and, so that ~200-400 candidates per image survive score 0.3 and neighbouring priors fire on the same class,
    n_hot/8 clusters per image of 8 anchors within a 40-anchor window boosted by U(7,11) on one class (the construction of
    tests/test_detect_gpu.py:synth_logits); box offsets N(0, 0.2)."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    conf = torch.randn((B, A, 81), generator=g, device="cuda")
    conf[..., 80] += 4.0
    nc = max(1, n_hot // 8)
    centre = torch.randint(0, A - 40, (B, nc, 1), generator=g, device="cuda")
    idx = (centre + torch.randint(0, 40, (B, nc, 8), generator=g, device="cuda")).reshape(B, -1)
    k = torch.randint(0, 80, (B, nc, 1), generator=g, device="cuda").expand(B, nc, 8).reshape(B, -1)
    boost = torch.rand((B, nc * 8), generator=g, device="cuda") * 4.0 + 7.0
    conf[torch.arange(B, device="cuda")[:, None], idx, k] += boost
    loc = torch.randn((B, A, 4), generator=g, device="cuda") * 0.2
    return conf.to(dtype).contiguous(), loc.to(dtype).contiguous()


def run_rank(args):
    if os.environ.get("SSD_BENCH_WATCHDOG"):             # development: dump every thread's stack and exit after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["SSD_BENCH_WATCHDOG"]), exit=True)
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()))   # (modulo: lets a 1-GPU box rehearse N ranks)
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL needs one GPU per rank; with fewer GPUs than ranks (a one-GPU box rehearsing the multi-rank path) fall back to gloo
        backend = os.environ.get("SSD_DIST_BACKEND", "nccl" if torch.cuda.device_count() >= world else "gloo")
        try:                                    # bind the communicator to this rank's GPU up front (no lazy-init surprises)
            dist.init_process_group(backend, device_id=torch.device("cuda", torch.cuda.current_device())
                                    if backend == "nccl" else None)
        except TypeError:
            dist.init_process_group(backend)

    import ssd_object_detection_amd.ops as ops
    from ssd_object_detection_amd import optimizers
    for kv in filter(None, os.environ.get("SSD_BENCH_KNOBS", "").split(",")):      # development: NAME=VALUE dispatch overrides
        from ssd_object_detection_amd import _lib                                  # (same-box A/B of a kernel choice)
        name, val = kv.split("=")
        _lib.check(_lib.lib().ssd_dev_knob(name.encode(), int(val)))
    from ssd_object_detection_amd.models import SSDObjectDetectionModel
    from ssd_object_detection_amd.data_loaders.synthetic import synth_batch_gt

    B = args.batch
    model = SSDObjectDetectionModel(classes=80, log_dir="gpurun_out/bench", seed=0, distributed=world > 1,
                                    timestamp_dir=False)
    eng = model.get_engine()
    opt = optimizers.Adam(optimizers.ExponentialDecay(1e-3, 100, 0.99), beta_1=0.9, beta_2=0.999, epsilon=1e-7)

    # synthetic input, resident in HBM: NBATCH distinct batches cycled (images uniform[0,1) f32 as the loader contract
    # delivers them, COCO-shaped boxes); the normalisation (x-0.5)*2 of get_train_set (reference :214) runs inside the step
    NBATCH = 2
    gen = torch.Generator(device="cuda").manual_seed(1234 + rank)
    batches = []
    for i in range(NBATCH):
        img = torch.rand((B, 300, 300, 3), generator=gen, device="cuda", dtype=torch.float32)
        cls_l, box_l = synth_batch_gt((rank * NBATCH + i) * B, B)
        batches.append((img, ops.pack_gt(box_l, cls_l)))
    pset = model._pset
    match_out = None
    xbuf = torch.empty((B, 300, 300, 8), dtype=torch.bfloat16, device="cuda")      # the network input, one buffer for every step

    last = {}

    # the whole loop body on the model's high-priority stream (what tools/train.py's loop does too): no hand-over between the
    # caller's stream and the step's main stream per step.  SSD_BENCH_STEP_STREAM=0: the loop on the default stream.
    step_stream = model.main_stream() if os.environ.get("SSD_BENCH_STEP_STREAM", "1") == "1" else None

    def step(i):
        nonlocal match_out
        if step_stream is not None and torch.cuda.current_stream() != step_stream:
            with torch.cuda.stream(step_stream):
                return step(i)
        img, gt = batches[i % NBATCH]
        match_out = model.match_async(gt, out=match_out)     # A3-A5 on the device, side stream, under the forward pass
        cls, loc, mask = match_out
        x = ops.image_prep(img, normalize=True, out=xbuf)    # A8: (x-0.5)*2, bf16, 8 channels -- one fused pass
        last["conf"], last["loc"], _ = model._train_step(x, cls, loc, mask, opt)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed_steps(n, first=0, fn=None):
        fn = fn or step
        sync_all()
        t0 = time.perf_counter()
        for i in range(n):
            fn(first + i)
        sync_all()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    for i in range(args.warmup):
        step(i)
    if world > 1 and model._reducer is not None:
        model._reducer.profile = True           # two events per bucket on the communication stream
    elapsed = timed_steps(args.steps)
    status = float(model.last_info["status"])
    loss_vals = {k: float(model.last_info[k]) for k in ("loc loss", "cls loss pos", "cls loss neg")}

    images = B * world * args.steps
    value = images / elapsed
    result = {
        "metric": "images/sec SSD300 train step, batch 64",
        "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "BASELINE configs[2]: SSD300 full train step (anchor match+encode, image normalisation, conv fwd, "
                               "loss, conv bwd, per-variable clip (64 variables) + Adam), per-GPU batch %d, 80 classes, "
                               "synthetic COCO-shaped boxes" % B,
                   "per_gpu_batch": B, "global_batch": B * world,
                   "parallelism": "dp%d" % world if world > 1 else "single"},
        "loss_check": dict(loss_vals, status=status),
    }
    flops_step = executed_conv_flops(eng, B)          # of the last batch's gradient rows (sparse head backward) or dense
    result["conv_flops_per_step"] = {"executed": flops_step, "dense_equivalent": FLOP_TRAIN_PER_IMAGE * B,
                                     "note": "executed = dense network minus the head gradients' products with all-zero rows"}
    result["frac_of_conv_gemm_roofline"] = round(flops_step / B * value / (world * PEAK_BF16_TFLOPS * 1e12), 4)

    if rank == 0 and world == 1:
        result["config3_train_plus_eval"] = train_plus_eval(torch, ops, model, pset, B, step, last, timed_steps, args)

    if world > 1:
        result["comm"] = comm_report(torch, dist, model, backend, world, elapsed / args.steps, timed_steps,
                                     min(args.steps, 10), args.warmup + args.steps)

    if rank == 0 and not args.no_kernel_timing:
        kernel_sections(np, torch, ops, model, eng, pset, batches, match_out, B, result, step)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(np, torch)

    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()                          # rank 0 was still timing kernels: leave together
        dist.destroy_process_group()


def train_plus_eval(torch, ops, model, pset, B, step, last, timed_steps, args):
    """BASELINE configs[2] as titled: "full train step incl. backbone MFMA convs + NMS eval pass".  The same step with the
    inference post-processing (A9 + A9': ssd_score_decode + per-image, per-class ssd_nms) appended on the step's stream.  A
    randomly initialised network puts no anchor above the detection threshold (its softmax is ~1/81 everywhere), which would
    make the NMS a no-op: the pass runs on SURVEY.md 8(d)'s NMS input (background-dominated logits with clustered hot anchors,
    ~290 candidates per image) in the step's own logit dtype; the same pass on the step's own logits is timed beside it."""
    n = max(5, min(args.steps, 20))
    conf_s, loc_s = nms_inputs(torch, B, pset.A, torch.bfloat16)
    stats = {}

    def step_eval(i, own=False):
        step(i)
        conf, loc = (last["conf"], last["loc"]) if own else (conf_s, loc_s)
        score, dcls, box, cand = ops.score_decode(conf, loc, pset, 0.3)
        keep = ops.nms(score, dcls, box, cand, 0.45, 400)
        stats["cand"], stats["keep"] = cand, keep

    for i in range(2):
        step_eval(i)
    dt = timed_steps(n, fn=step_eval) / n
    cand = float(stats["cand"].sum().item()) / B
    kept = float(stats["keep"].sum().item()) / B
    dt_own = timed_steps(n, fn=lambda i: step_eval(i, own=True)) / n
    return {"workload": "BASELINE configs[2]: the full train step + NMS eval pass (ssd_score_decode + ssd_nms, score 0.3, IoU 0.45, "
                        "<= 400 candidates per image) on SURVEY 8(d)'s NMS input, bf16 logits, batch %d" % B,
            "images_per_sec": round(B / dt, 2), "ms_per_step": round(dt * 1e3, 3), "steps": n,
            "candidates_per_image": round(cand, 1), "kept_per_image": round(kept, 1),
            "on_the_steps_own_logits": {"ms_per_step": round(dt_own * 1e3, 3),
                                        "candidates_per_image": round(float(stats["cand"].sum().item()) / B, 1),
                                        "note": "random-init weights: nothing passes score 0.3, the NMS has no work"}}


def executed_conv_flops(eng, B):
    """Convolution FLOPs one step of batch B executes: the dense trunk + head forward, plus for the heads' backward either the
    dense 2 x forward (dense path) or what the sparse kernels do for the rows of the engine's last loss call: per row of
    level l the Z GEMM (npad x 9 Cin MACs) and the gathered weight gradient (the same count)."""
    hgb = eng.head_grad_buffers(B)
    if hgb is None:
        return FLOP_TRAIN_PER_IMAGE * B
    counts = hgb.count.cpu().tolist()
    sparse = sum(2 * 2.0 * counts[l] * hgb.npad[l] * 9 * c for l, (_, _, c) in enumerate(eng.fm))
    return FLOP_TRUNK_TRAIN_PER_IMAGE * B + sparse


def calibration(torch, ops, target_ms=60.0):
    """Same-run yardstick (dev entry ssd_dev_mfma_calibration): an LDS-fed bf16 MFMA loop on random operands without memory
    traffic, >= 50 ms, with the in-kernel clock from s_memtime / s_memrealtime stamps (MI355X_MICROARCH.md, DVFS give-back 6)."""
    import ctypes
    from ssd_object_detection_amd import _lib
    L = _lib.lib()
    nwg = L.ssd_dev_mfma_calibration_workgroups()
    stamps = torch.zeros((2 * nwg,), dtype=torch.int64, device="cuda")
    sink = torch.empty((512 * nwg,), dtype=torch.float32, device="cuda")

    def run(iters):
        _lib.check(L.ssd_dev_mfma_calibration(iters, ops._ptr(stamps), ops._ptr(sink), ops._stream()))
    it0 = 2000
    t = timed(torch, lambda: run(it0), 3)                  # sizes the long run (and warms the clocks)
    iters = max(it0, int(it0 * target_ms * 1e-3 / t))
    t = timed(torch, lambda: run(iters), 1, warm=0)
    st = stamps.view(nwg, 2).cpu().numpy().astype("float64")
    import numpy as np
    clk = float(np.median(st[:, 0] / np.maximum(st[:, 1], 1.0))) * 0.1      # x 100 MHz -> GHz
    tf = L.ssd_dev_mfma_calibration_flops(iters) / t / 1e12
    return {"kernel": "k_mfma_calibration: 256 workgroups x 8 waves, v_mfma_f32_16x16x32_bf16 fed by ds_read_b128 from LDS, random "
                      "operands, no global traffic", "ms": round(t * 1e3, 2), "tflops": round(tf, 1),
            "frac_of_nominal_peak": round(tf / PEAK_BF16_TFLOPS, 4), "in_kernel_clock_ghz": round(clk, 3)}


def comm_report(torch, dist, model, backend, world, step_s, timed_steps, n_local, first):
    """What the gradient exchange costs: rank count as the communicator reports it, every bucket's all-reduce timed alone
    (bytes, ms, bus bandwidth 2(N-1)/N * bytes / t), its in-step time from events on the communication stream, and the
    exposed (non-overlapped) part = DP step time - the same step with the exchange switched off (MAX over ranks both)."""
    red = model._reducer
    out = {"backend": backend + (" (RCCL over xGMI)" if backend == "nccl" else " (rehearsal: ranks share a GPU)"),
           "ranks_in_communicator": dist.get_world_size(), "buckets": []}
    in_step = red.bucket_times_ms() if red is not None else []
    if red is not None:
        for k, (t0, t1) in enumerate(red.buckets):
            start, end = red._range(t0, t1)
            view = red.flat[start:end]
            scratch = view.clone()
            torch.cuda.synchronize()
            dist.barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            dist.all_reduce(scratch)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(5):
                dist.all_reduce(scratch)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            nbytes = (end - start) * 4
            out["buckets"].append({"tensors": [t0, t1], "bytes": nbytes, "allreduce_alone_ms": round(ms, 3),
                                   "busbw_GBs": round(2 * (world - 1) / world * nbytes / (ms * 1e-3) / 1e9, 1),
                                   "in_step_ms": round(in_step[k], 3) if k < len(in_step) else None})
    # the same step without the exchange (every rank on its own): what remains is the exposed communication
    model.distributed = False
    dt_local = timed_steps(n_local, first) / n_local
    model.distributed = True
    out["step_ms"] = round(step_s * 1e3, 3)
    out["step_without_exchange_ms"] = round(dt_local * 1e3, 3)
    out["exposed_comm_ms"] = round(max(0.0, step_s - dt_local) * 1e3, 3)
    return out


def in_step_time(torch, eng, nodes, step_fn, steps=5):
    """Seconds per step the weight-gradient launches of `nodes` take inside real train steps (engine.wgrad_probe)."""
    for i in range(2):
        step_fn(i)
    eng.wgrad_probe = {"nodes": set(nodes), "events": []}
    try:
        for i in range(steps):
            step_fn(i)
        torch.cuda.synchronize()
        ev = eng.wgrad_probe["events"]
    finally:
        eng.wgrad_probe = None
    return sum(e0.elapsed_time(e1) for _, e0, e1 in ev) * 1e-3 / steps


def kernel_sections(np, torch, ops, model, eng, pset, batches, match_out, B, result, step_fn):
    """Per-stage timing with HIP events on the launch stream (outside the timed region)."""
    img, gt = batches[0]
    x = ops.image_prep(img, normalize=True)
    cls, loc, mask = match_out
    t_fwd = timed(torch, lambda: eng.forward(x), 3)
    ploc, pconf = eng.forward(x)
    hgb = eng.head_grad_buffers(B)              # the loss gradient as compact rows (None: dense head path selected)
    if hgb is not None:
        ops.ssd_loss_heads(pconf, ploc, cls, loc, mask, hgb)
        t_bwd = timed(torch, lambda: eng.backward(None, None, heads=hgb), 3)
    else:
        _, dconf, dloc = ops.ssd_loss(pconf, ploc, cls, loc, mask)
        t_bwd = timed(torch, lambda: eng.backward(dloc, dconf), 3)
    conv_flops = executed_conv_flops(eng, B)
    conv_time = t_fwd + t_bwd
    cal = calibration(torch, ops)
    result["calibration"] = cal
    dom = dominant_kernel(torch, ops, eng, B)
    dom_nodes = dom.pop("nodes")
    # memory-side traffic of the convolution launches of one step, from rocprofv3 PMC passes (separate --pmc FETCH_SIZE /
    # WRITE_SIZE runs, FETCH_SIZE doubled: gfx950 tallies 128-byte requests at 64 B, MI355X_MICROARCH.md "HBM"); the file
    # records the commit it was collected at.  Only valid for the batch it was collected at.
    conv_pmc, conv_pmc_file = pmc_profile("conv_pmc")
    conv_traffic = int(conv_pmc["conv_hbm_bytes_per_step"]) if (B == 64 and conv_pmc) else None
    agg = conv_flops / conv_time / 1e12
    # `roofline` is the dominant kernel of the step (most device time in profiles/r03_bench_kernel_stats.csv), timed alone
    # with HIP events on its launch stream; the aggregate over every convolution launch of a step is beside it.  Both against
    # the nominal dense bf16 peak and against what this device sustained for the same instruction mix in this run.
    # the same launches where they run: HIP events around them on the engine's side stream during real steps (they share the
    # CUs with the high-priority data-gradient stream there), mean over the probed steps
    # (rank 0 alone is in this function: its probe steps must not enter a collective the other ranks never join)
    was_distributed = getattr(model, "distributed", False)
    model.distributed = False
    try:
        in_step = in_step_time(torch, eng, dom_nodes, step_fn)
    finally:
        model.distributed = was_distributed
    dom["us_per_step_in_step"] = round(in_step * 1e6, 1)
    dom["achieved_in_step"] = round(dom["flops"] / in_step / 1e12, 2)
    dom["frac_in_step"] = round(dom["flops"] / in_step / 1e12 / PEAK_BF16_TFLOPS, 4)
    dom["definition"] = ("achieved / frac: the kernel's launches of one step timed ALONE on the launch stream (us_per_step); "
                         "*_in_step: the same launches timed inside real steps, beside the other stream's kernels; "
                         "aggregate: every convolution launch of a step, executed FLOPs / forward + backward time; "
                         "(rounds 1-2 reported the aggregate as the top-level figure: compare `aggregate` across rounds)")
    result["roofline"] = dict(
        dom, bound="mfma", peak=PEAK_BF16_TFLOPS, unit="TFLOP/s", frac=round(dom["achieved"] / PEAK_BF16_TFLOPS, 4),
        frac_of_calibration=round(dom["achieved"] / cal["tflops"], 4), traffic=(conv_pmc or {}).get("dominant_kernel_hbm_bytes_per_step"),
        traffic_note="HBM bytes of the kernel's launches of one step (PMC); algorithmic: each layer's x and dy read once = %d" % dom["algorithmic_bytes"],
        traffic_source=conv_pmc_file, traffic_commit=(conv_pmc or {}).get("commit"),
        aggregate={"kernel": "all convolution launches of one step (forward + backward, two streams)",
                   "achieved": round(agg, 2), "frac": round(agg / PEAK_BF16_TFLOPS, 4),
                   "frac_of_calibration": round(agg / cal["tflops"], 4), "flops_executed": conv_flops,
                   "traffic": conv_traffic, "fwd_ms": round(t_fwd * 1e3, 3), "bwd_ms": round(t_bwd * 1e3, 3),
                   "fwd_tflops": round(FLOP_FWD_PER_IMAGE * B / t_fwd / 1e12, 2)})

    # ---- anchor matching (A3-A5): graph-replayed ----
    total_gt = gt[3]
    t_match = graph_timed(torch, lambda: ops.match_encode(*gt, pset, 0.5, out=match_out), 50)
    mbytes = B * pset.A * MATCH_BYTES_PER_ANCHOR + 20 * total_gt
    mpmc, mpmc_file = pmc_profile("match_pmc")
    mtraffic = None
    if B == 64 and mpmc:
        mtraffic = int(sum(2 * v if k.endswith("FETCH_SIZE") else v for k, v in mpmc.items()
                           if k.endswith(("FETCH_SIZE", "WRITE_SIZE"))) * 1024)
    result["roofline_match"] = hbm_entry("ssd_match_encode", mbytes, t_match, B, traffic=mtraffic, traffic_source=mpmc_file,
                                         traffic_commit=(mpmc or {}).get("commit"), total_gt=total_gt,
                                         floor_us_at_6300GBs=round(mbytes / 6.3e12 * 1e6, 2), batch_us=round(t_match * 1e6, 2))

    # ---- loss forward + backward (A6) on the network's bf16 logits of this batch ----
    s = pconf.element_size()
    if hgb is not None:
        # the form the step uses: the gradient leaves as compact rows (only the selected anchors are written), so the
        # algorithmic bytes are one read of conf + loc and the targets; the dense form is timed beside it
        t_loss = graph_timed(torch, lambda: ops.ssd_loss_heads(pconf, ploc, cls, loc, mask, hgb), 30)
        t_dense = graph_timed(torch, lambda: ops.ssd_loss(pconf, ploc, cls, loc, mask), 30)
        # algorithmic bytes = SURVEY.md 8(d)'s figure for the loss (conf + loc read, dconf + dloc written, targets): 3.15 MB per
        # image with bf16 logits -- kept for both forms so that they compare; the rows form does not perform the dense
        # gradient write (bytes_moved_by_this_form)
        lbytes = B * pset.A * (2 * 81 * s + 2 * 4 * s + 16 + 4 + 1)
        counts = hgb.count.cpu().tolist()[:hgb.levels]
        moved = B * pset.A * (81 * s + 4 * s + 16 + 4 + 1) + sum(c * p * 2 for c, p in zip(counts, hgb.npad))
        result["roofline_loss"] = hbm_entry("ssd_loss_fwd_bwd_heads (k_loss_rows, k_loss_hist<2>, <3>, k_hg_count, k_hg_assign, k_loss_grad_rows: six launches, no memset node)",
                                            lbytes, t_loss, B, logits="bf16", batch_us=round(t_loss * 1e6, 2),
                                            dense_gradient_form_us=round(t_dense * 1e6, 2), bytes_moved_by_this_form=int(moved),
                                            gradient_rows_per_level=counts, pixels_per_level=[B * h for h in hgb.hw])
    else:
        t_loss = graph_timed(torch, lambda: ops.ssd_loss(pconf, ploc, cls, loc, mask), 30)
        lbytes = B * pset.A * (2 * 81 * s + 2 * 4 * s + 16 + 4 + 1)      # conf+loc in, dconf+dloc out, gloc f32x4, cls i32, mask u8
        result["roofline_loss"] = hbm_entry("ssd_loss_fwd_bwd (k_loss_rows, k_loss_hist/select, k_loss_grad, k_loss_final)",
                                            lbytes, t_loss, B, logits="bf16", batch_us=round(t_loss * 1e6, 2))

    # ---- inference post-processing (A9 + A9'): score + decode + per-class NMS on the SURVEY 8(d) NMS input ----
    result["roofline_detect"], m2_detect = detect_section(torch, ops, pset, B, torch.float32)
    result["roofline_detect"]["bf16_logits"] = detect_section(torch, ops, pset, B, torch.bfloat16)[1]
    result["m2"] = {"metric": "anchor-match + NMS microseconds per image, batch %d" % B,
                    "match_encode_us": round(t_match / B * 1e6, 4), "score_decode_us": m2_detect["score_decode_us"],
                    "nms_us": m2_detect["nms_us"],
                    "total_us": round(t_match / B * 1e6 + m2_detect["score_decode_us"] + m2_detect["nms_us"], 4)}

    # ---- N1: device-side input preprocessing of a batch of COCO-sized uint8 images ----
    from ssd_object_detection_amd.data_loaders.synthetic import synth_raw_sample
    raw = [synth_raw_sample(i)[0] for i in range(B)]
    hw = np.array([r.shape[:2] for r in raw], np.int32)
    sizes = [int(r.size) for r in raw]
    off = np.zeros(B, np.int64)
    off[1:] = np.cumsum(sizes[:-1])
    flat = torch.from_numpy(np.concatenate([r.reshape(-1) for r in raw])).cuda()
    off_d, hw_d = torch.from_numpy(off).cuda(), torch.from_numpy(hw).cuda()
    xprep = ops.image_resize_prep(flat, off_d, hw_d, 300, True)
    t_prep = timed(torch, lambda: ops.image_resize_prep(flat, off_d, hw_d, 300, True, out=xprep), 20)
    pbytes = int(sum(sizes)) + B * 300 * 300 * 8 * 2
    result["roofline_prep"] = hbm_entry("ssd_image_resize_prep (k_image_resize_prep)", pbytes, t_prep, B)

    # ---- BASELINE configs[1]: batch 32 bf16, anchor-match + loss only ----
    result["config2_match_loss"] = config2_section(torch, ops, pset)

    # ---- BASELINE configs[4] anchor side: 24 564 anchors through priors -> match -> loss -> score/decode -> NMS ----
    result["cfg5_anchors"] = cfg5_section(torch, ops, B)


def dominant_kernel(torch, ops, eng, B):
    """k_conv3x3_wgrad_patch<16,2> (the weight gradients of the 3x3 / stride-1 layers on maps of 75 x 75 and larger): every
    trunk layer whose weight-gradient dispatch resolves to it, launched alone on this stream with the step's own operands;
    algorithmic FLOPs of those launches / their summed time."""
    from ssd_object_detection_amd import _lib
    L = _lib.lib()
    c = eng._acts(B)
    want = None
    layers, nodes, flops, secs, abytes = [], [], 0.0, 0.0, 0
    for i, nd in enumerate(eng.nodes):
        if nd["kind"] != "conv":
            continue
        plan = L.ssd_conv2d_bwd_weight_plan(B, nd["hin"], nd["hin"], nd["cin"], nd["cout"], nd["cout"], nd["k"], nd["stride"],
                                            nd["pt"], nd["pl"], nd["hout"], nd["hout"])
        name = L.ssd_conv_plan_name(plan).decode() if plan > 0 else ""
        if "wgrad_patch<16,2>" not in name:
            continue
        want = name
        wt, bt = eng.conv_params[i]
        x, dy = c["acts"][i], c["gacts"][i + 1]
        if x is None:
            continue
        t = timed(torch, lambda: ops.conv2d_bwd_weight(x, dy, nd["cout"], nd["k"], nd["stride"], nd["pt"], nd["pl"],
                                                      dw=eng.view(wt, eng.grad), dbias=eng.view(bt, eng.grad), ws=eng._ws), 5)
        layers.append("conv%d" % i)
        nodes.append(i)
        secs += t
        abytes += 2 * B * (nd["hin"] * nd["hin"] * nd["cin"] + nd["hout"] * nd["hout"] * nd["cout"]) + 4 * nd["cout"] * 9 * nd["cin"]
        flops += 2.0 * B * nd["hout"] * nd["hout"] * nd["cout"] * 9 * nd["cin"]
    return {"kernel": "%s (weight gradients of %s)" % (want, ", ".join(layers)),
            "launches_per_step": len(layers), "us_per_step": round(secs * 1e6, 1), "achieved": round(flops / secs / 1e12, 2),
            "flops": flops, "algorithmic_bytes": abytes, "nodes": nodes}


def config2_section(torch, ops, pset):
    """BASELINE configs[1]: SSD300, batch 32, bf16, anchor-match + loss only (the reference's default batch size,
    config/default.yml:19) -- microseconds per image and HBM fractions of the two stages on their own."""
    from ssd_object_detection_amd.data_loaders.synthetic import synth_batch_gt
    B = 32
    cls_l, box_l = synth_batch_gt(9000, B)
    gt = ops.pack_gt(box_l, cls_l)
    out = ops.match_encode(*gt, pset, 0.5)
    t_match = graph_timed(torch, lambda: ops.match_encode(*gt, pset, 0.5, out=out), 50)
    g = torch.Generator(device="cuda").manual_seed(32)
    conf = torch.randn((B, pset.A, 81), generator=g, device="cuda").bfloat16()
    loc = (0.5 * torch.randn((B, pset.A, 4), generator=g, device="cuda")).bfloat16()
    t_loss = graph_timed(torch, lambda: ops.ssd_loss(conf, loc, *out), 30)
    mbytes = B * pset.A * MATCH_BYTES_PER_ANCHOR + 20 * gt[3]
    lbytes = B * pset.A * (2 * 81 * 2 + 2 * 4 * 2 + 21)
    return {"workload": "BASELINE configs[1]: SSD300 batch 32 bf16, synthetic COCO 80-class boxes, anchor-match + loss only "
                        "(loss with its dense gradient, ssd_loss_fwd_bwd)",
            "match_encode_us_per_image": round(t_match / B * 1e6, 4), "match_GBs": round(mbytes / t_match / 1e9, 1),
            "match_frac_of_hbm_peak": round(mbytes / t_match / 1e9 / PEAK_HBM_GBS, 4),
            "loss_us_per_image": round(t_loss / B * 1e6, 4), "loss_GBs": round(lbytes / t_loss / 1e9, 1),
            "loss_frac_of_hbm_peak": round(lbytes / t_loss / 1e9 / PEAK_HBM_GBS, 4),
            "match_plus_loss_us_per_image": round((t_match + t_loss) / B * 1e6, 4)}


def detect_section(torch, ops, pset, B, dtype, score_thresh=0.3, iou_thresh=0.45):
    conf, loc = nms_inputs(torch, B, pset.A, dtype)
    sd = ops.score_decode(conf, loc, pset, score_thresh)
    score, dcls, box, cand = sd
    ncand = float(cand.sum().item()) / B
    t_sd = graph_timed(torch, lambda: ops.score_decode(conf, loc, pset, score_thresh), 30)
    t_nms = graph_timed(torch, lambda: ops.nms(score, dcls, box, cand, iou_thresh, 400), 30)
    keep = ops.nms(score, dcls, box, cand, iou_thresh, 400)
    nbytes = B * pset.A * (81 + 4) * conf.element_size()
    per = {"score_decode_us": round(t_sd / B * 1e6, 4), "nms_us": round(t_nms / B * 1e6, 4),
           "candidates_per_image": round(ncand, 1), "kept_per_image": round(float(keep.sum().item()) / B, 1)}
    entry = hbm_entry("ssd_score_decode + ssd_nms (k_score_decode, k_nms)", nbytes, t_sd + t_nms, B,
                      logits=str(dtype).replace("torch.", ""), **per)
    entry["score_decode_GBs"] = round((nbytes + B * pset.A * 25) / t_sd / 1e9, 1)     # + score, cls, box, cand written
    per["GBs"] = entry["achieved"]
    return entry, per


def cfg5_section(torch, ops, B):
    from ssd_object_detection_amd.data_loaders.synthetic import synth_batch_gt
    ps = ops.build_priors(grids=CFG5_GRIDS, s_ref=CFG5_S_REF, ratios=CFG5_RATIOS, in_size=512)
    A = ps.A
    cls_l, box_l = synth_batch_gt(5000, B)
    gt = ops.pack_gt(box_l, cls_l)
    out = ops.match_encode(*gt, ps, 0.5)
    t_match = graph_timed(torch, lambda: ops.match_encode(*gt, ps, 0.5, out=out), 30)
    conf, loc = nms_inputs(torch, B, A, torch.bfloat16, seed=11)
    t_loss = graph_timed(torch, lambda: ops.ssd_loss(conf, loc, *out), 20)
    score, dcls, box, cand = ops.score_decode(conf, loc, ps, 0.3, 512.0)
    t_sd = graph_timed(torch, lambda: ops.score_decode(conf, loc, ps, 0.3, 512.0), 20)
    t_nms = graph_timed(torch, lambda: ops.nms(score, dcls, box, cand, 0.45, 400), 20)
    mbytes = B * A * MATCH_BYTES_PER_ANCHOR + 20 * gt[3]
    lbytes = B * A * (2 * 81 * 2 + 2 * 4 * 2 + 21)
    # the same geometry end to end: the SSD recipe at 512 x 512 with seven levels (engine.SSD512_TRUNK), batch 16
    from ssd_object_detection_amd.engine import SSDEngine, SSD512_TRUNK, SSD512_NUM_PRIORS
    Bs = 16
    eng = SSDEngine(classes=81, in_size=512, trunk=SSD512_TRUNK, num_priors=SSD512_NUM_PRIORS, seed=0)
    assert eng.A == A
    img = torch.rand((Bs, 512, 512, 3), device="cuda")
    gts = ops.pack_gt(box_l[:Bs], cls_l[:Bs])
    tgt = ops.match_encode(*gts, ps, 0.5)

    def step512():
        ops.match_encode(*gts, ps, 0.5, out=tgt)
        x = ops.image_prep(img, normalize=True)
        ploc, pconf = eng.forward(x)
        hg = eng.head_grad_buffers(Bs)
        if hg is not None:
            ops.ssd_loss_heads(pconf, ploc, *tgt, hg)
            eng.backward(None, None, heads=hg)
        else:
            _, dconf, dloc = ops.ssd_loss(pconf, ploc, *tgt)
            eng.backward(dloc, dconf)
        eng.clip_scales(0.01)
        eng.adam(1e-3, eng.grad, 1.0, True)
    t_step = timed(torch, step512, 5, warm=2)
    del eng
    # the same step on configs[4]'s own trunk: ResNet-50 (resnet_engine.py: single stream, no fused schedule yet)
    from ssd_object_detection_amd.resnet_engine import ResNet50SSDEngine
    eng = ResNet50SSDEngine(classes=81, seed=0)
    assert eng.A == A
    t_step_r50 = timed(torch, step512, 3, warm=2)
    n_r50 = eng.n_params
    del eng
    # configs[4]'s "fp8 MFMA convs": the >= 256-channel 3x3 layers of that trunk forward in block-scaled fp8 (MX e4m3 on
    # v_mfma_scale_f32_16x16x128_f8f6f4, csrc/fp8conv.hip) beside the bf16 kernels, batch 16; activations are quantised per call
    # (timed separately), the filters once.  Error vs fp32: tests/test_fp8_gpu.py (3.7e-2 relative L2 from the format itself).
    fp8 = {"layers": [], "note": "first correct kernel: the generic 128x128 implicit GEMM at one byte per element, no LDS patch reuse"}
    tot8 = tot16 = totq = flops = 0.0
    for name, Hh, Cin, Cout in (("block3_conv2", 128, 256, 256), ("block3_conv3", 128, 256, 256), ("conv10", 64, 256, 512), ("conv11", 64, 512, 512)):
        xx = torch.randn((Bs, Hh, Hh, Cin), device="cuda").relu().bfloat16()
        ww = (torch.randn((Cout, 3, 3, Cin), device="cuda") / (9 * Cin) ** 0.5).bfloat16()
        bb = torch.zeros(Cout, device="cuda")
        wq, ws8 = ops.quantize_mx_fp8(ww)
        xq, xs8 = ops.quantize_mx_fp8(xx)
        y8 = ops.conv3x3_fwd_mxfp8(xq, xs8, wq, ws8, bb)
        y16 = ops.conv2d_fwd(xx, ww, bb, 1, 1, 1, Hh, Hh, True)
        t8 = timed(torch, lambda: ops.conv3x3_fwd_mxfp8(xq, xs8, wq, ws8, bb, out=y8), 10)
        t16 = timed(torch, lambda: ops.conv2d_fwd(xx, ww, bb, 1, 1, 1, Hh, Hh, True, out=y16), 10)
        tq = timed(torch, lambda: ops.quantize_mx_fp8(xx), 10)
        fl = 2.0 * Bs * Hh * Hh * Cout * 9 * Cin
        err = float((y8.float() - y16.float()).norm() / y16.float().norm())
        fp8["layers"].append({"layer": name, "shape": [Bs, Hh, Hh, Cin, Cout], "fp8_us": round(t8 * 1e6, 1), "bf16_us": round(t16 * 1e6, 1),
                              "quantise_input_us": round(tq * 1e6, 1), "fp8_tflops": round(fl / t8 / 1e12, 1),
                              "bf16_tflops": round(fl / t16 / 1e12, 1), "rel_l2_vs_bf16_kernel": round(err, 4)})
        tot8 += t8; tot16 += t16; totq += tq; flops += fl
        del xx, ww, xq, xs8, y8, y16
    fp8.update({"fp8_us": round(tot8 * 1e6, 1), "bf16_us": round(tot16 * 1e6, 1), "quantise_us": round(totq * 1e6, 1),
                "fp8_tflops": round(flops / tot8 / 1e12, 1), "frac_of_fp8_peak_5PF": round(flops / tot8 / 5e15, 4)})
    return {"train_step": {"workload": "SSD recipe at 512x512, 7 levels, %d anchors, batch %d, bf16 (VGG-style trunk, engine.SSD512_TRUNK): "
                                       "match + prep + fwd + loss + bwd + 72-variable clip + Adam" % (A, Bs),
                           "images_per_sec": round(Bs / t_step, 1), "ms_per_step": round(t_step * 1e3, 3)},
            "train_step_resnet50": {"workload": "BASELINE configs[4]: SSD512 on a ResNet-50 trunk (v1.5, folded batch norm; %d parameters), 7 levels, "
                                                "%d anchors, batch %d, bf16: match + prep + fwd + loss + bwd + clip + Adam, one stream" % (n_r50, A, Bs),
                                    "images_per_sec": round(Bs / t_step_r50, 1), "ms_per_step": round(t_step_r50 * 1e3, 3)},
            "config": {"workload": "BASELINE configs[4] anchors: A=%d (grids 64,32,16,8,4,2,1; per cell 4,6,6,6,6,4,4), batch %d, "
                                   "anchor-side kernels; the ResNet-50 trunk and the fp8 forward are the sub-entries beside it; no reference counterpart" % (A, B)},
            "match_encode_us_per_image": round(t_match / B * 1e6, 4), "match_GBs": round(mbytes / t_match / 1e9, 1),
            "loss_us_per_image": round(t_loss / B * 1e6, 4), "loss_GBs": round(lbytes / t_loss / 1e9, 1),
            "score_decode_us_per_image": round(t_sd / B * 1e6, 4), "nms_us_per_image": round(t_nms / B * 1e6, 4),
            "candidates_per_image": round(float(cand.sum().item()) / B, 1), "fp8_forward": fp8}


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
    """The oracle port of the same step on the host cores, on a bounded sample (about 25 s): anchor matching by the literal
    C restatement of utils/bbox.py (one thread, as the reference's generator is; plus a per-n_t sweep and an all-cores
    run, one image per worker), network forward/backward by the plain-PyTorch fp32 restatement on all cores, loss by the
    numpy float64 restatement, NMS by the C restatement of the build-defined NMS."""
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
    out = {"value": round(Bc * reps / tot, 3), "unit": "images/sec", "cores": cores, "kind": "port",
           "sample": "%d steps of batch %d (same step: C port of utils/bbox.py matching single-threaded, torch-CPU fp32 "
                     "network fwd+bwd on all cores, numpy f64 loss); TensorFlow itself is not installable here" % (reps, Bc),
           "match_us_per_image_1core": round(tmatch / (Bc * reps) * 1e6, 1)}

    # BASELINE configs[0]: ONE synthetic image through the same CPU step (the reference's plumbing case)
    Bc, cls_l, box_l, img = 1, cls_l[:1], box_l[:1], img[:1].contiguous()
    one_step()
    reps1, tot1 = 0, 0.0
    t_begin = time.perf_counter()
    while time.perf_counter() - t_begin < 4.0 and reps1 < 20:
        t = one_step()
        tot1 += t["match"] + t["net+loss"]
        reps1 += 1
    out["config0_one_image"] = {"value": round(reps1 / tot1, 3), "unit": "images/sec", "sample": "%d steps of batch 1" % reps1,
                                "workload": "BASELINE configs[0]: 1 synthetic 300x300 image, CPU path (oracle port)"}

    # matching by ground-truth count (SURVEY.md 8(d) sweep), one core, bounded to ~1 s per point
    sweep = {}
    for n_t in (1, 8, 32, 64, 93):
        cl, bl = synth_batch_gt(7000, 3, n_t=n_t)
        t0, n = time.perf_counter(), 0
        while n < 3 and time.perf_counter() - t0 < 1.5:
            c_oracle.match_encode(cl[n], bl[n], pri, 0.5)
            n += 1
        sweep[str(n_t)] = round((time.perf_counter() - t0) / n * 1e6, 1)
    out["match_us_per_image_1core_by_nt"] = sweep
    # one worker thread per core over the images of a COCO-shaped batch (ctypes releases the GIL inside the C matcher, so
    # threads are real parallelism here and no process is forked from one that holds the GPU)
    from concurrent.futures import ThreadPoolExecutor
    cl, bl = synth_batch_gt(0, 64)
    with ThreadPoolExecutor(cores) as pool:
        t0 = time.perf_counter()
        list(pool.map(lambda cb: c_oracle.match_encode(cb[0], cb[1], pri, 0.5)[0][0], zip(cl, bl)))
        out["match_us_per_image_all_cores"] = round((time.perf_counter() - t0) / 64 * 1e6, 1)
    # NMS (build-defined; C restatement) on the same kind of input as roofline_detect, one core
    r2 = np.random.default_rng(7)
    conf = r2.standard_normal((4, 8732, 81)).astype(np.float32)
    conf[..., 80] += 4.0
    for b in range(4):                                    # the hot clusters of nms_inputs()
        for c0 in r2.integers(0, 8732 - 40, 40):
            conf[b, c0 + r2.integers(0, 40, 8), int(r2.integers(0, 80))] += r2.uniform(7, 11, 8).astype(np.float32)
    loc = (r2.standard_normal((4, 8732, 4)) * 0.2).astype(np.float32)
    t_sd = t_nms = 0.0
    ncand = 0
    for i in range(4):
        t0 = time.perf_counter()
        sc, cl_, cand = O.score(conf[i], 0.3)
        box = O.decode(loc[i], pri).astype(np.float32)
        t1 = time.perf_counter()
        c_oracle.nms(sc.astype(np.float32), cl_.astype(np.int32), box, cand, 0.45, 400)
        t2 = time.perf_counter()
        t_sd += t1 - t0
        t_nms += t2 - t1
        ncand += int(np.sum(cand))
    out["score_decode_us_per_image_1core_numpy"] = round(t_sd / 4 * 1e6, 1)
    out["nms_us_per_image_1core"] = round(t_nms / 4 * 1e6, 1)
    out["nms_candidates_per_image"] = round(ncand / 4, 1)
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    run_rank(args)


if __name__ == "__main__":
    main()
