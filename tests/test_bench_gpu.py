"""The bench line as the driver invokes it: one rank, and two ranks sharing this box's GPU over gloo (the launcher's rehearsal
mode) -- with the kernel-timing sections ON, which rank 0 runs alone: its probe steps must not enter a collective the other
ranks never join (a round-4 regression test: the in-step probe of the dominant kernel once did, and `--gpus 2` hung)."""
import json
import os
import subprocess
import sys

import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(extra, timeout):
    env = dict(os.environ, SSD_BENCH_WATCHDOG=str(timeout - 20))     # a hang dumps every thread's stack instead of timing out silently
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, cwd=ROOT, env=env, capture_output=True,
                       timeout=timeout)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    return json.loads(p.stdout.decode().strip().splitlines()[-1])


def test_bench_line_one_rank():
    d = run_bench(["--steps", "3", "--warmup", "1", "--no-cpu-baseline"], 420)
    assert d["n_gpus"] == 1 and d["unit"] == "images/sec" and d["value"] > 0 and d["dtype"] == "bf16"
    r = d["roofline"]
    assert r["bound"] == "mfma" and 0 < r["frac"] < 1 and r["us_per_step_in_step"] >= r["us_per_step"] * 0.9
    assert d["loss_check"]["status"] == 0.0


def test_bench_line_two_ranks_rehearsal():
    d = run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], 600)
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 128
    assert "comm" in d and d["comm"]["backend"].startswith("gloo") and "roofline" in d
