"""CPU suite: convolution dispatch coverage.  The library's dispatch queries (ssd_conv2d_*_plan, include/ssd_hip.h) are
host functions -- the dispatch code with launching switched off -- so which kernel serves which layer can be checked
without a GPU:
  * every (forward, data-gradient, weight-gradient) kernel the batch-64 SSD300 train step launches is reached by at
    least one oracle-compared case of tests/test_conv_gpu.py (tests/conv_cases.py), so no kernel of the benchmarked
    step runs unchecked;
  * every case still names the kernels it was written for (the GPU tests assert the same before computing)."""
import pytest

from tests.conv_cases import CASES, FIRST_LAYER_CASE, FULL_SIZE_CASES, WS_BYTES, _geom, plan_name, plan_names


def _network_plans(B):
    from ssd_object_detection_amd import _lib
    from ssd_object_detection_amd.engine import SSD300_TRUNK, SSD300_NUM_PRIORS
    L = _lib.lib()
    out, s, fms = [], 300, []
    for i, (kind, cin, cout, k, stride, mode, feat) in enumerate(SSD300_TRUNK):
        ho, _, pt, _ = _geom(s, s, k, stride, mode)
        if kind == "conv":
            pooled = i + 1 < len(SSD300_TRUNK) and SSD300_TRUNK[i + 1][0] == "pool"
            out.append(("conv%d fwd" % i, L.ssd_conv2d_fwd_plan(B, s, s, cin, cout, k, stride, pt, pt, ho, ho, 2 if pooled else 0, WS_BYTES)))
            if i > 0:
                out.append(("conv%d dgrad" % i, L.ssd_conv2d_bwd_data_plan(B, s, s, cin, cout, k, stride, pt, pt, ho, ho, 0, WS_BYTES)))
            out.append(("conv%d wgrad" % i, L.ssd_conv2d_bwd_weight_plan(B, s, s, cin, cout, cout, k, stride, pt, pt, ho, ho)))
        s = ho
        if feat:
            fms.append((ho, cout))
    for lvl, ((h, c), n) in enumerate(zip(fms, SSD300_NUM_PRIORS)):
        nout, npad = n * 85, (n * 85 + 7) // 8 * 8
        out.append(("head%d fwd" % lvl, L.ssd_conv2d_head_fwd_plan(B, h, h, c, n, 81, WS_BYTES)))
        out.append(("head%d dgrad" % lvl, L.ssd_conv2d_bwd_data_plan(B, h, h, c, npad, 3, 1, 1, 1, h, h, 0, WS_BYTES)))
        out.append(("head%d wgrad" % lvl, L.ssd_conv2d_bwd_weight_plan(B, h, h, c, nout, npad, 3, 1, 1, 1, h, h)))
    return L, out


def test_cases_name_the_kernels_they_reach():
    for case in CASES + FULL_SIZE_CASES + [FIRST_LAYER_CASE]:
        assert plan_names(case[:8]) == case[8], case[:8]


@pytest.mark.parametrize("B", [64, 32, 4])
def test_every_kernel_of_the_train_step_has_an_oracle_case(B):
    L, plans = _network_plans(B)
    assert all(p > 0 for _, p in plans), [n for n, p in plans if p <= 0]
    fused_pool = 0x800                                        # tests/test_conv_gpu.py::test_conv_fwd_pool_fused covers the flag
    tested = set()
    for case in CASES + FULL_SIZE_CASES:
        tested.update(case[8])
    tested.update((FIRST_LAYER_CASE[8][0], FIRST_LAYER_CASE[8][2]))
    for name, plan in plans:
        kernel = plan_name(L, plan & ~fused_pool)
        base = kernel.split("+")[0]
        flags = set(kernel.split("+")[1:])
        hit = [t for t in tested if t.split("+")[0] == base and flags <= set(t.split("+")[1:])]
        assert hit, "%s at batch %d runs %s, which no oracle-compared case reaches" % (name, B, kernel)
