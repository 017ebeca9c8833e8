"""bench.py's device-memory plan for the driver's own command (`--gpus 1 --steps 20 --warmup 5`) in every mode, from host arithmetic
(bsx_batch_plan_bytes — the same function bsx_batch_create sizes its allocations with): under 0.9 of the device.  Round 4's line lost
C4 and the PCIe-inclusive leg to out-of-memory under exactly that command."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bsmap_amd as B  # noqa: E402

spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)

ENTRIES = {"pe": 1_476_000_000, "se": 1_476_000_000, "trim": 1_476_000_000, "rrbs": 56_000_000}


@pytest.mark.parametrize("mode", sorted(bench.MODES))
@pytest.mark.parametrize("steps,warmup", [(20, 5), (12, 3), (6, 3)])
def test_planned_device_bytes_fit(mode, steps, warmup):
    """every mode with bench.py's own defaults for it (units per step, batches in flight, starting pools)"""
    M = bench.MODES[mode]
    L = B.lib()
    try:
        p = B.make_params(**M["kw"])
        B_, nfl, limits = bench.mode_defaults(mode)
        plan = bench.memory_plan(B, p, M["pe"], B_, steps, warmup, nfl, ENTRIES[mode], True, mode == "rrbs", limits=limits)
    finally:
        assert L.bsx_set_heavy_limits(0, 0) == 0
    assert plan["peak_frac_of_device"] < 0.9, plan
    # the ring: resident reads do not grow with --steps
    assert plan["per_batch_GB"]["per_unit"] < 16.0, plan


def test_plan_follows_the_batch_size_and_the_limits():
    p = B.make_params(s=16, v=6, I=4, m=28, x=500, pairend=1)
    a = B.plan_bytes(p, 1 << 20, True, 1_476_000_000)
    b = B.plan_bytes(p, 8 << 20, True, 1_476_000_000)
    assert b["per_unit"] > 7 * a["per_unit"] and b["scratch"] == a["scratch"] and b["pools"] > a["pools"]   # (beyond 2^21 units the pools follow the batch)
    assert B.plan_bytes(p, 2 << 20, True, 1_476_000_000)["pools"] == a["pools"]
    small = B.plan_bytes(p, 4096, True, 1_000_000)
    assert small["pools"] < a["pools"] / 5 and small["scratch"] < a["scratch"]
    L = B.lib()
    try:
        assert L.bsx_set_heavy_limits(2048, 8192) == 0
        c = B.plan_bytes(p, 1 << 20, True, 1_476_000_000)
        assert c["pools"] < a["pools"] / 8
    finally:
        L.bsx_set_heavy_limits(0, 0)
    assert B.plan_bytes(p, 1 << 20, True, 1_476_000_000) == a
