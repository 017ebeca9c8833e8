"""bench.py's host-side logic that runs without a GPU: argument defaults, the CPU helpers, the other_configs command lines"""
import json
import os
import subprocess
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as BN


def test_default_steps_are_a_multiple_of_every_in_flight_count():
    """with 2, 3 or 4 batches in flight every batch runs the same number of timed steps (a ragged last round costs 5-10 %)"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    src = open(os.path.join(ROOT, "bench.py")).read()
    import re
    steps = int(re.search(r'"--steps", type=int, default=(\d+)', src).group(1))
    warm = int(re.search(r'"--warmup", type=int, default=(\d+)', src).group(1))
    assert steps % 2 == 0 and steps % 3 == 0 and steps % 4 == 0 and warm >= 2


def test_modes_cover_baseline_configs():
    tags = sorted(m["tag"] for m in BN.MODES.values())
    assert tags == ["C2", "C3", "C4", "C5"]
    assert BN.MODES["pe"]["metric"].startswith("aligned reads/sec")
    for m in BN.MODES.values():
        assert m["workload"].startswith(m["tag"] + ":")


def test_usable_cpus_and_node_cpus():
    n = BN.usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    pin = BN.node_cpus(0, n)
    assert pin is None or (len(pin) <= n and set(pin) <= set(range(os.cpu_count() or 1)))


def test_other_configs_runs_every_other_mode_in_a_child(monkeypatch):
    """the leg starts one child per mode with the side legs off and copies the child's line; a failing child is reported, not fatal"""
    calls = []

    def fake_run(cmd, capture_output, text, timeout):
        calls.append(cmd)
        mode = cmd[cmd.index("--mode") + 1]
        if mode == "rrbs":
            return types.SimpleNamespace(stdout="no json here\n", stderr="x" * 500 + "hipMalloc -> out of memory", returncode=1)
        if mode == "trim" and "--in-flight" not in cmd:
            return types.SimpleNamespace(stdout="", stderr="first attempt died", returncode=-9)
        line = {"value": 1e6, "ms_per_step": 100.0, "steps": 6, "config": {"workload": "w", "batches_in_flight": 2, "aligned_fraction": 1.0, "pairs_per_step": 1 << 21},
                "roofline": {"per_read": {"n_cand": 5.0}, "dominant_kernel": {"name": "k", "ms_per_step": 1.0, "candidates_per_s": 2.0}}}
        return types.SimpleNamespace(stdout="noise\n" + json.dumps(line) + "\n", stderr="", returncode=0)
    monkeypatch.setattr(subprocess, "run", fake_run)
    res = BN.other_configs(types.SimpleNamespace(genome="hg38", pairs_per_step=1 << 20, units_given=False))
    assert all("--pairs-per-step" not in c for c in calls) and res["C2"]["ms_per_2^20_units"] == 50.0
    # one child per mode; a failing child is tried once more with two batches in flight
    assert [c[c.index("--mode") + 1] for c in calls] == ["se", "rrbs", "rrbs", "trim", "trim"]
    for c in calls:
        assert c[c.index("--other-configs") + 1] == "0" and c[c.index("--cpu-seconds") + 1] == "0" and int(c[c.index("--steps") + 1]) % 6 == 0
    assert res["C2"]["reads_per_s"] == 1e6 and res["C5"]["dominant_kernel"]["name"] == "k" and "error" in res["C4"]
    # the record says WHY: return code and the end of the child's stderr (round 4's line said "list index out of range")
    assert [a["rc"] for a in res["C4"]["attempts"]] == [1, 1] and res["C4"]["attempts"][0]["stderr_tail"].endswith("out of memory")
    assert res["C5"]["failed_attempts"][0]["rc"] == -9


def test_roofline_block_is_flat_short_and_priced_against_the_microbenchmark():
    """the driver's record keeps scalars and the first 120 characters of a string: `bound` fits, `frac` = achieved / peak with the peak taken from the
    microbenchmark's ceiling of the kernel's inner word (not from achieved / utilisation), instructions per evaluation and the per-kernel serial times are top-level"""
    import numpy as np
    assert [BN.inner_word_instructions(m, False) for m in ("pe", "se", "rrbs")] == [18, 15, 12] and BN.inner_word_instructions("pe", True) == 24
    ceil = 660e9
    dk = {"name": "k_hscan_same", "candidates_per_s": 1.2e12, "binding_unit": "valu_issue", "binding_unit_utilisation": 0.64,
          "bound": "VALU issue (against the ceiling of the kernel's own inner word at its 3.4 resident waves per SIMD) at 0.64; the others: texture path 0.60, LDS 0.16; not HBM (L2 hit 0.61)",
          "bound_evidence": {"fractions": {"valu_issue": 0.64, "l2_hit": 0.61}, "inner_word_instructions": 18, "instructions_per_evaluation": 23.2,
                             "peak_candidates_per_s": ceil / 18 * 64, "useful_frac": 1.2e12 / 64 * 18 / ceil}}
    serial = {"ms_per_step": 240.0, "stage_ms": {"k_align": 70.0, "k_hctrl": 23.0, "order": 6.0, "scan": 137.0, "control_passes": 20.0}}
    args = type("A", (), {"steps": 2})()
    r = BN.roofline_block(dk, "x", 1.0, None, "none", None, 224.0, [448.0], (37000, 0), 4e12, [1.0] * 17, 1 << 23, args, None, serial, np)
    assert len(r["bound"]) <= 120 and "k_hscan_same" in r["bound"] and r["bound_long"].startswith("VALU issue")
    assert abs(r["peak"] - ceil / 18 * 64 / 1e9) < 1e-6 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and abs(r["frac"] - 0.511) < 0.01
    assert r["issue_utilisation"] == 0.64 and r["instructions_per_evaluation"] == 23.2 and r["inner_word_instructions"] == 18
    assert r["binding_kernel"] == "k_hscan_same" and r["serial_ms_k_hctrl"] == 23.0 and r["serial_ms_k_align"] == 70.0 and r["serial_control_passes"] == 20.0
    for k, v in r.items():   # what the record keeps: everything named above is a scalar
        if k in ("bound", "achieved", "peak", "frac", "issue_utilisation", "instructions_per_evaluation", "binding_kernel", "serial_ms_scan"):
            assert isinstance(v, (int, float, str))
    # a control-bound config names the control kernel
    serial["stage_ms"].update(k_hctrl=400.0)
    assert BN.roofline_block(dk, "x", 1.0, None, "none", None, 224.0, [448.0], (37000, 0), 4e12, [1.0] * 17, 1 << 23, args, None, serial, np)["binding_kernel"] == "k_hctrl"
