"""Register / spill budget of the hot kernels (CPU-side: hipcc cross-compiles gfx950 and reports the resource usage).
The main kernel's per-unit code is inlined into one 128-VGPR function; twice in round 2 an innocent-looking edit in a rarely
taken branch (an integer division, a second binary search) pushed it from 256 B to 1.3 KB of spill per lane — 49 -> 67 ms per
2^20 pairs — and only a profile showed it.  This test fails the build instead."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "bsmap_amd", "csrc", "bsx_align.hip")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

# kernel (mangled-name fragment) -> (max VGPRs, max scratch bytes per lane)
BUDGET = {
    "7k_alignILb1ELb0ELb0EE": (96, 128),    # paired-end WGBS main kernel (the headline config) with the work counters (the counted pass only): five waves per SIMD, 104 B today
    "7k_alignILb0ELb0ELb0EE": (96, 128),    # single-end
    "7k_alignILb1ELb0ELb1EE": (96, 160),    # the same with the context prefilter (counters off: what the command line and the bench's timed region run), 160 B today: 35 spilled VGPRs — edits that took it to 50-67 cost 6-10 ms per 2^22 pairs
    "7k_alignILb0ELb0ELb1EE": (96, 128),
    "7k_alignILb1ELb1ELb1EE": (96, 160),    # exact mode on top of the context prefilter (round 6: BSX_P1_EXACT no longer falls back to the plain scan), 144 B today
    # the scan kernels of the heavy pipeline, without (ILb0E: what the command line and the bench's timed region run) and with the work counters
    "7k_hscanILb0EE": (80, 0), "7k_hscanILb1EE": (80, 0),                    # one task per wave: six waves per SIMD (read words and masks live in VGPRs)
    "12k_hscan_sameILb0EE": (128, 0), "12k_hscan_sameILb1EE": (128, 0),      # WGBS (groups of tasks over one window and read offset): four chunks per step, four waves per SIMD, no scratch
    "14k_hscan_sharedILb0EE": (96, 0), "14k_hscan_sharedILb1EE": (96, 0),    # RRBS
    # the control kernel of the heavy pipeline runs one block per CU BESIDE the scan kernels of the other batches in flight: what its waves hold of their SIMD's 512
    # registers decides how many scan waves fit next to them.  Single-end (C4, C2): 224 leaves room for three 96-register scan waves; round 6 saw an edit lift it
    # to 256 — two scan waves — and C4 go from 415 to 464 ms per step through code that never ran there.
    "7k_hctrlILb0EE": (224, 128), "7k_hctrlILb1EE": (256, 128),
}


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_hot_kernels_stay_within_their_register_budget(tmp_path):
    res = subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
                          "-c", SRC, "-o", str(tmp_path / "a.o")], capture_output=True, text=True, timeout=1200, cwd=os.path.dirname(SRC))
    assert res.returncode == 0, res.stderr[-2000:]
    usage, name = {}, None
    for line in res.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            usage[name] = {}
            continue
        m = re.search(r"remark:\s+(VGPRs|ScratchSize \[bytes/lane\]): (\d+)", line)
        if m and name:
            usage[name][m.group(1).split()[0]] = int(m.group(2))
    for frag, (vg, sc) in BUDGET.items():
        hits = [u for n, u in usage.items() if frag in n]
        assert len(hits) == 1, (frag, list(usage))
        assert hits[0]["VGPRs"] <= vg and hits[0]["ScratchSize"] <= sc, (frag, hits[0], "budget", vg, sc)
