"""The bit-plane form of the mismatch rule (bsx_dev.h: bsx_plane_word / _shift / _bmask / _mismatch), which the scan kernels of the
heavy pipeline evaluate on the GPU, checked on the host against the per-nt definition of the reference's rule
(align.h:167-200, param.h:125-147): totals and both early-out counts for every candidate position and read length."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_plane_rule_equals_per_nt_definition(tmp_path):
    exe = str(tmp_path / "planes_check")
    subprocess.check_call([HIPCC, "-O1", "-o", exe, os.path.join(ROOT, "tests", "harness", "planes_check.cpp")], stderr=subprocess.DEVNULL)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("ok"), out.stdout + out.stderr


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_scan_order_mapping_takes_every_block_exactly_once(tmp_path):
    """bsx_order_block (which block of the scan order a grid block takes: as dispatched, one contiguous eighth per XCD, or pieces dealt to
    the XCDs in turn) for every mode, order length and grid size, incl. grids that sweep: no block twice, none left out"""
    exe = str(tmp_path / "order_check")
    subprocess.check_call([HIPCC, "-O1", "-o", exe, os.path.join(ROOT, "tests", "harness", "order_check.cpp")], stderr=subprocess.DEVNULL)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("ok"), out.stdout + out.stderr
