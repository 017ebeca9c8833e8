"""The reference-side binding INTEGRATION.md shows a maintainer (bsx_bridge.cpp) must at least be valid C++ against the
reference's own headers and include/bsx.h: field names of Param / SingleAlign / ReadInf, the s_OutHit signature, every
bsx_* call.  Container-only (needs /root/reference and g++); the stub is extracted from INTEGRATION.md itself so the
document is what gets checked."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REF) or shutil.which("g++") is None, reason="needs the reference tree and g++ (build container only)")
def test_integration_bridge_compiles_against_reference_headers(tmp_path):
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```cpp\n(.*?)```", text, re.S)
    assert blocks and "bsx_bridge.cpp" in blocks[0]
    src = tmp_path / "bsx_bridge.cpp"
    src.write_text(blocks[0])
    # the flags oracle/Makefile builds the reference with, syntax check only
    cmd = ["g++", "-std=gnu++98", "-include", "unistd.h", "-include", "time.h", "-DMAXHITS=1000", "-DTHREAD", "-DREAD_144", "-fsyntax-only",
           "-Wall", "-I", REF, "-I", os.path.join(REF, "samtools"), "-I", os.path.join(ROOT, "include"), str(src)]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr[-3000:]
    assert "error" not in res.stderr
