"""bsmap_amd/csrc/bsx_textout.h (text output through a shared mapping of the file): the file holds exactly the bytes a sequential
writer would have written — header by pwrite, then batches of pieces (some empty, some of several megabytes) copied by 1 to 12
threads at page-unaligned offsets — on the default temp file system and on /dev/shm (tmpfs), where the command line uses it."""
import os
import subprocess

import pytest

from conftest import HOST_SAN_FLAGS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("to") / "textout_check")
    subprocess.run(["g++"] + HOST_SAN_FLAGS + ["-o", exe, os.path.join(ROOT, "tests", "harness", "textout_check.cpp")], check=True)   # (ASan + UBSan: conftest.py)
    return exe


@pytest.mark.parametrize("where", ["tmp", "shm"])
@pytest.mark.parametrize("seed,threads,batches", [(1, 1, 5), (2, 4, 9), (3, 12, 12)])
def test_mapped_output_equals_sequential_writer(where, seed, threads, batches, harness, tmp_path):
    if where == "shm":
        if not os.path.isdir("/dev/shm") or not os.access("/dev/shm", os.W_OK):
            pytest.skip("no writable /dev/shm")
        out = f"/dev/shm/bsx_textout_test_{os.getpid()}_{seed}"
    else:
        out = str(tmp_path / "out.txt")
    try:
        res = subprocess.run([harness, out, str(seed), str(threads), str(batches)], capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stderr      # (the harness reads the file back and compares it with the sequential text)
        assert os.path.getsize(out) == int(res.stdout) > 100000
        assert open(out, "rb").read(11) == b"@HD\tVN:1.0\n"
    finally:
        if os.path.exists(out):
            os.remove(out)
