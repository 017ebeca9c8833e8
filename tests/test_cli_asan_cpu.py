"""The command line's host code under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers on the host side; `make -C bsmap_amd/csrc asan`).
No GPU here: the runs below end where the reference would be uploaded — behind the option parser, the file checks, the lane planner (`--lanes`: the parent
counts records and forks without ever touching a device) and the pinned-buffer fallbacks.  A sanitizer report aborts the binary; the test also greps for one."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bsmap_amd", "csrc")
EXE = os.path.join(ROOT, "bsmap_amd", "bsmap_asan")

pytestmark = pytest.mark.skipif(shutil.which("g++") is None or not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs g++ and hipcc (libbsx.so is linked)")


@pytest.fixture(scope="module")
def exe():
    subprocess.run(["make", "-C", CSRC, "asan"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=1800)
    return EXE


def _run(exe, args, cwd):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=99", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=98", BSX_PIN="0")
    r = subprocess.run([exe] + args, capture_output=True, text=True, cwd=cwd, env=env, timeout=300)
    assert "AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr and r.returncode not in (98, 99), (args, r.returncode, r.stderr[-1500:])
    return r


def _inputs(tmp_path, n=300):
    import random
    rng = random.Random(5)
    g = "".join(rng.choice("ACGT") for _ in range(20000))
    (tmp_path / "g.fa").write_text(">chr1 test\n" + "\n".join(g[i:i + 60] for i in range(0, len(g), 60)) + "\n>chr2\n" + g[500:3000] + "\n")
    with open(tmp_path / "a.fq", "w") as fa, open(tmp_path / "b.fq", "w") as fb:
        for i in range(n):
            p = rng.randrange(0, len(g) - 400)
            fa.write("@r%d/1\n%s\n+\n%s\n" % (i, g[p:p + 100], "I" * 100))
            fb.write("@r%d/2\n%s\n+\n%s\n" % (i, g[p + 150:p + 250][::-1], "I" * 100))
    return str(tmp_path / "g.fa"), str(tmp_path / "a.fq"), str(tmp_path / "b.fq")


def test_usage_and_option_errors(exe, tmp_path):
    r = _run(exe, [], str(tmp_path))
    assert "Usage:" in r.stdout + r.stderr
    for args in (["-a"], ["-v", "99", "-a", "x", "-d", "y"], ["-a", "/nonexistent/reads.fq", "-d", "/nonexistent/ref.fa", "-o", "o.sam"], ["-Z"], ["-D", "C-CGG", "-s", "99", "-a", "x", "-d", "y"],
                 ["-w", "5000", "-a", "x", "-d", "y"], ["-A", "ACGT", "-A", "GGGG", "-q", "20", "-a", "x", "-d", "y"]):
        _run(exe, args, str(tmp_path))


def test_whole_front_end_up_to_the_device(exe, tmp_path):
    """real inputs: option parsing, output-file set-up, reference path, then the first libbsx call reports that there is no device"""
    g, a, b = _inputs(tmp_path)
    for extra in ([], ["-b", b, "-m", "28", "-x", "500"], ["-b", b, "-o", str(tmp_path / "o.bam")], ["-2", str(tmp_path / "u.bsp")], ["-B", "10", "-E", "120"], ["-G", "0", "-p", "3"]):
        r = _run(exe, ["-a", a, "-d", g, "-o", str(tmp_path / "o.sam"), "-v", "4", "-s", "16"] + extra, str(tmp_path))
        txt = (r.stdout + r.stderr).lower()
        assert r.returncode == 1 and ("device" in txt or "host buffer" in txt), (r.returncode, txt[-400:])   # (page-locked buffers need the device, too: whichever thread gets there first reports)


def test_lane_planner_in_the_parent(exe, tmp_path):
    """--lanes: the parent counts the records of both files in parallel, cuts them into ranges and forks the lanes — all host code (csrc/bsx_lanes.h);
    the lanes themselves then fail for want of a device"""
    g, a, b = _inputs(tmp_path, n=1000)
    for extra in (["--lanes=3"], ["--lanes=2", "--lane-files"], ["--lanes=4", "-B", "100", "-E", "900"]):
        _run(exe, ["-a", a, "-b", b, "-d", g, "-o", str(tmp_path / "o.sam"), "-m", "28", "-x", "500"] + extra, str(tmp_path))
