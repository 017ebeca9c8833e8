import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# Host-side C++ under test (the harnesses of tests/harness/ over csrc/bsx_reads.h, bsx_textout.h, bsx_bam_out.h, bsx_lanes.h; the command line's own
# sanitizer build, Makefile `asan`) is compiled with AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers on the host; never on the GPU box's
# device code).  A report aborts the harness, which fails the test.
HOST_SAN_FLAGS = ["-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer"]
os.environ.setdefault("ASAN_OPTIONS", "detect_leaks=1:abort_on_error=1")
os.environ.setdefault("UBSAN_OPTIONS", "print_stacktrace=1:halt_on_error=1")

os.environ.setdefault("BSX_POISON", "1")  # libbsx fills its heavy-pipeline pools with 0xA5 so that tests never rely on zeroed fresh memory


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_ffi
    oracle_ffi.build()
    return oracle_ffi
