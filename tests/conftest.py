import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


os.environ.setdefault("BSX_POISON", "1")  # libbsx fills its heavy-pipeline pools with 0xA5 so that tests never rely on zeroed fresh memory


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_ffi
    oracle_ffi.build()
    return oracle_ffi
