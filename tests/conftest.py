import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_oracle():
    """The oracle is compiled here when a compiler is around; on the GPU box the
    prebuilt oracle/_ref/liboracle.so travels with the tree."""
    so = os.path.join(ROOT, "oracle", "_ref", "liboracle.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("crc32c_oracle.c", "wal_oracle.c")]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    yield
