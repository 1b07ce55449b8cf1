import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the native pieces once per session if they are missing (hipcc cross-compiles on CPU)."""
    import subprocess
    if not os.path.exists(os.path.join(ROOT, "pearray_amd", "csrc", "libprgpu.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "pearray_amd", "csrc"), "-j8"])
    if not os.path.exists(os.path.join(ROOT, "oracle", "libpr_oracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libpr_oracle.so"])
