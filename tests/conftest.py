import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

DATA = os.path.join(ROOT, "tests", "golden", "data")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the in-tree libraries are build products (git-ignored): bring them up to date — a no-op when they are (hipcc cross-compiles
    # without a GPU).  A failed build surfaces as the libraries' own "missing, build first" errors in the tests that need them.
    import subprocess
    for sub in (os.path.join("draco-oxide_amd", "csrc"), "oracle"):
        try:
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, sub)], check=False, timeout=1800, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        except Exception:
            pass


@pytest.fixture(scope="session")
def data_dir():
    return DATA
