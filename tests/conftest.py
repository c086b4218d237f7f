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
    # without a GPU).  A failed build fails the session: a stale .so from before the change must never make the suite green.
    import subprocess
    for sub in (os.path.join("draco-oxide_amd", "csrc"), "oracle"):
        try:
            r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, sub)], timeout=1800, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        except FileNotFoundError:
            continue   # no `make` on this machine: the libraries' own "missing, build first" errors speak for themselves
        if r.returncode != 0:
            raise pytest.UsageError(f"`make -C {sub}` failed (exit {r.returncode}); refusing to test against a stale library.  Build log tail:\n" + r.stdout[-3000:])


@pytest.fixture(scope="session")
def data_dir():
    return DATA
