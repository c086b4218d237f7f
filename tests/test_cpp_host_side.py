"""The C++ host side of the boundary (include/draco_mi.hpp: draco_oxide::core::MeshBuilder, encode::Config, encode::encode —
the reference's names and semantics above the C ABI) through tests written like the reference's own (tests/cpp)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")
EXE = os.path.join(CPP, "reference_style_tests")


def _build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    subprocess.check_call(["make", "-s", "-C", CPP])


def test_cpp_builder_tests_on_the_host():
    _build()
    out = subprocess.run([EXE, "--host"], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "all checks passed" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_cpp_encode_tests_against_the_oracle():
    _build()
    out = subprocess.run([EXE, "--all", os.path.join(ROOT, "tests", "golden", "data")], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "all checks passed" in out.stdout, out.stdout + out.stderr
