"""Builds and runs the C++ host-API parity test (tests/cpp/test_host_api.cpp)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "build", "test_host_api")


def test_cpp_host_api_compiles():
    subprocess.check_call(["make", "-C", ROOT, "-s", "build/test_host_api"])
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_cpp_host_api_matches_oracle():
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-C", ROOT, "-s", "build/test_host_api"])
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "cpp host api OK" in out.stdout
