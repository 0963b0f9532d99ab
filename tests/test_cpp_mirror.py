"""Builds and runs the C++ mirror tests (tests/cpp/test_mirror.cpp over include/sumcheck_amd.hpp)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_mirror.cpp")
BIN = os.path.join(ROOT, "tests", "cpp", "test_mirror.bin")
LIBDIR = os.path.join(ROOT, "sumcheck_amd")


def build_cpp():
    cmd = ["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), SRC, "-o", BIN, "-L", LIBDIR, "-lsumcheck_hip",
           f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return BIN


def test_cpp_mirror_compiles_against_the_header():
    """CPU: the header-only mirror and the test program compile and link against the C ABI"""
    assert os.path.exists(build_cpp())


@pytest.mark.gpu
def test_cpp_mirror_reference_tests():
    if not os.path.exists(BIN):
        build_cpp()
    out = subprocess.run([BIN], capture_output=True, text=True, timeout=600)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ALL TESTS PASSED" in out.stdout
