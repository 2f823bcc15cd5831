"""Builds tests/cpp/test_mirror.cpp (the reference's Huffman/RLE test properties written against the C++ mirror
include/mpcodec.hpp) with g++ against libmpcodec.so and runs it: CPU part here, device round trip under -m gpu."""
import os
import subprocess

import pytest

from conftest import ROOT

EXE = os.path.join(ROOT, "tests", "cpp", "test_mirror")


def _build():
    import imageexperiments_amd as ia
    lib = os.path.dirname(ia.library_path())
    src = os.path.join(ROOT, "tests", "cpp", "test_mirror.cpp")
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(src), os.path.getmtime(ia.library_path())):
        subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), src, "-o", EXE,
                        "-L", lib, "-lmpcodec", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return EXE


def test_cpp_mirror_host_properties():
    exe = _build()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "8 tests, 0 failed" in r.stdout


@pytest.mark.gpu
def test_cpp_mirror_device_round_trip():
    exe = _build()
    r = subprocess.run([exe, "--gpu"], capture_output=True, text=True, timeout=600)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "1 tests, 0 failed" in r.stdout
