"""CPU hardening (verdict r1 item 8): the product's host code under AddressSanitizer + UBSan.  `make asan-host` builds
tests/cpp/asan_host.cpp (the host sources in one translation unit) with g++ -fsanitize=address,undefined and drives the
dictionary builder, the entropy coders and the container parser through round trips and a corpus of truncated / bit-flipped
containers, the reference's own .mn among them.  Any sanitizer report aborts the run.  (`make asan` adds the oracle's
golden-fixture tests with the oracle itself built under the sanitizers.)"""
import subprocess

from conftest import ROOT


def test_host_code_is_clean_under_asan_and_ubsan():
    r = subprocess.run(["make", "-s", "asan-host"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "asan_host: 0 failed" in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
