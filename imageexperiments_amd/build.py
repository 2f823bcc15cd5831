"""Builds imageexperiments_amd/lib/libmpcodec.so for gfx950 with hipcc (in-tree, so the .so travels with gpurun).

    python -m imageexperiments_amd.build [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libmpcodec.so")
SOURCES = ["host_dictionary.cpp", "host_bitstream.cpp", "host_codec.cpp", "mp_kernels.hip", "mpcodec_capi.cpp"]
HEADERS = ["host_dictionary.h", "host_bitstream.h", "host_codec.h", "mp_device.h", os.path.join("..", "..", "include", "mpcodec.h")]
# -ffp-contract=off: host and device must round every mul and add separately (the reference is built
# with MSVC /fp:precise and the integer outputs depend on it).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC",
         "-Wall", "-Wno-unused-result"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS if os.path.exists(os.path.join(CSRC, s))]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    """Compile every HIP/C++ source of the product into one shared library. Returns its path."""
    if not force and not _stale():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = [hipcc()] + FLAGS + ["-shared", "-o", LIB] + srcs
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
