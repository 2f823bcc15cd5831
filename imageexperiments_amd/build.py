"""Builds imageexperiments_amd/lib/libmpcodec.so for gfx950 with hipcc (in-tree, so the .so travels with gpurun).

    python -m imageexperiments_amd.build [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libmpcodec.so")
SOURCES = ["host_dictionary.cpp", "host_bitstream.cpp", "host_codec.cpp", "host_stats.cpp", "mp_kernels.hip", "mp_pursuit.hip", "mp_streams.hip", "mp_entropy.hip", "mpcodec_capi.cpp", "mpcodec_multi.cpp"]
HEADERS = ["host_dictionary.h", "host_bitstream.h", "host_codec.h", "host_stats.h", "mp_device.h", "mpc_internal.h", os.path.join("..", "..", "include", "mpcodec.h")]
# -ffp-contract=off: host and device must round every mul and add separately (the reference is built
# with MSVC /fp:precise and the integer outputs depend on it).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-pthread",
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
    verify_scalar_sweeps(LIB)
    return LIB


def verify_scalar_sweeps(lib=LIB):
    """Disassemble the gfx950 code object inside the library and check that every multiply of the sweep kernels
    takes its dictionary operand from an SGPR (s_load-fed).  If the compiler ever decides the row pointer is
    divergent it silently emits vector loads instead -- same results, ~6x slower -- so the build refuses that."""
    import re
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        return None
    data = open(lib, "rb").read()
    starts = [m.start() for m in re.finditer(b"\x7fELF", data)]
    report = {}
    for k, st in enumerate(starts[1:]):
        tmp = os.path.join(os.path.dirname(lib), f".co{k}.elf")
        with open(tmp, "wb") as f:
            f.write(data[st:])
        out = subprocess.run([objdump, "-d", tmp], capture_output=True, text=True).stdout
        os.remove(tmp)
        for name in ("mp_base_kernel", "mp_detail_kernel"):  # the exact sweeps (the filter kernel has no scalar-fed loop)
            m = re.search(name + r"[^\n]*>:(.*?)s_endpgm", out, re.S)
            if not m:
                continue
            muls = re.findall(r"v_mul_f64[^\n]*", m.group(1))
            scalar = [x for x in muls if re.search(r", s\[\d+:\d+\]", x)]
            report[name] = (len(scalar), len(muls))
    bad = {k: v for k, v in report.items() if v[0] != v[1] or v[1] == 0}
    if bad or not report:
        raise RuntimeError(f"sweep kernels lost their scalar dictionary operands: {report}")
    return report


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
