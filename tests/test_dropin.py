"""The signature-exact C++ boundary (dropin/compressionlib_dropin.cpp): CompressionLib's entry points with the reference's
exact parameter lists over libmpcodec.so.  tests/cpp/test_dropin.cpp is a caller written like Compression.cpp (its -c, -n and
-s modes, double names and ...Fast names); it is compiled against tests/cpp/refstub (declarations of the reference's interface:
test scaffolding, the GPU box has no /root/reference) together with the drop-in and must emit the oracle's bytes."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

EXE = os.path.join(ROOT, "tests", "cpp", "test_dropin")
REFERENCE = "/root/reference"


def _build():
    import imageexperiments_amd as ia
    lib = os.path.dirname(ia.library_path())
    srcs = [os.path.join(ROOT, "tests", "cpp", "test_dropin.cpp"), os.path.join(ROOT, "dropin", "compressionlib_dropin.cpp")]
    deps = srcs + [ia.library_path(), os.path.join(ROOT, "tests", "cpp", "refstub", "reference_interface.h")]
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(d) for d in deps):
        subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "tests", "cpp", "refstub", "CompressionLib", "inc"),
                        "-I", os.path.join(ROOT, "include")] + srcs + ["-o", EXE, "-L", lib, "-lmpcodec", f"-Wl,-rpath,{lib}",
                                                                         "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return EXE


def _run(mode, W, H, seed, quality, out, env=None):
    r = subprocess.run([_build(), mode, str(W), str(H), str(seed), quality, str(out)], capture_output=True, text=True, timeout=600,
                       env={**os.environ, **(env or {})})
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    return r.stdout


def test_dropin_compiles_and_links_against_the_reference_interface():
    """every symbol Compression.cpp needs from CompressionLib (SURVEY 8b's nm list) is defined by the drop-in with the
    reference's signature: the test caller links"""
    exe = _build()
    syms = subprocess.run(["nm", "-C", exe], capture_output=True, text=True).stdout
    for name in ("compressed::createCompressionContext(unsigned long, unsigned long, double)",
                 "compressed::createCompressionContextFast(unsigned long, unsigned long, double)",
                 "compressed::encodeImage(img::image<img::rgb> const*, unsigned long, unsigned long, double const*, double const*, double const*",
                 "compressed::encodeImageFast(img::image<img::rgb> const*, unsigned long, unsigned long, Eigen::VectorXf const&",
                 "compressed::decodeImage(unsigned char const*, unsigned long)", "compressed::decodeImageFast(unsigned char const*, unsigned long)",
                 "compressed::calculatePSNR(img::image<img::rgb> const*, img::image<img::rgb> const*)",
                 "matching::CalcMPDynamic(int, double const*, std::vector<matching::BasisChoice_t",
                 "matching::FromCoeffsDynamic(int, double const*"):
        assert name in syms, name


def test_dropin_builds_against_the_reference_s_own_value_types(tmp_path):
    """The drop-in and the Compression.cpp-shaped caller compiled against the reference's REAL SimpleMatrix/inc/{mathmatrix,
    mathvector}.h and ImageHelper/inc/image.h (they compile here unmodified; their member functions come from
    SimpleMatrix/src/{mathmatrix,mathvector}.cpp compiled where they lie) -- only CompressedImage.h / MatchingPursuit.h stay
    declarations, because they include Eigen (an empty submodule of the tree).  Host-only mode `h` of the caller: dictionary
    shapes, a dynamic dictionary with repeats, FromCoeffsDynamic, calculatePSNR through the real math::Matrix / math::Vector /
    img::image must print exactly what the build against the restated value types prints.  Build container only: nothing of
    /root/reference exists on the GPU box."""
    if not os.path.isdir(os.path.join(REFERENCE, "SimpleMatrix", "inc")):
        pytest.skip("the reference tree is not present (GPU box)")
    import imageexperiments_amd as ia
    lib = os.path.dirname(ia.library_path())
    real = str(tmp_path / "test_dropin_real")
    srcs = [os.path.join(ROOT, "tests", "cpp", "test_dropin.cpp"), os.path.join(ROOT, "dropin", "compressionlib_dropin.cpp"),
            os.path.join(REFERENCE, "SimpleMatrix", "src", "mathmatrix.cpp"), os.path.join(REFERENCE, "SimpleMatrix", "src", "mathvector.cpp")]
    subprocess.run(["g++", "-std=c++20", "-O1", "-ffp-contract=off", "-DMPC_TEST_REAL_REFERENCE_HEADERS", "-I", REFERENCE,
                    "-I", os.path.join(ROOT, "tests", "cpp", "refstub", "CompressionLib", "inc"), "-I", os.path.join(ROOT, "include")]
                   + srcs + ["-o", real, "-L", lib, "-lmpcodec", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    outs = []
    for exe in (real, _build()):
        for quality in ("3.5", "max"):
            out = tmp_path / (os.path.basename(exe) + quality + ".txt")
            r = subprocess.run([exe, "h", "96", "64", "31", quality, str(out)], capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, (exe, r.returncode, r.stdout, r.stderr)
            outs.append(out.read_text())
    assert outs[0] == outs[2] and outs[1] == outs[3]
    first = outs[0].split("\n")[0].split()
    assert first == ["32", "8", "510", "64", "510"]


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,quality", [(200, 136, "3.5"), (70, 50, "max"), (129, 77, "2.0")])
def test_compression_cpp_call_sequence_emits_the_oracles_bytes(tmp_path, oracle, W, H, quality):
    K = 32
    rgb = oracle.synth_frame(W, H, 4242)
    octx = oracle.OracleContext(K, 8, 0.0 if quality == "max" else float(quality))
    q = np.ones((3, K)) if quality == "max" else None
    want = bytes(octx.encode_image(rgb, quant=q))
    want_fast = bytes(oracle.OracleFastContext(octx).encode_image(rgb, quant=q))
    # double names; the ...Fast names Compression.cpp itself calls (float flavour: oracle/mpo_fast.c; parity unpinned against the
    # reference's Eigen results); the Fast names served from the double path on request
    for mode, env, expect in (("c", None, want), ("f", None, want_fast), ("f", {"MPC_FAST_EXACT": "1"}, want)):
        out = tmp_path / f"{mode}.mn"
        _run(mode, W, H, 4242, quality, out, env)
        assert out.read_bytes() == expect, (mode, env)


@pytest.mark.gpu
def test_compression_cpp_caller_on_two_lanes_emits_the_oracles_bytes(tmp_path, oracle):
    """MPC_DEVICES=0,0: the drop-in's encodeImage stripes the frame over two lanes (mpc_encode_images_multi, both on the one
    device of this box); the Compression.cpp-shaped caller still emits the oracle's bytes."""
    W, H, K = 200, 136, 32
    rgb = oracle.synth_frame(W, H, 4242)
    want = bytes(oracle.OracleContext(K, 8, 3.5).encode_image(rgb))
    out = tmp_path / "two_lanes.mn"
    _run("c", W, H, 4242, "3.5", out, {"MPC_DEVICES": "0,0"})
    assert out.read_bytes() == want


@pytest.mark.gpu
def test_round_trip_psnr_and_patch_records(tmp_path, oracle):
    W, H, K = 160, 96, 32
    rgb = oracle.synth_frame(W, H, 99)
    octx = oracle.OracleContext(K, 8, 3.5)
    blob = bytes(octx.encode_image(rgb))
    out = tmp_path / "decoded.rgb"
    text = _run("n", W, H, 99, "3.5", out)
    decoded = np.frombuffer(out.read_bytes(), np.uint8).reshape(H, W, 3)
    want = oracle.decode_image(blob)
    assert (decoded == want).all()
    psnr = float(text.split()[1])
    assert abs(psnr - oracle.lib().mpo_psnr(rgb.ctypes.data_as(oracle.C.POINTER(oracle.C.c_uint8)),
                                            want.ctypes.data_as(oracle.C.POINTER(oracle.C.c_uint8)), W, H)) < 1e-9
    assert int(text.split()[3]) == len(blob)
    # -s mode: CalcMPDynamic on random patches through the context's closures
    recs = tmp_path / "patches.txt"
    _run("s", W, H, 99, "3.5", recs)
    lines = recs.read_text().strip().split("\n")
    assert len(lines) == 64 * 3
    import random  # noqa: F401  (std::mt19937 below via numpy's bit generator)
    mt = np.random.MT19937()
    st = mt.state
    key = np.empty(624, np.uint32)
    key[0] = 99
    for i in range(1, 624):
        key[i] = (1812433253 * (int(key[i - 1]) ^ (int(key[i - 1]) >> 30)) + i) & 0xFFFFFFFF
    st["state"]["key"] = key
    st["state"]["pos"] = 624
    mt.state = st
    for p in range(4):                                         # the first patches suffice to pin the sequence
        x = int(mt.random_raw(1)[0]) % (W - 8)
        y = int(mt.random_raw(1)[0]) % (H - 8)
        for ch in range(3):
            patch = np.zeros(64)
            for offx in range(8):
                for offy in range(8):
                    r_, g_, b_ = (float(v) for v in rgb[y + offy, x + offx])
                    Y = 0.299 * r_ + 0.587 * g_ + 0.114 * b_
                    patch[offx + offy * 8] = Y if ch == 0 else ((0.436 / (1.0 - 0.114)) * (b_ - Y) if ch == 1 else (0.615 / (1.0 - 0.299)) * (r_ - Y))
            cnt, d, k, _, _ = octx.calc_mp(ch, patch)
            got = [int(v) for v in lines[3 * p + ch].split()]
            assert got[0] == cnt
            assert got[1::2] == [int(v) for v in d[:cnt]] and got[2::2] == [int(v) for v in k[:cnt]]


@pytest.mark.gpu
def test_foreign_closures_are_probed(tmp_path):
    _run("x", 64, 48, 7, "3.5", tmp_path / "x.txt")
    assert (tmp_path / "x.txt").read_text() == "ok"
