"""the "-s" patch statistics of Compression.cpp:200-302 (how Data/stats.txt and the variance tables
were made).  The images behind Data/stats.txt are not in the reference tree, so the numbers themselves are "parity
unpinned" beyond the pursuit; what the fixture does pin is the report format (every line of it re-formats to itself)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN

@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the gpu-marked tests need a real MI355X (there is no CPU fallback)")
    return torch


LINE = re.compile(r"^(coeff|deltaId) (\d+) min (\S+) max (\S+) range (\S+) mean (\S+) variance (\S+) std dev (\S+)$")


def test_format_double_reproduces_every_number_of_the_golden_report():
    """std::format("{}", double): each of the ~1150 numbers in the reference's own Data/stats.txt (integers, long
    fractions, denormals in scientific notation) parses and prints back to the same text."""
    import imageexperiments_amd as ia
    n = 0
    for line in open(os.path.join(GOLDEN, "stats.txt")).read().splitlines():
        if line.endswith(" stats"):
            continue
        m = LINE.match(line)
        # one line of the fixture (Y coeff 11) was trimmed by hand to min/max/mean/variance; its numbers still count
        toks = m.groups()[2:] if m else re.findall(r"(?:min|max|mean|variance) (\S+)", line)
        assert len(toks) in (4, 6)
        for tok in toks:
            assert ia.format_double(float(tok)) == tok, (line, tok)
            n += 1
    assert n == 191 * 6 + 4


def test_golden_report_std_dev_is_sqrt_of_variance_and_range_is_max_minus_min():
    """The derived columns of the golden report follow Compression.cpp:277 (range = max - min where min was
    initialised, std dev = sqrt(sampleVariance)); pins how the report derives them."""
    import math
    import imageexperiments_amd as ia
    for line in open(os.path.join(GOLDEN, "stats.txt")).read().splitlines():
        m = LINE.match(line)
        if not m:
            continue
        lo, hi, rng, mean, var, sd = (float(t) for t in m.groups()[2:])
        assert ia.format_double(math.sqrt(var)) == m.group(8)
        assert ia.format_double(hi - lo) == m.group(5)


def test_special_values_print_like_std_format():
    import imageexperiments_amd as ia
    assert ia.format_double(0.0) == "0"
    assert ia.format_double(-0.0) == "-0"
    assert ia.format_double(4080.0) == "4080"
    assert ia.format_double(0.5) == "0.5"
    assert ia.format_double(1e-323) == "1e-323"
    assert ia.format_double(float("inf")) == "inf"


def test_patch_statistics_need_a_device():
    import imageexperiments_amd as ia
    ctx = ia.create_compression_context(8, 8, 0.0, device=-1)
    with pytest.raises(ia.MpcError) as e:
        ia.PatchStatistics(ctx, 1)
    assert e.value.status == ia.api.MPC_ERR_NO_DEVICE
    ctx.close()


def test_oracle_patch_statistics_are_welford_of_the_oracle_records(oracle):
    """The oracle's "-s" restatement against an independent numpy evaluation: same mt19937 draws (x then y), same
    records as mpo_calc_mp with quant 1.0 on the gathered patch, mean/variance equal numpy's to rounding."""
    K = 8
    ctx = oracle.OracleContext(K, 8, 0.0)
    rgb = oracle.synth_frame(96, 64, 5)
    st = oracle.OraclePatchStats(ctx, 42)
    st.add_image(rgb, 40)
    out = st.read()
    rs = np.random.RandomState()                      # MT19937 with init_genrand(seed) like std::mt19937::seed
    rs.seed(42)
    draws = rs.randint(0, 2 ** 32, size=80, dtype=np.uint64)
    vals = [[[[] for _ in range(K)] for _ in range(2)] for _ in range(3)]
    ones = np.ones(K)
    for p in range(40):
        x, y = int(draws[2 * p] % (96 - 8)), int(draws[2 * p + 1] % (64 - 8))
        yuv = np.zeros((3, 64))
        for dy in range(8):
            for dx in range(8):
                r, g, b = (int(v) for v in rgb[y + dy, x + dx])
                cy, cu, cv = C.c_double(), C.c_double(), C.c_double()
                oracle.lib().mpo_yuv_from_rgb(r, g, b, C.byref(cy), C.byref(cu), C.byref(cv))
                yuv[:, dx + 8 * dy] = cy.value, cu.value, cv.value
        for ch in range(3):
            cnt, d, k, _, _ = ctx.calc_mp(ch, yuv[ch], ones)
            for i in range(cnt):
                vals[ch][0][i].append(float(k[i]))
                vals[ch][1][i].append(float(d[i]))
    for ch in range(3):
        for kind in range(2):
            for i in range(K):
                v = np.array(vals[ch][kind][i])
                N, lo, hi, mean, ss = out[ch, kind, i]
                assert N == len(v)
                if len(v):
                    assert lo == v.min() and hi == v.max()
                    assert abs(mean - v.mean()) <= 1e-9 * max(1.0, abs(v.mean()))
                    assert abs(ss - ((v - v.mean()) ** 2).sum()) <= 1e-6 * max(1.0, ss)
    st.close()


@pytest.mark.gpu
def test_patch_statistics_equal_oracle_bit_for_bit(gpu, oracle):
    """Product (patches through the device tile encoder, Welford on the host) == oracle, every double identical,
    over two images of different sizes sharing one generator as the reference's file loop does."""
    import imageexperiments_amd as ia
    K = 32
    ctx = ia.create_compression_context(K, 8, 0.0, device=0)
    octx = oracle.OracleContext(K, 8, 0.0)
    a, b = ia.PatchStatistics(ctx, 7), oracle.OraclePatchStats(octx, 7)
    for (W, H, seed, n) in ((200, 120, 3, 150), (64, 333, 4, 90), (7, 50, 5, 10)):      # the last one is skipped (:233)
        rgb = oracle.synth_frame(W, H, seed)
        a.add_image(rgb, n)
        b.add_image(rgb, n)
    x, y = a.read(), b.read()
    assert x.tobytes() == y.tobytes()
    assert x[0, 0, 0, 0] == 240.0                     # every patch has a first Y step with quant 1.0
    text = a.report().splitlines()
    assert len(text) == 6 * (K + 1) and text[0] == "Y coeff stats" and text[K + 1] == "Y basisId stats"
    m = LINE.match(text[1])
    assert m and float(m.group(6)) == x[0, 0, 0, 3] and float(m.group(4)) == x[0, 0, 0, 2]
    a.close(); b.close(); ctx.close()


@pytest.mark.gpu
def test_patch_statistics_reject_one_block_images(gpu):
    import imageexperiments_amd as ia
    ctx = ia.create_compression_context(8, 8, 0.0, device=0)
    st = ia.PatchStatistics(ctx, 1)
    with pytest.raises(ia.MpcError):
        st.add_image(np.zeros((8, 40, 3), np.uint8), 4)
    st.close(); ctx.close()
