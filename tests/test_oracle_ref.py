"""Oracle (plain C) vs the REFERENCE's own object code (oracle/_ref/libref.so =
SimpleMatrix + ImageHelper/misc compiled in place from /root/reference).
Everything here must agree bit-for-bit."""
import ctypes as C

import numpy as np
import pytest


@pytest.fixture(scope="module")
def ref(oracle):
    r = oracle.ref()
    if r is None:
        pytest.skip("oracle/_ref/libref.so not built (needs /root/reference)")
    return r


def _bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


@pytest.mark.parametrize("n", [1, 2, 3, 5, 8, 13, 31, 32, 47, 63, 64])
def test_symmetric_eigensolver_bit_exact(oracle, ref, n):
    """symmeigen.cpp:34-244 vs mpo_symm_eigen on model covariances and random symmetric input."""
    L = oracle.lib()
    rng = np.random.default_rng(n)
    pts = [(x, y) for x in range(8) for y in range(8)][:n]
    cases = []
    for ch in range(3):
        cases.append(np.array([[L.mpo_cov_model(ch, float(a[0] - b[0]), float(a[1] - b[1])) for b in pts]
                               for a in pts]))
    m = rng.standard_normal((n, n))
    cases.append(m + m.T)
    cases.append(np.diag(rng.standard_normal(n)))        # already diagonal: scale == 0 branch
    for A in cases:
        A = np.ascontiguousarray(A)
        v1, e1, v2, e2 = np.zeros((n, n)), np.zeros(n), np.zeros((n, n)), np.zeros(n)
        L.mpo_symm_eigen(oracle._dp(A), n, oracle._dp(v1), oracle._dp(e1))
        ref.ref_symm_eigen(oracle._dp(A), n, oracle._dp(v2), oracle._dp(e2))
        assert (_bits(v1) == _bits(v2)).all()
        assert (_bits(e1) == _bits(e2)).all()


def test_gemv_row_order_bit_exact(oracle, ref, octx32):
    """math::Multiply(Matrix,Vector) mathmatrix.cpp:426 == sequential tot += l*r (the oracle's row_dot),
    checked through mpo_calc_mp's first selection on real dictionary rows."""
    rng = np.random.default_rng(7)
    D = np.ascontiguousarray(np.vstack([octx32.base, octx32.det[0][:63]]))
    for _ in range(20):
        r = np.ascontiguousarray(rng.standard_normal(64) * rng.choice([1.0, 30.0, 500.0]))
        p = np.zeros(D.shape[0])
        ref.ref_multiply(oracle._dp(D), D.shape[0], 64, oracle._dp(r), oracle._dp(p))
        # reference Select (MatchingPursuit.cpp:7-25) on the reference's projections
        best, idx = 0.0, -1
        for i in range(510):
            if abs(p[i]) > abs(best):
                best, idx = p[i], i
        cnt, d, k, res, S = octx32.calc_mp(0, r, quant=np.ones(32))
        assert d[0] == idx
        q = int(np.sign(best) * np.floor(abs(best) + 0.5))            # round half away, quant = 1
        assert k[0] == ((q << 1) ^ (q >> 31)) & 0xFFFF


def test_scale_subtract_bit_exact(oracle, ref, octx32):
    """residual update MatchingPursuit.cpp:70-71 via Vector::Scale/Subtract: one full MP step."""
    rng = np.random.default_rng(11)
    for _ in range(20):
        r = np.ascontiguousarray(rng.standard_normal(64) * 200.0)
        q = np.full(32, 1.0e9)
        q[0] = 3.0                                   # step 1 quantises to 0 -> count == 1, residual after 1 update
        cnt, d, k, res, S = octx32.calc_mp(0, r, quant=q)
        if cnt != 1:
            continue
        coeff = 3.0 * float(oracle.lib().mpo_zigzag_dec(int(k[0])))
        rr = r.copy()
        ref.ref_scale_subtract(oracle._dp(rr), oracle._dp(np.ascontiguousarray(octx32.base[d[0]])), coeff, 64)
        assert (_bits(rr) == _bits(res)).all()


def test_yuv_rgb_bit_exact(oracle, ref):
    """img::YUVFromRGB / RGBFromYUV misc.cpp:7-36."""
    L = oracle.lib()
    rng = np.random.default_rng(3)
    y = np.zeros(3)
    a, b, c = C.c_double(), C.c_double(), C.c_double()
    cols = rng.integers(0, 256, (5000, 3)).tolist() + [[0, 0, 0], [255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255]]
    for r8, g8, b8 in cols:
        L.mpo_yuv_from_rgb(r8, g8, b8, C.byref(a), C.byref(b), C.byref(c))
        ref.ref_yuv_from_rgb(r8, g8, b8, oracle._dp(y))
        assert (a.value, b.value, c.value) == tuple(y)
    out1 = (C.c_uint8 * 3)()
    out2 = np.zeros(3, np.uint8)
    for yy, uu, vv in (rng.standard_normal((5000, 3)) * [120, 80, 80] + [128, 0, 0]).tolist():
        L.mpo_rgb_from_yuv(yy, uu, vv, C.cast(C.byref(out1, 0), C.POINTER(C.c_uint8)),
                           C.cast(C.byref(out1, 1), C.POINTER(C.c_uint8)),
                           C.cast(C.byref(out1, 2), C.POINTER(C.c_uint8)))
        ref.ref_rgb_from_yuv(yy, uu, vv, out2.ctypes.data_as(C.POINTER(C.c_uint8)))
        assert tuple(out1) == tuple(out2)
