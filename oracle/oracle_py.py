"""ctypes binding of oracle/_build/liboracle.so and oracle/_ref/libref.so.

ORACLE = test infrastructure, NOT product code.  Only tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke() may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
MAX_K = 32


def build(ref=True):
    """(Re)build the C restatement; build _ref only where /root/reference exists."""
    subprocess.run(["make", "-s", "-C", HERE], check=True)
    if ref and os.path.isdir("/root/reference/SimpleMatrix/src"):
        subprocess.run(["make", "-s", "-C", HERE, "ref"], check=True)


class Line(C.Structure):
    _fields_ = [("ax", C.c_int), ("ay", C.c_int), ("bx", C.c_int), ("by", C.c_int)]


class Ctx(C.Structure):
    _fields_ = [("K", C.c_int), ("bs", C.c_int), ("N", C.c_int), ("nbase", C.c_int),
                ("lines", C.POINTER(Line)), ("base", C.POINTER(C.c_double)),
                ("det_rows", C.POINTER(C.c_int)), ("det_off", C.POINTER(C.c_size_t)),
                ("det", C.POINTER(C.c_double) * 3), ("quant", (C.c_double * MAX_K) * 3)]


class U16V(C.Structure):
    _fields_ = [("d", C.POINTER(C.c_uint16)), ("n", C.c_size_t), ("cap", C.c_size_t)]


class Streams(C.Structure):
    _fields_ = [("W", C.c_int), ("H", C.c_int), ("K", C.c_int), ("bs", C.c_int),
                ("quant", (C.c_uint16 * MAX_K) * 3), ("lengths", U16V),
                ("codes", U16V * (6 * MAX_K))]


class Bits(C.Structure):
    _fields_ = [("w", C.POINTER(C.c_uint64)), ("cap", C.c_size_t), ("wword", C.c_size_t),
                ("wbit", C.c_size_t), ("rword", C.c_size_t), ("rbit", C.c_size_t)]


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        path = os.environ.get("ORACLE_LIB") or os.path.join(HERE, "_build", "liboracle.so")     # ORACLE_LIB: `make asan-oracle`
        if not os.path.exists(path):
            build(ref=False)
        L = C.CDLL(path)
        dp = C.POINTER(C.c_double)
        u8p = C.POINTER(C.c_uint8)
        u16p = C.POINTER(C.c_uint16)
        L.mpo_ctx_create.restype = C.POINTER(Ctx)
        L.mpo_ctx_create.argtypes = [C.c_int, C.c_int, C.c_double]
        L.mpo_ctx_destroy.argtypes = [C.POINTER(Ctx)]
        L.mpo_symm_eigen.argtypes = [dp, C.c_int, dp, dp]
        L.mpo_create_basis.argtypes = [dp, C.c_int, dp]
        L.mpo_quant_tables.argtypes = [C.c_int, C.c_int, C.c_double, dp, dp, dp]
        L.mpo_cov_model.restype = C.c_double
        L.mpo_cov_model.argtypes = [C.c_int, C.c_double, C.c_double]
        L.mpo_calc_mp.restype = C.c_int
        L.mpo_calc_mp.argtypes = [C.POINTER(Ctx), C.c_int, dp, dp, u16p, u16p, dp, C.POINTER(C.c_uint32)]
        L.mpo_from_coeffs.argtypes = [C.POINTER(Ctx), C.c_int, dp, C.c_int, u16p, u16p, dp]
        L.mpo_gather_tile.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp]
        L.mpo_encode_tiles.argtypes = [C.POINTER(Ctx), u8p, C.c_int, C.c_int, dp, dp, dp, C.c_int, C.c_int,
                                       u16p, u16p, u16p, dp, C.POINTER(C.c_uint32)]
        L.mpo_encode_image.restype = C.POINTER(C.c_uint8)
        L.mpo_encode_image.argtypes = [C.POINTER(Ctx), u8p, C.c_int, C.c_int, dp, dp, dp, C.POINTER(C.c_size_t)]
        L.mpo_decode_image.restype = C.c_int
        L.mpo_decode_image.argtypes = [u8p, C.c_size_t, C.POINTER(u8p), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        fp = C.POINTER(C.c_float)
        L.mpo_fast_create.restype = C.c_void_p
        L.mpo_fast_create.argtypes = [C.POINTER(Ctx)]
        L.mpo_fast_destroy.argtypes = [C.c_void_p]
        L.mpo_calc_mp_fast.restype = C.c_int
        L.mpo_calc_mp_fast.argtypes = [C.c_void_p, C.c_int, fp, fp, u16p, u16p, fp, C.POINTER(C.c_uint32)]
        L.mpo_from_coeffs_fast.argtypes = [C.c_void_p, C.c_int, fp, C.c_int, u16p, u16p, fp]
        L.mpo_encode_tiles_fast.argtypes = [C.c_void_p, u8p, C.c_int, C.c_int, dp, dp, dp, C.c_int, C.c_int,
                                            u16p, u16p, u16p, dp, C.POINTER(C.c_uint32)]
        L.mpo_encode_image_fast.restype = C.POINTER(C.c_uint8)
        L.mpo_encode_image_fast.argtypes = [C.c_void_p, u8p, C.c_int, C.c_int, dp, dp, dp, C.POINTER(C.c_size_t)]
        L.mpo_decode_image_fast.restype = C.c_int
        L.mpo_decode_image_fast.argtypes = [u8p, C.c_size_t, C.POINTER(u8p), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.mpo_read_compressed.restype = C.c_int
        L.mpo_read_compressed.argtypes = [u8p, C.c_size_t, C.POINTER(Streams)]
        L.mpo_write_compressed.restype = C.POINTER(C.c_uint8)
        L.mpo_write_compressed.argtypes = [C.POINTER(Streams), C.POINTER(C.c_size_t)]
        L.mpo_streams_free.argtypes = [C.POINTER(Streams)]
        L.mpo_u16v_push.argtypes = [C.POINTER(U16V), C.c_uint16]
        L.mpo_psnr.restype = C.c_double
        L.mpo_psnr.argtypes = [u8p, u8p, C.c_int, C.c_int]
        L.mpo_synth_frame.argtypes = [u8p, C.c_int, C.c_int, C.c_uint32]
        L.mpo_patch_stats_create.restype = C.c_void_p
        L.mpo_patch_stats_create.argtypes = [C.c_int, C.c_uint32]
        L.mpo_patch_stats_destroy.argtypes = [C.c_void_p]
        L.mpo_patch_stats_add_image.argtypes = [C.c_void_p, C.POINTER(Ctx), u8p, C.c_int, C.c_int, C.c_int]
        L.mpo_patch_stats_read.argtypes = [C.c_void_p, dp]
        L.mpo_yuv_from_rgb.argtypes = [C.c_uint8, C.c_uint8, C.c_uint8, dp, dp, dp]
        L.mpo_rgb_from_yuv.argtypes = [C.c_double, C.c_double, C.c_double, u8p, u8p, u8p]
        L.mpo_bits_init.argtypes = [C.POINTER(Bits)]
        L.mpo_bits_free.argtypes = [C.POINTER(Bits)]
        L.mpo_bits_write.argtypes = [C.POINTER(Bits), C.c_uint64, C.c_int]
        L.mpo_bits_read.restype = C.c_uint64
        L.mpo_bits_read.argtypes = [C.POINTER(Bits), C.c_int]
        L.mpo_bits_peek.restype = C.c_uint64
        L.mpo_bits_peek.argtypes = [C.POINTER(Bits), C.c_int]
        L.mpo_bits_skip.argtypes = [C.POINTER(Bits), C.c_int]
        L.mpo_bits_size.restype = C.c_size_t
        L.mpo_bits_size.argtypes = [C.POINTER(Bits)]
        L.mpo_bits_remaining.restype = C.c_size_t
        L.mpo_bits_remaining.argtypes = [C.POINTER(Bits)]
        L.mpo_bits_append.argtypes = [C.POINTER(Bits), C.POINTER(Bits)]
        L.mpo_bits_save.restype = C.POINTER(C.c_uint8)
        L.mpo_bits_save.argtypes = [C.POINTER(Bits), C.POINTER(C.c_size_t)]
        L.mpo_bits_load.argtypes = [C.POINTER(Bits), u8p, C.c_size_t, C.c_size_t]
        L.mpo_zigzag_enc.restype = C.c_uint32
        L.mpo_zigzag_enc.argtypes = [C.c_int32]
        L.mpo_zigzag_dec.restype = C.c_int32
        L.mpo_zigzag_dec.argtypes = [C.c_uint32]
        L.mpo_golomb_write.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(Bits)]
        L.mpo_golomb_read.restype = C.c_uint32
        L.mpo_golomb_read.argtypes = [C.c_uint32, C.POINTER(Bits)]
        L.mpo_golomb_len.restype = C.c_uint32
        L.mpo_golomb_len.argtypes = [C.c_uint32, C.c_uint32]
        L.mpo_elias_write.argtypes = [C.c_uint32, C.POINTER(Bits)]
        L.mpo_elias_read.restype = C.c_uint32
        L.mpo_elias_read.argtypes = [C.POINTER(Bits)]
        L.mpo_elias_len.restype = C.c_uint32
        L.mpo_elias_len.argtypes = [C.c_uint32]
        L.mpo_ef_write.restype = C.c_int
        L.mpo_ef_write.argtypes = [u16p, C.c_size_t, C.c_uint16, C.POINTER(Bits)]
        L.mpo_ef_read.restype = C.c_int
        L.mpo_ef_read.argtypes = [u16p, C.c_size_t, C.c_uint16, C.POINTER(Bits)]
        L.mpo_ef_len.restype = C.c_uint32
        L.mpo_ef_len.argtypes = [C.c_size_t, C.c_uint16]
        L.mpo_huffman_encode.argtypes = [u16p, C.c_size_t, C.POINTER(Bits)]
        L.mpo_huffman_decode.restype = C.c_int
        L.mpo_huffman_decode.argtypes = [C.POINTER(Bits), C.POINTER(U16V)]
        L.mpo_rle_encode.argtypes = [u16p, C.c_size_t, C.POINTER(U16V)]
        L.mpo_rle_decode.argtypes = [u16p, C.c_size_t, C.POINTER(U16V)]
        L.mpo_u16v_free.argtypes = [C.POINTER(U16V)]
        L.mpo_set_umap_order.argtypes = [C.c_int]
        L.mpo_distinct_line_shapes.restype = C.c_int
        L.mpo_distinct_line_shapes.argtypes = [C.c_int, C.POINTER(Line), C.c_int]
        _lib = L
    return _lib


def ref():
    """oracle/_ref/libref.so (reference SimpleMatrix/ImageHelper object code) or None."""
    global _ref
    if _ref is None:
        path = os.path.join(HERE, "_ref", "libref.so")
        if not os.path.exists(path):
            return None
        R = C.CDLL(path)
        dp = C.POINTER(C.c_double)
        R.ref_symm_eigen.argtypes = [dp, C.c_int, dp, dp]
        R.ref_multiply.argtypes = [dp, C.c_int, C.c_int, dp, dp]
        R.ref_scale_subtract.argtypes = [dp, dp, C.c_double, C.c_int]
        R.ref_yuv_from_rgb.argtypes = [C.c_uint8, C.c_uint8, C.c_uint8, dp]
        R.ref_rgb_from_yuv.argtypes = [C.c_double, C.c_double, C.c_double, C.POINTER(C.c_uint8)]
        _ref = R
    return _ref


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _u8p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _u16p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint16))


class OracleContext:
    """createCompressionContext(K, blockSize, bpp) of the double path."""

    def __init__(self, K=32, bs=8, bpp=3.5):
        self.L = lib()
        self.p = self.L.mpo_ctx_create(K, bs, float(bpp))
        if not self.p:
            raise ValueError("bad K / blockSize")
        c = self.p.contents
        self.K, self.bs, self.N, self.nbase = c.K, c.bs, c.N, c.nbase
        self.base = np.ctypeslib.as_array(c.base, shape=(self.nbase, self.N)).copy()
        self.det_rows = np.ctypeslib.as_array(c.det_rows, shape=(self.nbase,)).copy()
        self.det_off = np.ctypeslib.as_array(c.det_off, shape=(self.nbase + 1,)).copy()
        total = int(self.det_off[-1])
        self.det = [np.ctypeslib.as_array(c.det[ch], shape=(total, self.N)).copy() for ch in range(3)]
        self.quant = np.array([[c.quant[ch][i] for i in range(K)] for ch in range(3)], dtype=np.float64)
        self.lines = [(c.lines[i].ax, c.lines[i].ay, c.lines[i].bx, c.lines[i].by) for i in range(self.nbase)]

    def close(self):
        if self.p:
            self.L.mpo_ctx_destroy(self.p)
            self.p = None

    def __del__(self):
        self.close()

    def calc_mp(self, ch, vec, quant=None):
        q = np.ascontiguousarray(self.quant[ch] if quant is None else quant, dtype=np.float64)
        v = np.ascontiguousarray(vec, dtype=np.float64)
        d = np.zeros(self.K, np.uint16)
        k = np.zeros(self.K, np.uint16)
        res = np.zeros(self.N, np.float64)
        S = C.c_uint32(0)
        cnt = self.L.mpo_calc_mp(self.p, ch, _dp(q), _dp(v), _u16p(d), _u16p(k), _dp(res), C.byref(S))
        return cnt, d, k, res, S.value

    def from_coeffs(self, ch, count, d, k, quant=None):
        q = np.ascontiguousarray(self.quant[ch] if quant is None else quant, dtype=np.float64)
        d = np.ascontiguousarray(d, np.uint16)
        k = np.ascontiguousarray(k, np.uint16)
        out = np.zeros(self.N, np.float64)
        self.L.mpo_from_coeffs(self.p, ch, _dp(q), int(count), _u16p(d), _u16p(k), _dp(out))
        return out

    def encode_tiles(self, rgb, quant=None, tx_begin=0, tx_end=None):
        """Per-tile records in the reference's x-outer/y-inner tile order."""
        rgb = np.ascontiguousarray(rgb, np.uint8)
        H, W = rgb.shape[:2]
        q = np.ascontiguousarray(self.quant if quant is None else quant, dtype=np.float64)
        tx = (W + self.bs - 1) // self.bs
        ty = (H + self.bs - 1) // self.bs
        tx_end = tx if tx_end is None else tx_end
        T = tx * ty
        counts = np.zeros((T, 3), np.uint16)
        delta = np.zeros((T, 3, self.K), np.uint16)
        coef = np.zeros((T, 3, self.K), np.uint16)
        energy = np.zeros((T, 3), np.float64)
        swept = np.zeros((T, 3), np.uint32)
        self.L.mpo_encode_tiles(self.p, _u8p(rgb), W, H, _dp(q[0]), _dp(q[1]), _dp(q[2]), tx_begin, tx_end,
                                _u16p(counts), _u16p(delta), _u16p(coef), _dp(energy),
                                swept.ctypes.data_as(C.POINTER(C.c_uint32)))
        return counts, delta, coef, energy, swept

    def encode_image(self, rgb, quant=None):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        H, W = rgb.shape[:2]
        q = np.ascontiguousarray(self.quant if quant is None else quant, dtype=np.float64)
        n = C.c_size_t(0)
        p = self.L.mpo_encode_image(self.p, _u8p(rgb), W, H, _dp(q[0]), _dp(q[1]), _dp(q[2]), C.byref(n))
        out = bytes(C.string_at(p, n.value))
        _libc_free(p)
        return out


class OracleFastContext:
    """The `...Fast` (float) flavour over an OracleContext's dictionary: mpo_fast.c's definition of the float mode."""

    def __init__(self, octx):
        self.ctx = octx
        self.L = octx.L
        self.K, self.N = octx.K, octx.N
        self.p = self.L.mpo_fast_create(octx.p)

    def close(self):
        if self.p:
            self.L.mpo_fast_destroy(self.p)
            self.p = None

    def __del__(self):
        self.close()

    def calc_mp(self, ch, vec, quant=None):
        fp = C.POINTER(C.c_float)
        q = np.ascontiguousarray(self.ctx.quant[ch] if quant is None else quant, dtype=np.float32)
        v = np.ascontiguousarray(vec, dtype=np.float32)
        d = np.zeros(self.K, np.uint16)
        k = np.zeros(self.K, np.uint16)
        res = np.zeros(self.N, np.float32)
        S = C.c_uint32(0)
        cnt = self.L.mpo_calc_mp_fast(self.p, ch, q.ctypes.data_as(fp), v.ctypes.data_as(fp), _u16p(d), _u16p(k),
                                      res.ctypes.data_as(fp), C.byref(S))
        return cnt, d, k, res, S.value

    def encode_tiles(self, rgb, quant=None, tx_begin=0, tx_end=None):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        H, W = rgb.shape[:2]
        q = np.ascontiguousarray(self.ctx.quant if quant is None else quant, dtype=np.float64)
        bs = self.ctx.bs
        tx, ty = (W + bs - 1) // bs, (H + bs - 1) // bs
        tx_end = tx if tx_end is None else tx_end
        T = tx * ty
        counts = np.zeros((T, 3), np.uint16)
        delta = np.zeros((T, 3, self.K), np.uint16)
        coef = np.zeros((T, 3, self.K), np.uint16)
        energy = np.zeros((T, 3), np.float64)
        swept = np.zeros((T, 3), np.uint32)
        self.L.mpo_encode_tiles_fast(self.p, _u8p(rgb), W, H, _dp(q[0]), _dp(q[1]), _dp(q[2]), tx_begin, tx_end,
                                     _u16p(counts), _u16p(delta), _u16p(coef), _dp(energy),
                                     swept.ctypes.data_as(C.POINTER(C.c_uint32)))
        return counts, delta, coef, energy, swept

    def encode_image(self, rgb, quant=None):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        H, W = rgb.shape[:2]
        q = np.ascontiguousarray(self.ctx.quant if quant is None else quant, dtype=np.float64)
        n = C.c_size_t(0)
        p = self.L.mpo_encode_image_fast(self.p, _u8p(rgb), W, H, _dp(q[0]), _dp(q[1]), _dp(q[2]), C.byref(n))
        out = bytes(C.string_at(p, n.value))
        _libc_free(p)
        return out


def decode_image_fast(data):
    L = lib()
    buf = np.frombuffer(data, np.uint8)
    out = C.POINTER(C.c_uint8)()
    W = C.c_int(0)
    H = C.c_int(0)
    rc = L.mpo_decode_image_fast(_u8p(buf), len(data), C.byref(out), C.byref(W), C.byref(H))
    if rc != 0:
        raise ValueError("invalid bitstream")
    img = np.ctypeslib.as_array(out, shape=(H.value, W.value, 3)).copy()
    _libc_free(out)
    return img


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


class OraclePatchStats:
    """The "-s" mode of Compression.cpp:200-302 (one std::mt19937, Welford statistics per step)."""

    def __init__(self, ctx, seed):
        self.ctx = ctx
        self.p = ctx.L.mpo_patch_stats_create(ctx.K, int(seed))

    def add_image(self, rgb, patches):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        H, W, _ = rgb.shape
        self.ctx.L.mpo_patch_stats_add_image(self.p, self.ctx.p, _u8p(rgb), W, H, int(patches))

    def read(self):
        out = np.zeros((3, 2, self.ctx.K, 5), np.float64)
        self.ctx.L.mpo_patch_stats_read(self.p, _dp(out))
        return out

    def close(self):
        if self.p:
            self.ctx.L.mpo_patch_stats_destroy(self.p)
            self.p = None

    def __del__(self):
        self.close()


def _libc_free(p):
    _libc.free(C.cast(p, C.c_void_p))


def decode_image(data):
    L = lib()
    buf = np.frombuffer(data, np.uint8)
    out = C.POINTER(C.c_uint8)()
    W = C.c_int(0)
    H = C.c_int(0)
    rc = L.mpo_decode_image(_u8p(buf), len(data), C.byref(out), C.byref(W), C.byref(H))
    if rc != 0:
        raise ValueError("invalid bitstream")
    img = np.ctypeslib.as_array(out, shape=(H.value, W.value, 3)).copy()
    _libc_free(out)
    return img


def read_compressed(data):
    """-> dict(W,H,K,bs,quant[3][K],lengths,codes[6K]) with DC diff undone."""
    L = lib()
    buf = np.frombuffer(data, np.uint8)
    s = Streams()
    rc = L.mpo_read_compressed(_u8p(buf), len(data), C.byref(s))
    if rc != 0:
        L.mpo_streams_free(C.byref(s))
        raise ValueError("invalid bitstream")
    K = s.K

    def arr(v):
        return np.ctypeslib.as_array(v.d, shape=(v.n,)).copy() if v.n else np.zeros(0, np.uint16)
    out = dict(W=s.W, H=s.H, K=K, bs=s.bs,
               quant=np.array([[s.quant[ch][i] for i in range(K)] for ch in range(3)], np.uint16),
               lengths=arr(s.lengths), codes=[arr(s.codes[i]) for i in range(6 * K)])
    L.mpo_streams_free(C.byref(s))
    return out


def write_compressed(st):
    """Inverse of read_compressed (codes given with DC diff undone, as the encoder holds them)."""
    L = lib()
    s = Streams()
    s.W, s.H, s.K, s.bs = st["W"], st["H"], st["K"], st["bs"]
    K = s.K
    for ch in range(3):
        for i in range(K):
            s.quant[ch][i] = int(st["quant"][ch][i])

    def fill(v, a):
        a = np.ascontiguousarray(a, np.uint16)
        n = a.size
        mem = _libc_malloc(max(2 * n, 2))
        C.memmove(mem, a.ctypes.data, 2 * n)
        v.d = C.cast(mem, C.POINTER(C.c_uint16))
        v.n = n
        v.cap = n
    fill(s.lengths, st["lengths"])
    for i in range(6 * K):
        fill(s.codes[i], st["codes"][i])
    n = C.c_size_t(0)
    p = L.mpo_write_compressed(C.byref(s), C.byref(n))
    out = bytes(C.string_at(p, n.value))
    _libc_free(p)
    L.mpo_streams_free(C.byref(s))
    return out


_libc.malloc.restype = C.c_void_p
_libc.malloc.argtypes = [C.c_size_t]


def _libc_malloc(n):
    return _libc.malloc(n)


def synth_frame(W, H, seed=12345):
    out = np.zeros((H, W, 3), np.uint8)
    lib().mpo_synth_frame(_u8p(out), W, H, seed)
    return out
