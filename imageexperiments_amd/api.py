"""ctypes binding of libmpcodec.so (include/mpcodec.h) and a thin host-side mirror of the reference's
`compressed::` / `matching::` interface (CompressionLib/inc/CompressedImage.h, MatchingPursuit.h).

There is no CPU fallback: if the library is missing this module raises, and without a GPU the
encode entry points raise MpcError(MPC_ERR_NO_DEVICE).
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
MAX_K = 32
HIST_BINS = 8192

MPC_OK, MPC_ERR_ARGUMENT, MPC_ERR_NO_DEVICE, MPC_ERR_HIP, MPC_ERR_BITSTREAM, MPC_ERR_ALLOC = range(6)


class MpcError(RuntimeError):
    def __init__(self, status, text):
        super().__init__(f"mpcodec status {status}: {text}")
        self.status = status


def library_path():
    """In-tree library; MPCODEC_LIB selects another build of it (A/B timing of two kernel versions in one run)."""
    return os.environ.get("MPCODEC_LIB") or os.path.join(HERE, "lib", "libmpcodec.so")


_lib = None

_u8p = C.POINTER(C.c_uint8)
_u16p = C.POINTER(C.c_uint16)
_u32p = C.POINTER(C.c_uint32)
_dp = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)


def load_library():
    """Load libmpcodec.so; raises (loudly) if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (soname libamdhip64.so.7). Loading
    # torch FIRST makes the dynamic linker bind libmpcodec.so's libamdhip64.so.7 dependency to that same copy;
    # the other order leaves two runtimes in the process and the second one sees no devices.
    import torch  # noqa: F401
    path = library_path()
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: build it with `python -m imageexperiments_amd.build` "
                          "(there is no CPU fallback for the hot path)")
    L = C.CDLL(path)
    vp = C.c_void_p
    L.mpc_version.restype = C.c_char_p
    L.mpc_last_error.restype = C.c_char_p
    L.mpc_context_create.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, C.POINTER(vp)]
    L.mpc_context_destroy.argtypes = [vp]
    for f in ("K", "block_size", "num_base", "detail_rows", "device", "max_waves"):
        getattr(L, "mpc_context_" + f).argtypes = [vp]
        getattr(L, "mpc_context_" + f).restype = C.c_int
    L.mpc_context_set_fast.argtypes = [vp, C.c_int]
    try:
        L.mpc_context_set_tile_encode_workgroups.argtypes = [vp, C.c_int]
    except AttributeError:
        if not os.environ.get("MPCODEC_LIB"):             # an older build may be loaded for A/B timing only
            raise
    L.mpc_context_is_fast.argtypes = [vp]
    L.mpc_context_is_fast.restype = C.c_int
    L.mpc_context_get_quant.argtypes = [vp, _dp]
    L.mpc_context_set_quant.argtypes = [vp, _dp]
    L.mpc_context_get_dictionary.argtypes = [vp, _dp, _i32p, _dp, _dp, _dp]
    L.mpc_encode_tiles_device.argtypes = [vp, vp, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, _dp,
                                          vp, vp, vp, vp, C.c_int, vp]
    L.mpc_encode_batch_device.argtypes = [vp, vp, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, _dp,
                                          vp, vp, vp, vp, C.c_int, vp]
    L.mpc_encode_tiles.argtypes = [vp, _u8p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, _dp,
                                   _u16p, vp, _dp, _u32p]
    L.mpc_histogram_device.argtypes = [vp, vp, vp, C.c_longlong, vp, vp]
    L.mpc_calc_mp.argtypes = [vp, C.c_int, _dp, _dp, vp, C.POINTER(C.c_int)]
    L.mpc_calc_mp_batch.argtypes = [vp, C.c_int, _dp, _dp, C.c_int, vp, _u16p, _dp, _u32p]
    L.mpc_reserve.argtypes = [vp, C.c_longlong]
    L.mpc_kernel_timing_enable.argtypes = [vp, C.c_int]
    L.mpc_kernel_timing_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.POINTER(C.c_double)]
    L.mpc_kernel_counters_read.argtypes = [vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    _bind_bitstream(L)
    _lib = L
    return L


def _bind_bitstream(L):
    """Entry points of the host entropy stage / container."""
    vp = C.c_void_p
    L.mpc_free.argtypes = [vp]
    L.mpc_write_compressed.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _dp, _u16p, C.c_size_t,
                                       C.POINTER(_u16p), C.POINTER(C.c_size_t), C.POINTER(_u8p), C.POINTER(C.c_size_t)]
    L.mpc_assemble_streams.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _dp, _u16p, vp,
                                       C.POINTER(_u8p), C.POINTER(C.c_size_t)]
    L.mpc_assemble_planar_streams.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _dp, _u16p, vp,
                                       C.POINTER(_u8p), C.POINTER(C.c_size_t)]
    _ullp = C.POINTER(C.c_ulonglong)
    for f in ("mpc_assemble_symbol_streams", "mpc_assemble_symbol_streams_by_plan"):
        getattr(L, f).argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _dp, _u16p, _u16p, _ullp, C.POINTER(_u8p), C.POINTER(C.c_size_t)]
    L.mpc_code_symbol_streams_device.argtypes = [vp, C.c_int, C.c_int, _dp, _u16p, _u16p, _ullp, C.POINTER(_u8p), C.POINTER(C.c_size_t),
                                                 C.POINTER(C.c_int)]
    L.mpc_read_compressed.argtypes = [_u8p, C.c_size_t, C.POINTER(vp)]
    L.mpc_streams_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.mpc_streams_quant.argtypes = [vp, _u16p]
    L.mpc_streams_length.argtypes = [vp, C.c_int]
    L.mpc_streams_length.restype = C.c_size_t
    L.mpc_streams_copy.argtypes = [vp, C.c_int, _u16p]
    L.mpc_streams_free.argtypes = [vp]
    L.mpc_huffman_encode.argtypes = [_u16p, C.c_size_t, C.POINTER(_u8p), C.POINTER(C.c_size_t)]
    L.mpc_huffman_decode.argtypes = [_u8p, C.c_size_t, C.POINTER(_u16p), C.POINTER(C.c_size_t)]
    L.mpc_rle_encode.argtypes = [_u16p, C.c_size_t, C.POINTER(_u16p), C.POINTER(C.c_size_t)]
    L.mpc_rle_decode.argtypes = [_u16p, C.c_size_t, C.POINTER(_u16p), C.POINTER(C.c_size_t)]
    L.mpc_encode_image.argtypes = [vp, _u8p, C.c_int, C.c_int, _dp, C.POINTER(_u8p), C.POINTER(C.c_size_t)]
    L.mpc_encode_images.argtypes = [vp, C.POINTER(_u8p), C.c_int, C.c_int, C.c_int, _dp, C.POINTER(_u8p), C.POINTER(C.c_size_t)]
    try:
        L.mpc_encode_images_multi.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(_u8p), C.c_int, C.c_int, C.c_int, _dp, C.POINTER(_u8p), C.POINTER(C.c_size_t)]
    except AttributeError:
        if not os.environ.get("MPCODEC_LIB"):
            raise
    L.mpc_encode_images_device.argtypes = [vp, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, _dp, C.POINTER(_u8p), C.POINTER(C.c_size_t)]
    L.mpc_records_to_container_device.argtypes = [vp, vp, vp, C.c_int, C.c_int, _dp, vp, C.POINTER(_u8p), C.POINTER(C.c_size_t)]
    try:
        L.mpc_interleave_stripe_device.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]
    except AttributeError:
        if not os.environ.get("MPCODEC_LIB"):             # an older build may be loaded for A/B timing only
            raise
    L.mpc_encode_image_device.argtypes = [vp, vp, C.c_int, C.c_int, _dp, C.POINTER(_u8p), C.POINTER(C.c_size_t)]
    L.mpc_container_job_begin.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.c_int, _dp, vp]
    L.mpc_container_job_tables.argtypes = [vp, C.c_int]
    L.mpc_container_job_collect.argtypes = [vp, C.c_int, C.POINTER(_u8p), C.POINTER(C.c_size_t)]
    try:
        L.mpc_container_job_cancel.argtypes = [vp, C.c_int]
    except AttributeError:
        if not os.environ.get("MPCODEC_LIB"):
            raise
    L.mpc_decode_tiles_device.argtypes = [vp, vp, vp, _dp, C.c_int, C.c_int, vp, vp]
    L.mpc_decode_image.argtypes = [vp, _u8p, C.c_size_t, C.POINTER(_u8p), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.mpc_patch_stats_create.argtypes = [vp, C.c_uint, C.POINTER(vp)]
    L.mpc_patch_stats_destroy.argtypes = [vp]
    L.mpc_patch_stats_destroy.restype = None
    L.mpc_patch_stats_add_image.argtypes = [vp, _u8p, C.c_int, C.c_int, C.c_int]
    L.mpc_patch_stats_read.argtypes = [vp, _dp]
    L.mpc_patch_stats_report.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t)]
    L.mpc_format_double.argtypes = [C.c_double, C.c_char_p, C.c_int]
    L.mpc_psnr.argtypes = [_u8p, _u8p, C.c_int, C.c_int]
    L.mpc_psnr.restype = C.c_double


def _take_bytes(L, p, n):
    out = C.string_at(p, n.value)
    L.mpc_free(C.cast(p, C.c_void_p))
    return out


class _Owned:
    """Keeps a buffer the library malloc'ed alive for the numpy view built on it, and frees it with mpc_free."""

    def __init__(self, L, p):
        self.L, self.p = L, C.cast(p, C.c_void_p)

    def __del__(self):
        if self.p:
            self.L.mpc_free(self.p)
            self.p = None


def _take_view(L, p, n):
    """The container as a read-only uint8 array ON the library's buffer (no copy; the array owns the buffer)."""
    if not n.value:
        L.mpc_free(C.cast(p, C.c_void_p))
        return np.zeros(0, np.uint8)
    owner = _Owned(L, p)
    buf = (C.c_uint8 * n.value).from_address(C.cast(p, C.c_void_p).value)
    buf._owner = owner                                   # the ctypes array is the base of the view: freed with it
    out = np.frombuffer(buf, np.uint8)
    out.flags.writeable = False
    return out


def _take_u16(L, p, n):
    out = np.ctypeslib.as_array(p, shape=(n.value,)).copy() if n.value else np.zeros(0, np.uint16)
    L.mpc_free(C.cast(p, C.c_void_p))
    return out


def _u16(a):
    a = np.ascontiguousarray(a, np.uint16)
    return a, a.ctypes.data_as(_u16p)


# -- host entropy stage (huffman:: / compressed:: free functions of the reference) -----------------------------
def huffman_encode(data):
    """huffman::huffmanEncode (Huffman.h:15) -> bytes."""
    L = load_library()
    a, p = _u16(data)
    out, n = _u8p(), C.c_size_t(0)
    _check(L.mpc_huffman_encode(p, a.size, C.byref(out), C.byref(n)))
    return _take_bytes(L, out, n)


def huffman_decode(blob):
    """huffman::huffmanDecode (Huffman.h:18) -> uint16 array; raises MpcError(MPC_ERR_BITSTREAM) on invalid data."""
    L = load_library()
    buf = np.frombuffer(bytes(blob), np.uint8)
    out, n = _u16p(), C.c_size_t(0)
    _check(L.mpc_huffman_decode(buf.ctypes.data_as(_u8p), buf.size, C.byref(out), C.byref(n)))
    return _take_u16(L, out, n)


def run_length_encode(data):
    L = load_library()
    a, p = _u16(data)
    out, n = _u16p(), C.c_size_t(0)
    _check(L.mpc_rle_encode(p, a.size, C.byref(out), C.byref(n)))
    return _take_u16(L, out, n)


def run_length_decode(data):
    L = load_library()
    a, p = _u16(data)
    out, n = _u16p(), C.c_size_t(0)
    _check(L.mpc_rle_decode(p, a.size, C.byref(out), C.byref(n)))
    return _take_u16(L, out, n)


def write_compressed(width, height, K, block_size, quant, lengths, codes):
    """compressed::writeCompressed (CompressedImage.cpp:403): codes = 6K uint16 arrays, DC not yet differenced."""
    L = load_library()
    q = np.ascontiguousarray(quant, np.float64).reshape(3 * K)
    ln, lp = _u16(lengths)
    arrs = [np.ascontiguousarray(c, np.uint16) for c in codes]
    ptrs = (_u16p * (6 * K))(*[a.ctypes.data_as(_u16p) for a in arrs])
    sizes = (C.c_size_t * (6 * K))(*[a.size for a in arrs])
    out, n = _u8p(), C.c_size_t(0)
    _check(L.mpc_write_compressed(width, height, K, block_size, q.ctypes.data_as(_dp), lp, ln.size, ptrs, sizes,
                                  C.byref(out), C.byref(n)))
    return _take_bytes(L, out, n)


def assemble_streams(width, height, K, block_size, quant, counts, choices):
    """Host half of encodeImage: whole-frame records (tile t = tx*tiles_y + ty) -> container bytes."""
    L = load_library()
    q = np.ascontiguousarray(quant, np.float64).reshape(3 * K)
    cn, cp = _u16(counts)
    ch = np.ascontiguousarray(choices)
    out, n = _u8p(), C.c_size_t(0)
    _check(L.mpc_assemble_streams(width, height, K, block_size, q.ctypes.data_as(_dp), cp, ch.ctypes.data_as(C.c_void_p),
                                  C.byref(out), C.byref(n)))
    return _take_bytes(L, out, n)


def assemble_planar_streams(width, height, K, block_size, quant, counts, planar):
    """assemble_streams for records in planar order, planar[ch, step, tile] (uint32 deltaId | intCoeff << 16)."""
    L = load_library()
    q = np.ascontiguousarray(quant, np.float64).reshape(3 * K)
    cn, cp = _u16(counts)
    pl = np.ascontiguousarray(planar)
    out, n = _u8p(), C.c_size_t(0)
    _check(L.mpc_assemble_planar_streams(width, height, K, block_size, q.ctypes.data_as(_dp), cp, pl.ctypes.data_as(C.c_void_p),
                                         C.byref(out), C.byref(n)))
    return _take_bytes(L, out, n)


def _symbol_streams(K, streams):
    streams = [np.ascontiguousarray(x, np.uint16).ravel() for x in streams]
    assert len(streams) == 6 * K
    off = np.zeros(6 * K + 1, np.uint64)
    off[1:] = np.cumsum([len(x) for x in streams])
    symbols = np.concatenate(streams) if int(off[-1]) else np.zeros(1, np.uint16)
    return np.ascontiguousarray(symbols, np.uint16), off


def assemble_symbol_streams(width, height, K, block_size, quant, counts, streams, by_plan=False):
    """Entropy stage + container on the host from streams that are already assembled: streams[6K] = codes[0..6K) (live symbols,
    the step-0 coefficient streams already difference coded), counts[3*tiles] = the lengths stream.
    by_plan: take the route of the device-side entropy stage (statistics -> tables and offsets -> codes) on the host."""
    L = load_library()
    q = np.ascontiguousarray(quant, np.float64).reshape(3 * K)
    cn, cp = _u16(counts)
    symbols, off = _symbol_streams(K, streams)
    out, n = _u8p(), C.c_size_t(0)
    fn = L.mpc_assemble_symbol_streams_by_plan if by_plan else L.mpc_assemble_symbol_streams
    _check(fn(width, height, K, block_size, q.ctypes.data_as(_dp), cp, symbols.ctypes.data_as(_u16p),
              off.ctypes.data_as(C.POINTER(C.c_ulonglong)), C.byref(out), C.byref(n)))
    return _take_bytes(L, out, n)


def read_compressed(blob):
    """compressed::readCompressed (CompressedImage.cpp:635) -> dict(W,H,K,bs,quant[3,K],lengths,codes[6K])."""
    L = load_library()
    buf = np.frombuffer(bytes(blob), np.uint8)
    h = C.c_void_p()
    _check(L.mpc_read_compressed(buf.ctypes.data_as(_u8p), buf.size, C.byref(h)))
    try:
        W, H, K, bs = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        _check(L.mpc_streams_info(h, C.byref(W), C.byref(H), C.byref(K), C.byref(bs)))
        quant = np.zeros((3, K.value), np.uint16)
        _check(L.mpc_streams_quant(h, quant.ctypes.data_as(_u16p)))

        def stream(i):
            a = np.zeros(L.mpc_streams_length(h, i), np.uint16)
            if a.size:
                _check(L.mpc_streams_copy(h, i, a.ctypes.data_as(_u16p)))
            return a
        return dict(W=W.value, H=H.value, K=K.value, bs=bs.value, quant=quant, lengths=stream(-1),
                    codes=[stream(i) for i in range(6 * K.value)])
    finally:
        L.mpc_streams_free(h)


def decode_image(blob, ctx):
    """compressed::decodeImage (CompressedImage.h:75) -> uint8 [H,W,3]; reconstructed on ctx's device (no host path)."""
    L = load_library()
    buf = np.frombuffer(blob, np.uint8)                      # no copy of the container: bytes, bytearray and arrays alike
    out, W, H = _u8p(), C.c_int(), C.c_int()
    _check(L.mpc_decode_image(ctx.h, buf.ctypes.data_as(_u8p), buf.size, C.byref(out),
                              C.byref(W), C.byref(H)))
    # the pixels stay in the buffer the library returned (the array owns it and frees it with mpc_free): no 48 MB copy per frame
    img = _take_view(L, out, C.c_size_t(3 * W.value * H.value)).reshape(H.value, W.value, 3)
    return img


def calculate_psnr(original, decoded):
    """compressed::calculatePSNR (CompressedImage.h:57)."""
    L = load_library()
    a = np.ascontiguousarray(original, np.uint8)
    b = np.ascontiguousarray(decoded, np.uint8)
    return L.mpc_psnr(a.ctypes.data_as(_u8p), b.ctypes.data_as(_u8p), a.shape[1], a.shape[0])


def format_double(v):
    """std::format("{}", double) as the reference's reports print numbers (shortest round-trip text)."""
    buf = C.create_string_buffer(64)
    n = load_library().mpc_format_double(float(v), buf, 64)
    if n < 0:
        raise ValueError("buffer too small")
    return buf.value.decode()


class PatchStatistics:
    """The "-s" mode of Compression.cpp:200-302: random patches of every image through CalcMPDynamic with all
    quantisers 1.0 (on ctx's device), Welford statistics of intCoeff / deltaId per step, and the text report."""

    def __init__(self, ctx, seed):
        self.L = load_library()
        self.ctx = ctx
        self.h = None
        h = C.c_void_p()
        _check(self.L.mpc_patch_stats_create(ctx.h, int(seed), C.byref(h)))
        self.h = h

    def add_image(self, rgb, patches):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        H, W, _ = rgb.shape
        _check(self.L.mpc_patch_stats_add_image(self.h, rgb.ctypes.data_as(_u8p), W, H, int(patches)))

    def read(self):
        """[3 channels][intCoeff, deltaId][K steps][N, min, max, mean, sumSq]"""
        out = np.zeros((3, 2, self.ctx.K, 5), np.float64)
        _check(self.L.mpc_patch_stats_read(self.h, out.ctypes.data_as(_dp)))
        return out

    def report(self):
        text, n = C.c_char_p(), C.c_size_t()
        _check(self.L.mpc_patch_stats_report(self.h, C.byref(text), C.byref(n)))
        out = C.string_at(text, n.value).decode()
        self.L.mpc_free(C.cast(text, C.c_void_p))
        return out

    def close(self):
        if self.h:
            self.L.mpc_patch_stats_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


def _check(st):
    if st != MPC_OK:
        raise MpcError(st, load_library().mpc_last_error().decode())


CHOICE_DTYPE = np.dtype([("deltaId", "<u2"), ("intCoeff", "<u2")])


class CompressionContext:
    """compressed::CompressionContext (CompressedImage.h:28-36): K, BlockSize, the dictionary and the three
    quantisation tables; `device` >= 0 uploads the dictionary once to that GPU."""

    def __init__(self, K=32, block_size=8, bpp=3.5, device=-1):
        self.L = load_library()
        h = C.c_void_p()
        _check(self.L.mpc_context_create(int(K), int(block_size), float(bpp), int(device), C.byref(h)))
        self.h = h
        self.K = K
        self.block_size = block_size
        self.device = device
        self.num_base = self.L.mpc_context_num_base(h)
        self.detail_rows = self.L.mpc_context_detail_rows(h)

    def close(self):
        if getattr(self, "h", None):
            self.L.mpc_context_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- tables ----------------------------------------------------------------------------------
    def set_fast(self, on=True):
        """mpc_context_set_fast: the `...Fast` (float) flavour of the tile path for every later encode / decode of this context."""
        _check(self.L.mpc_context_set_fast(self.h, 1 if on else 0))
        return self

    @property
    def fast(self):
        return bool(self.L.mpc_context_is_fast(self.h))

    @property
    def quant(self):
        q = np.zeros((3, self.K), np.float64)
        _check(self.L.mpc_context_get_quant(self.h, q.ctypes.data_as(_dp)))
        return q

    def set_quant(self, q):
        q = np.ascontiguousarray(q, np.float64).reshape(3, self.K)
        _check(self.L.mpc_context_set_quant(self.h, q.ctypes.data_as(_dp)))

    def dictionary(self):
        """-> base[num_base,64], block_rows[num_base], detail[3][detail_rows,64] (host copies)."""
        n = self.block_size * self.block_size
        base = np.zeros((self.num_base, n))
        rows = np.zeros(self.num_base, np.int32)
        det = [np.zeros((self.detail_rows, n)) for _ in range(3)]
        _check(self.L.mpc_context_get_dictionary(self.h, base.ctypes.data_as(_dp), rows.ctypes.data_as(_i32p),
                                                 det[0].ctypes.data_as(_dp), det[1].ctypes.data_as(_dp),
                                                 det[2].ctypes.data_as(_dp)))
        return base, rows, det

    def set_tile_encode_workgroups(self, workgroups):
        """mpc_context_set_tile_encode_workgroups: CUs the tile encode may fill (0 = all), for callers with other work on the device."""
        _check(self.L.mpc_context_set_tile_encode_workgroups(self.h, int(workgroups)))

    @property
    def max_waves(self):
        return self.L.mpc_context_max_waves(self.h)

    # -- hot path --------------------------------------------------------------------------------
    def encode_tiles(self, rgb, tile_row_begin=0, tile_row_end=None, quant=None):
        """Host-buffer form. rgb: uint8 [H,W,3]. Returns counts[T,3], choices[T,3,K] (structured),
        energy[T,3], swept[T,3]; tile t = tx*rows + (ty - tile_row_begin)."""
        rgb = np.ascontiguousarray(rgb, np.uint8)
        H, W = rgb.shape[:2]
        ty = (H + 7) // 8
        tx = (W + 7) // 8
        tile_row_end = ty if tile_row_end is None else tile_row_end
        T = tx * (tile_row_end - tile_row_begin)
        counts = np.zeros((T, 3), np.uint16)
        choices = np.zeros((T, 3, self.K), CHOICE_DTYPE)
        energy = np.zeros((T, 3), np.float64)
        swept = np.zeros((T, 3), np.uint32)
        qp = None
        if quant is not None:
            quant = np.ascontiguousarray(quant, np.float64).reshape(3, self.K)
            qp = quant.ctypes.data_as(_dp)
        _check(self.L.mpc_encode_tiles(self.h, rgb.ctypes.data_as(_u8p), W, H, 3 * W, tile_row_begin, tile_row_end, qp,
                                       counts.ctypes.data_as(_u16p), choices.ctypes.data_as(C.c_void_p),
                                       energy.ctypes.data_as(_dp), swept.ctypes.data_as(_u32p)))
        return counts, choices, energy, swept

    def encode_tiles_device(self, d_rgb, width, height, row_stride, tile_row_begin, tile_row_end,
                            d_counts, d_choices, d_energy=0, d_swept=0, quant=None, waves=0, stream=0):
        """Device-pointer form (ints from tensor.data_ptr()); asynchronous on `stream` (hipStream_t int)."""
        qp = None
        if quant is not None:
            quant = np.ascontiguousarray(quant, np.float64).reshape(3, self.K)
            qp = quant.ctypes.data_as(_dp)
        _check(self.L.mpc_encode_tiles_device(self.h, d_rgb, width, height, row_stride, tile_row_begin, tile_row_end,
                                              qp, d_counts, d_choices, d_energy or None, d_swept or None,
                                              waves, stream or None))

    def encode_batch_device(self, d_rgb, frames, frame_stride, width, height, row_stride, tile_row_begin, tile_row_end,
                            d_counts, d_choices, d_energy=0, d_swept=0, quant=None, waves=0, stream=0):
        """`frames` frames `frame_stride` bytes apart, same tile rows of each, one launch."""
        qp = None
        if quant is not None:
            quant = np.ascontiguousarray(quant, np.float64).reshape(3, self.K)
            qp = quant.ctypes.data_as(_dp)
        _check(self.L.mpc_encode_batch_device(self.h, d_rgb, frames, frame_stride, width, height, row_stride,
                                              tile_row_begin, tile_row_end, qp, d_counts, d_choices,
                                              d_energy or None, d_swept or None, waves, stream or None))

    def histogram_device(self, d_counts, d_choices, tiles, d_hist, stream=0):
        _check(self.L.mpc_histogram_device(self.h, d_counts, d_choices, tiles, d_hist, stream or None))

    def reserve(self, max_tiles):
        """Pre-allocate the device workspace for calls of up to `max_tiles` tiles."""
        _check(self.L.mpc_reserve(self.h, int(max_tiles)))

    def kernel_timing(self, on=True):
        """Bracket every base-sweep launch with HIP events on the launch stream (measurement only)."""
        self.L.mpc_kernel_timing_enable(self.h, 1 if on else 0)

    def read_kernel_timing(self):
        """-> (summed base-sweep ms, launches, ms of the union of the launch intervals) since the last read."""
        ms, n, busy = C.c_double(0), C.c_longlong(0), C.c_double(0)
        _check(self.L.mpc_kernel_timing_read(self.h, C.byref(ms), C.byref(n), C.byref(busy)))
        return ms.value, n.value, busy.value

    def read_kernel_counters(self):
        """-> (MFMA instructions executed, tile-channel-steps) counted by the pursuit kernel since kernel_timing(True) / the last read."""
        a, b = C.c_ulonglong(0), C.c_ulonglong(0)
        _check(self.L.mpc_kernel_counters_read(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def encode_image(self, rgb, quant=None, view=False):
        """compressed::encodeImage (CompressedImage.h:59) -> bytes (view=True: a read-only uint8 array on the buffer the library
        returned, as a C caller holds it: no Python-side copy of the container)."""
        rgb = np.ascontiguousarray(rgb, np.uint8)
        H, W = rgb.shape[:2]
        qp = None
        if quant is not None:
            quant = np.ascontiguousarray(quant, np.float64).reshape(3, self.K)
            qp = quant.ctypes.data_as(_dp)
        out, n = _u8p(), C.c_size_t(0)
        _check(self.L.mpc_encode_image(self.h, rgb.ctypes.data_as(_u8p), W, H, qp, C.byref(out), C.byref(n)))
        return (_take_view if view else _take_bytes)(self.L, out, n)

    def encode_images(self, frames, quant=None, views=False):
        """encodeImage for a sequence of equally sized frames in host memory, pipelined (mpc_encode_images).  Returns a list of
        bytes objects; views=True: read-only uint8 arrays on the library's buffers (no Python-side copy)."""
        frames = [np.ascontiguousarray(f, np.uint8) for f in frames]
        H, W = frames[0].shape[:2]
        if any(f.shape[:2] != (H, W) for f in frames):
            raise ValueError("frames must have the same size")
        qp = None
        if quant is not None:
            quant = np.ascontiguousarray(quant, np.float64).reshape(3, self.K)
            qp = quant.ctypes.data_as(_dp)
        n = len(frames)
        ptrs = (_u8p * n)(*[f.ctypes.data_as(_u8p) for f in frames])
        outs = (_u8p * n)()
        sizes = (C.c_size_t * n)()
        _check(self.L.mpc_encode_images(self.h, ptrs, n, W, H, qp, outs, sizes))
        take = _take_view if views else _take_bytes
        return [take(self.L, outs[i], C.c_size_t(sizes[i])) for i in range(n)]

    def encode_images_device(self, d_frames, width, height, quant=None, views=False):
        """mpc_encode_images_device: frames already in device memory (ints from tensor.data_ptr(), tightly packed RGB).
        Returns a list of bytes objects (containers), pipelined like encode_images; views=True: read-only uint8 arrays on the
        buffers the library returned instead (what a C caller holds: no Python-side copy of every container)."""
        qp = None
        if quant is not None:
            quant = np.ascontiguousarray(quant, np.float64).reshape(3, self.K)
            qp = quant.ctypes.data_as(_dp)
        n = len(d_frames)
        ptrs = (C.c_void_p * n)(*[C.c_void_p(int(p)) for p in d_frames])
        outs = (_u8p * n)()
        sizes = (C.c_size_t * n)()
        _check(self.L.mpc_encode_images_device(self.h, ptrs, n, width, height, qp, outs, sizes))
        take = _take_view if views else _take_bytes
        return [take(self.L, outs[i], C.c_size_t(sizes[i])) for i in range(n)]

    def encode_image_device(self, d_rgb, width, height, quant=None):
        return self.encode_images_device([d_rgb], width, height, quant)[0]

    def interleave_stripe_device(self, d_part_counts, d_part_choices, width, height, tile_row_begin, tile_row_end, d_frame_counts,
                                 d_frame_choices, stream=0):
        """mpc_interleave_stripe_device: a row stripe's records (stripe order) -> their places in the whole frame's records."""
        _check(self.L.mpc_interleave_stripe_device(self.h, d_part_counts, d_part_choices, width, height, tile_row_begin, tile_row_end,
                                                   d_frame_counts, d_frame_choices, stream or None))

    def records_to_container_device(self, d_counts, d_choices, width, height, quant=None, stream=0):
        """mpc_records_to_container_device: whole-frame records in device memory -> container bytes."""
        qp = None
        if quant is not None:
            quant = np.ascontiguousarray(quant, np.float64).reshape(3, self.K)
            qp = quant.ctypes.data_as(_dp)
        out, n = _u8p(), C.c_size_t(0)
        _check(self.L.mpc_records_to_container_device(self.h, d_counts, d_choices, width, height, qp, stream or None, C.byref(out), C.byref(n)))
        return _take_bytes(self.L, out, n)

    def code_symbol_streams_device(self, width, height, counts, streams, quant=None):
        """mpc_code_symbol_streams_device: assembled streams (host) -> container bytes with the per-symbol work of the entropy
        stage on the device.  Returns (bytes, route): route 0 = device, 1 = the host route was taken."""
        qp = None
        if quant is not None:
            quant = np.ascontiguousarray(quant, np.float64).reshape(3, self.K)
            qp = quant.ctypes.data_as(_dp)
        cn, cp = _u16(counts)
        symbols, off = _symbol_streams(self.K, streams)
        out, n, route = _u8p(), C.c_size_t(0), C.c_int(-1)
        _check(self.L.mpc_code_symbol_streams_device(self.h, width, height, qp, cp, symbols.ctypes.data_as(_u16p),
                                                     off.ctypes.data_as(C.POINTER(C.c_ulonglong)), C.byref(out), C.byref(n), C.byref(route)))
        return _take_bytes(self.L, out, n), route.value

    def container_job_begin(self, slot, d_counts, d_choices, width, height, quant=None, stream=0):
        """mpc_container_job_begin: stream assembly + entropy phase 1 of whole-frame records in device memory, enqueued only."""
        qp = None
        if quant is not None:
            quant = np.ascontiguousarray(quant, np.float64).reshape(3, self.K)
            qp = quant.ctypes.data_as(_dp)
        _check(self.L.mpc_container_job_begin(self.h, slot, d_counts, d_choices, width, height, qp, stream or None))

    def container_job_tables(self, slot):
        """mpc_container_job_tables: wait for phase 1, build the code tables, enqueue phase 2 and the container's copy."""
        _check(self.L.mpc_container_job_tables(self.h, slot))

    def container_job_collect(self, slot, views=False):
        """mpc_container_job_collect: wait for the copy -> container bytes (views=True: a uint8 array on the library's buffer)."""
        out, n = _u8p(), C.c_size_t(0)
        _check(self.L.mpc_container_job_collect(self.h, slot, C.byref(out), C.byref(n)))
        return (_take_view if views else _take_bytes)(self.L, out, n)

    def container_job_cancel(self, slot):
        """mpc_container_job_cancel: give the slot up whatever step its job is at."""
        _check(self.L.mpc_container_job_cancel(self.h, slot))

    def calc_mp(self, channel, vectors, quant_k=None):
        """matching::CalcMPDynamic (MatchingPursuit.h:22) on the device for vectors[n,64].
        Returns counts[n], choices[n,K], energy[n], swept[n]."""
        v = np.ascontiguousarray(vectors, np.float64).reshape(-1, 64)
        n = v.shape[0]
        counts = np.zeros(n, np.uint16)
        choices = np.zeros((n, self.K), CHOICE_DTYPE)
        energy = np.zeros(n)
        swept = np.zeros(n, np.uint32)
        qp = None
        if quant_k is not None:
            quant_k = np.ascontiguousarray(quant_k, np.float64)
            qp = quant_k.ctypes.data_as(_dp)
        _check(self.L.mpc_calc_mp_batch(self.h, channel, qp, v.ctypes.data_as(_dp), n,
                                        choices.ctypes.data_as(C.c_void_p), counts.ctypes.data_as(_u16p),
                                        energy.ctypes.data_as(_dp), swept.ctypes.data_as(_u32p)))
        return counts, choices, energy, swept


def encode_images_multi(contexts, frames, quant=None, views=False):
    """mpc_encode_images_multi: frames (host, equal sizes) through several contexts, one per device lane: every frame's tile rows
    striped over the lanes, the stripes pulled to the frame's owner, one container per frame (byte-identical to encode_image)."""
    L = load_library()
    frames = [np.ascontiguousarray(f, np.uint8) for f in frames]
    H, W = frames[0].shape[:2]
    if any(f.shape[:2] != (H, W) for f in frames):
        raise ValueError("frames must have the same size")
    K = contexts[0].K
    qp = None
    if quant is not None:
        quant = np.ascontiguousarray(quant, np.float64).reshape(3, K)
        qp = quant.ctypes.data_as(_dp)
    n = len(frames)
    handles = (C.c_void_p * len(contexts))(*[c.h for c in contexts])
    ptrs = (_u8p * n)(*[f.ctypes.data_as(_u8p) for f in frames])
    outs = (_u8p * n)()
    sizes = (C.c_size_t * n)()
    _check(L.mpc_encode_images_multi(handles, len(contexts), ptrs, n, W, H, qp, outs, sizes))
    take = _take_view if views else _take_bytes
    return [take(L, outs[i], C.c_size_t(sizes[i])) for i in range(n)]


def create_compression_context(K=32, block_size=8, bpp=3.5, device=-1):
    """compressed::createCompressionContext(K, blockSize, bppAllocation) (CompressedImage.h:54)."""
    return CompressionContext(K, block_size, bpp, device)
