"""Symbol streams that exercise every branch of the entropy stage (CompressedImage.cpp:359-460, Huffman.cpp:246-279):
runs cut at 0x8001 symbols, runs across the device's 4096-symbol blocks, the `packed + 4 < size` boundary, empty and
one-symbol streams, symbols up to 0xFFFF, Huffman ties in every order of first appearance, Golomb-friendly streams.

K = 4 on a 2048x1536 frame (49 152 tiles): 24 code streams of up to 1 179 648 symbols in all, 147 456 lengths.
`as_coded`: what the encoder's stream assembly hands the entropy stage (step-0 coefficient streams already differenced);
`as_held`: the same with those three streams undifferenced (what writeCompressed / the oracle take)."""
import numpy as np

W, H, K, BS = 2048, 1536, 4, 8
TILES = (W // 8) * (H // 8)
CHUNK = 0x8001


def _zigzag(d):
    d = d.astype(np.int64)
    return ((d << 1) ^ (d >> 63)).astype(np.uint64)


def _dc_difference(v):
    v = v.astype(np.int64)
    prev = np.concatenate([[0], v[:-1]])
    return (_zigzag(v - prev) & 0xFFFF).astype(np.uint16)


def make():
    rng = np.random.default_rng(20241004)
    u16 = lambda a: np.asarray(a, dtype=np.uint16)                                    # noqa: E731
    geometric = lambda p, n: u16(np.minimum(rng.geometric(p, n) - 1, 65535))         # noqa: E731
    runs = lambda lens, vals: u16(np.repeat(vals, lens))                              # noqa: E731
    held = [
        u16([]),                                                                       # 0 empty
        u16([5]),                                                                      # 1 (DC slot) one symbol
        u16(np.zeros(100000)),                                                         # 2 three full chunks + a rest
        u16(np.full(CHUNK, 7)),                                                        # 3 exactly one chunk
        u16(np.full(CHUNK + 1, 7)),                                                    # 4 chunk + 1
        u16(np.full(CHUNK + 2, 9)),                                                    # 5 chunk + 2
        geometric(0.3, 60000),                                                         # 6 small alphabet
        u16(rng.integers(0, 65536, 20000)),                                            # 7 symbols beyond the LDS bins
        runs(rng.integers(1, 11, 9000), rng.integers(0, 4, 9000)),                     # 8 short runs (adjacent runs may merge)
        u16(np.cumsum(rng.integers(-3, 4, 40000)) + 1000),                             # 9 (DC slot, 2K+1) random walk
        runs([4095, 4097, 8192, 1, 4096, 4096, 3], [1, 2, 3, 4, 5, 5, 6]),             # 10 runs against the 4096-symbol blocks
        runs(np.full(5000, 3), rng.integers(0, 50, 5000)),                             # 11 runs of three
        u16(rng.integers(0, 8192, 50000)),                                             # 12 a large flat alphabet
        u16(np.concatenate([np.zeros(CHUNK - 1), [1]])),                               # 13 one symbol short of a chunk
        u16(np.concatenate([np.zeros(2 * CHUNK), [1, 1]])),                            # 14 two chunks exactly, then a pair
        runs(np.full(3000, 2), np.arange(3000) % 7),                                   # 15 pairs: run lengths make it longer
        u16(np.full(7, 3)),                                                            # 16 packed + 4 == size: not shorter
        u16(np.concatenate([np.cumsum(rng.integers(-1, 2, 5000)) + 300, np.full(70000, 123)])),   # 17 (DC slot, 4K+1) walk, then flat
        u16(np.full(50000, 0xFFFF)),                                                   # 18 the largest symbol, in runs
        u16(np.arange(30000)),                                                         # 19 all distinct, ascending
        u16(np.arange(30000)[::-1]),                                                   # 20 all distinct, descending
        u16(np.tile(rng.permutation(64), 200)),                                        # 21 64 symbols, equal counts: all ties
        u16(np.minimum(rng.geometric(0.004, 30000) - 1, 4000)),                        # 22 wide geometric: large Golomb M
        u16(np.concatenate([np.full(8, 3), rng.integers(0, 3, 4096 * 3 - 8), np.full(2 * CHUNK + 5, 2), [1]])),   # 23 mixture
    ]
    assert len(held) == 6 * K
    coded = list(held)
    for i in (1, 2 * K + 1, 4 * K + 1):
        coded[i] = _dc_difference(held[i])
    counts = u16(rng.integers(0, K + 1, 3 * TILES))
    assert sum(len(x) for x in coded) <= 2 * 3 * TILES * K
    return dict(W=W, H=H, K=K, bs=BS, counts=counts, as_coded=coded, as_held=held)


def quant(K):
    return np.arange(1, 3 * K + 1, dtype=np.float64).reshape(3, K) * 3
