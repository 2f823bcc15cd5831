#!/usr/bin/env python3
"""fuzz_streams.py -- random symbol streams through the device-side entropy stage against the oracle's writeCompressed.
A hunting tool, not a test: python tools/fuzz_streams.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def random_stream(rng, cap):
    kind = int(rng.integers(0, 8))
    n = int(rng.integers(0, cap + 1)) if rng.random() < 0.9 else 0
    if n == 0:
        return np.zeros(0, np.uint16)
    if kind == 0:
        return np.minimum(rng.geometric(rng.uniform(0.02, 0.9), n) - 1, 65535).astype(np.uint16)
    if kind == 1:
        return rng.integers(0, int(rng.choice([2, 16, 300, 9000, 65536])), n).astype(np.uint16)
    if kind == 2:                                                   # runs of random lengths, some beyond the 0x8001 cut
        out = []
        while sum(len(x) for x in out) < n:
            length = int(rng.choice([1, 2, 3, 7, 8, 4095, 4096, 4097, 0x8000, 0x8001, 0x8002, 70000])) if rng.random() < 0.5 else int(rng.integers(1, 40))
            out.append(np.full(length, rng.integers(0, int(rng.choice([3, 70000])) % 65536 + 1), np.uint16))
        return np.concatenate(out)[:n]
    if kind == 3:
        return np.full(n, rng.integers(0, 65536), np.uint16)
    if kind == 4:
        return (np.arange(n) % int(rng.integers(1, 70000))).astype(np.uint16)
    if kind == 5:                                                   # equal counts: every Huffman tie there is
        k = int(rng.integers(1, 300))
        return np.tile(rng.permutation(k), n // k + 1)[:n].astype(np.uint16)
    if kind == 6:
        return np.repeat(rng.integers(0, 5, n // 2 + 1), 2)[:n].astype(np.uint16)
    return (np.cumsum(rng.integers(-2, 3, n)) % 1000).astype(np.uint16)


def main():
    import imageexperiments_amd as ia
    from oracle import oracle_py as O
    import stream_cases
    O.build(ref=False)
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
    ctxs = {}
    bad = 0
    for c in range(cases):
        K = int(rng.integers(1, 7))
        W, H = 8 * int(rng.integers(8, 200)), 8 * int(rng.integers(8, 120))
        tiles = (W // 8) * (H // 8)
        cap = 2 * 3 * tiles * K
        held, left = [], cap
        for _ in range(6 * K):
            s = random_stream(rng, min(left, max(1, cap // (3 * K))))
            left -= len(s)
            held.append(s)
        coded = list(held)
        for i in (1, 2 * K + 1, 4 * K + 1):
            coded[i] = stream_cases._dc_difference(held[i]) if len(held[i]) else held[i]
        counts = rng.integers(0, K + 1, 3 * tiles).astype(np.uint16)
        q = stream_cases.quant(K)
        want = O.write_compressed(dict(W=W, H=H, K=K, bs=8, quant=q, lengths=counts, codes=held))
        if K not in ctxs:
            ctxs[K] = ia.create_compression_context(K, 8, 3.5, device=0)
        got, route = ctxs[K].code_symbol_streams_device(W, H, counts, coded, quant=q)
        if got != want:
            bad += 1
            print(f"MISMATCH case {c}: K={K} {W}x{H} route={route} sizes {[len(x) for x in held]}", flush=True)
    print(f"{cases} cases, {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
