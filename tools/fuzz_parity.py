#!/usr/bin/env python3
"""fuzz_parity.py -- random small frames (sizes, K, quality, content, both flavours) through the product against the oracle:
container bytes and decoded pixels.  A hunting tool, not a test: python tools/fuzz_parity.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import imageexperiments_amd as ia
    from oracle import oracle_py as O
    O.build(ref=False)
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctxs, octxs = {}, {}
    bad = 0
    for n in range(cases):
        K = int(rng.choice([1, 2, 5, 8, 13, 16, 24, 32]))
        bpp = float(rng.choice([0.0, 0.5, 1.0, 2.0, 3.5, 5.0, 8.0]))
        W, H = int(rng.integers(1, 200)), int(rng.integers(1, 160))
        kind = int(rng.integers(0, 5))
        if kind == 0:
            rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        elif kind == 1:
            rgb = np.full((H, W, 3), rng.integers(0, 256, 3), dtype=np.uint8)
        elif kind == 2:
            rgb = O.synth_frame(W, H, int(rng.integers(0, 1 << 30)))
        elif kind == 3:                                            # smooth gradients with a few edges
            y, x = np.mgrid[0:H, 0:W]
            rgb = np.stack([(x * 3 + y) % 256, (x // 8 * 40) % 256, (y * 5) % 256], -1).astype(np.uint8)
        else:                                                     # sparse specks on black
            rgb = np.zeros((H, W, 3), np.uint8)
            m = rng.random((H, W)) < 0.02
            rgb[m] = rng.integers(0, 256, (int(m.sum()), 3), dtype=np.uint8)
        fast = bool(rng.integers(0, 2))
        key = (K, bpp)
        if key not in ctxs:
            ctxs[key] = ia.create_compression_context(K, 8, bpp, device=0)
            octxs[key] = O.OracleContext(K, 8, bpp)
        ctx, octx = ctxs[key], octxs[key]
        ctx.set_fast(fast)
        want = (O.OracleFastContext(octx) if fast else octx).encode_image(rgb)
        got = ctx.encode_image(rgb)
        ok = got == want
        if ok:
            dec = ia.decode_image(got, ctx)
            ok = bool((dec == (O.decode_image_fast(got) if fast else O.decode_image(got))).all())
        if not ok:
            bad += 1
            print(f"MISMATCH case {n}: {W}x{H} K={K} bpp={bpp} kind={kind} fast={fast}", flush=True)
    print(f"{cases} cases, {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
