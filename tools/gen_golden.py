#!/usr/bin/env python3
"""gen_golden.py -- whole-frame golden fixtures made by the ORACLE (test infrastructure), in the build container.

    python tools/gen_golden.py [--only NAME ...] [--workers N]

Runs oracle/mpo_*.c (the plain-C restatement of the reference's double path) over every BASELINE.json frame,
multi-process over tile columns, assembles the container with the oracle's own writer and records

    sha256 / size of the container, sha256 of the records, sum of counts per channel, length of every stream

in tests/golden/frames.json.  `-m gpu` tests assert the product's bytes against these (tests/test_gpu_golden_frames.py).
Inputs are generated (BASELINE.md generator) or derived from fixtures under tests/golden/, never read from
/root/reference at test time:
  * synthetic frames: std::mt19937(seed) generator of SURVEY 8(d);
  * natural_mn:  the reference's own bitstream Data/r0c1de5e1t_3_5.mn decoded (oracle decoder here, product decoder
    in the GPU test -- pixel-exact to each other, tests/test_gpu_parity.py) and re-encoded: 16 Mpixel of natural content;
  * natural_jpg: Data/r0c1de5e1t.jpg (fixture copy) decoded by PIL -- BASELINE configs[0]'s stand-in per SURVEY 8(c);
    the decoded pixels' sha256 is recorded and the GPU test skips when the box's libjpeg decodes differently.
"""
import argparse
import hashlib
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

FRAMES = [
    # name, kind, W, H, K, quality, seed
    ("1080p_k8_q3.5", "synthetic", 1920, 1080, 8, 3.5, 12345),
    ("8k_k16_q3.5", "synthetic", 7680, 4320, 16, 3.5, 12345),
    ("natural_mn_k32_q3.5", "mn", 4928, 3264, 32, 3.5, 0),
    ("natural_jpg_k32_q3.5", "jpg", 4928, 3264, 32, 3.5, 0),
    ("odd_1003x517_k32_q3.5", "synthetic", 1003, 517, 32, 3.5, 777),      # ragged edges in both directions
] + [(f"batch_frame{f}_k32_q3.5", "synthetic", 4928, 3264, 32, 3.5, 12345 + f) for f in range(1, 8)     # configs[3]: frame f uses seed 12345 + f
] + [(f"raise_k32_q{q:.1f}", "synthetic", 4928, 3264, 32, q, 12345) for q in (2.0, 2.5, 3.0, 3.5, 4.0, 4.5, 5.0, 5.5, 6.0)] + [
    # the `...Fast` (float) flavour, by oracle/mpo_fast.c: a DEFINITION of the float mode, parity unpinned against the reference
    ("fast_1080p_k8_q3.5", "synthetic", 1920, 1080, 8, 3.5, 12345),
    ("fast_raise_k32_q3.5", "synthetic", 4928, 3264, 32, 3.5, 12345),
    ("fast_natural_mn_k32_q3.5", "mn", 4928, 3264, 32, 3.5, 0),
]

_ctx = None
_rgb = None
_keep = None


def load_frame(kind, W, H, seed):
    from oracle import oracle_py as O
    if kind == "synthetic":
        return O.synth_frame(W, H, seed)
    if kind == "mn":
        with open(os.path.join(GOLDEN, "r0c1de5e1t_3_5.mn"), "rb") as f:
            return O.decode_image(f.read())
    if kind == "jpg":
        from PIL import Image
        return np.ascontiguousarray(np.asarray(Image.open(os.path.join(GOLDEN, "r0c1de5e1t.jpg")).convert("RGB")))
    raise ValueError(kind)


def _init(K, q, shm_name, shape, fast=False):
    global _ctx, _rgb, _keep
    from multiprocessing import shared_memory
    from oracle import oracle_py as O
    _ctx = O.OracleContext(K, 8, q)
    if fast:
        _keep = _ctx                                     # the float copies refer to the double context's tables
        _ctx = O.OracleFastContext(_ctx)
    shm = shared_memory.SharedMemory(name=shm_name)
    _rgb = (shm, np.ndarray(shape, np.uint8, buffer=shm.buf))


def _work(rng):
    a, b = rng
    rgb = _rgb[1]
    ty = (rgb.shape[0] + 7) // 8
    counts, delta, coef, _e, swept = _ctx.encode_tiles(rgb, tx_begin=a, tx_end=b)
    sl = slice(a * ty, b * ty)
    return a, counts[sl].copy(), delta[sl].copy(), coef[sl].copy(), swept[sl].copy()


def encode_frame(rgb, K, q, workers, fast=False):
    """-> (container bytes, counts[T][3], delta[T][3][K], coef[T][3][K], swept[T][3]) by the oracle, column-parallel."""
    from multiprocessing import shared_memory
    from oracle import oracle_py as O
    H, W = rgb.shape[:2]
    tx, ty = (W + 7) // 8, (H + 7) // 8
    shm = shared_memory.SharedMemory(create=True, size=rgb.nbytes)
    try:
        np.ndarray(rgb.shape, np.uint8, buffer=shm.buf)[:] = rgb
        step = max(1, tx // (workers * 6))
        ranges = [(a, min(a + step, tx)) for a in range(0, tx, step)]
        with mp.get_context("fork").Pool(workers, initializer=_init, initargs=(K, q, shm.name, rgb.shape, fast)) as pool:
            parts = pool.map(_work, ranges, chunksize=1)
    finally:
        shm.close()
        shm.unlink()
    parts.sort(key=lambda p: p[0])
    counts = np.concatenate([p[1] for p in parts])
    delta = np.concatenate([p[2] for p in parts])
    coef = np.concatenate([p[3] for p in parts])
    swept = np.concatenate([p[4] for p in parts])
    octx = O.OracleContext(K, 8, q)
    codes = []
    for ch in range(3):
        for i in range(K):
            live = counts[:, ch] > i
            codes.append(delta[live, ch, i])
            codes.append(coef[live, ch, i])
    blob = O.write_compressed(dict(W=W, H=H, K=K, bs=8, quant=octx.quant.astype(np.uint16), lengths=counts.reshape(-1), codes=codes))
    return blob, counts, delta, coef, swept, [int(c.size) for c in codes]


def records_sha(counts, delta, coef):
    """sha256 over counts[T][3] u16 then the packed LIVE records [T][3][K] u32 (lo16 deltaId, hi16 intCoeff; entries
    at and beyond `count` zeroed: the terminating record never reaches the container)."""
    K = delta.shape[2]
    rec = delta.astype(np.uint32) | (coef.astype(np.uint32) << 16)
    step = np.arange(K)[None, None, :]
    rec = np.where(step < counts[:, :, None], rec, 0).astype(np.uint32)
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(counts, np.uint16).tobytes())
    h.update(np.ascontiguousarray(rec).tobytes())
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    ap.add_argument("--workers", type=int, default=os.cpu_count() or 4)
    args = ap.parse_args()
    from oracle import oracle_py as O
    O.build(ref=False)
    path = os.path.join(GOLDEN, "frames.json")
    out = {}
    if os.path.exists(path):
        with open(path) as f:
            out = json.load(f)
    for name, kind, W, H, K, q, seed in FRAMES:
        if args.only and name not in args.only:
            continue
        t0 = time.time()
        rgb = load_frame(kind, W, H, seed)
        assert rgb.shape == (H, W, 3), rgb.shape
        fast = name.startswith("fast_")
        blob, counts, delta, coef, swept, lens = encode_frame(rgb, K, q, args.workers, fast)
        out[name] = {
            "kind": kind, "width": W, "height": H, "K": K, "quality": q, "seed": seed, "flavour": "fast" if fast else "double",
            "rgb_sha256": hashlib.sha256(rgb.tobytes()).hexdigest(),
            "container_sha256": hashlib.sha256(blob).hexdigest(),
            "container_bytes": len(blob),
            "records_sha256": records_sha(counts, delta, coef),
            "sum_counts": [int(counts[:, ch].sum()) for ch in range(3)],
            "swept_rows": int(swept.astype(np.int64).sum()),
            "stream_lengths": [int(counts.size)] + lens,
        }
        print(f"{name}: {len(blob)} B, counts {out[name]['sum_counts']}, {time.time() - t0:.1f} s", flush=True)
        with open(path, "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)
            f.write("\n")


if __name__ == "__main__":
    main()
