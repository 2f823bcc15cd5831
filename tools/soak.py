#!/usr/bin/env python3
"""soak.py -- many frames through the frame pipeline, every container compared with the first (tuning / robustness aid).

    python tools/soak.py [raise|1080p|8k] [frames] [--host] [--fast]

Two alternating frames (seeds 12345, 12346) so that neighbouring slots hold different data; --host feeds them from host memory."""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import imageexperiments_amd as ia
    from bench import synth_frame, WORKLOADS
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    name = args[0] if args else "raise"
    n = int(args[1]) if len(args) > 1 else 200
    W, H, K, q = WORKLOADS[name][:4]
    ctx = ia.create_compression_context(K, 8, q, device=0)
    if "--fast" in sys.argv:
        ctx.set_fast(True)
    frames = [synth_frame(W, H, 12345), synth_frame(W, H, 12346)]
    want = [hashlib.sha256(ctx.encode_image(f)).hexdigest() for f in frames]
    t = time.perf_counter()
    if "--host" in sys.argv:
        blobs = ctx.encode_images([frames[i & 1] for i in range(n)], views=True)
    else:
        dev = [torch.from_numpy(f).cuda() for f in frames]
        blobs = ctx.encode_images_device([dev[i & 1].data_ptr() for i in range(n)], W, H, views=True)
    dt = time.perf_counter() - t
    bad = [i for i, b in enumerate(blobs) if hashlib.sha256(b).hexdigest() != want[i & 1]]
    print(f"{name}: {n} frames in {dt * 1e3:.0f} ms = {n * W * H / dt / 1e6:.0f} Mpix/s, {len(bad)} containers differ {bad[:8]}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
