#!/usr/bin/env python3
"""multi_lane_timing.py -- mpc_encode_images_multi with N lanes on ONE device (every lane the real code: stripes, peer copies,
interleave, container jobs; the lanes' tile encodes take turns on the device): ms per frame and the golden bytes.  A rehearsal of
the multi-GPU driver's pipelining, not a multi-GPU measurement.

    python tools/multi_lane_timing.py [lanes] [frames]
"""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import imageexperiments_amd as ia
    from imageexperiments_amd import api
    from bench import synth_frame, WORKLOADS, golden_of
    lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    W, H, K, q = WORKLOADS["raise"]
    ctxs = [ia.create_compression_context(K, 8, q, device=0) for _ in range(lanes)]
    frame = synth_frame(W, H, 12345)
    frames = [frame] * n
    api.encode_images_multi(ctxs, frames[:2 * lanes])
    t0 = time.perf_counter()
    out = api.encode_images_multi(ctxs, frames, views=True)
    dt = (time.perf_counter() - t0) / n
    gold = golden_of("raise", q, 12345)
    ok = all(hashlib.sha256(np.ascontiguousarray(b).tobytes()).hexdigest() == gold[0] for b in out) if gold else None
    print(f"{lanes} lanes on one device, {n} frames from host memory: {dt * 1e3:.3f} ms per frame = {W * H / dt / 1e6:.0f} Mpix/s, containers golden: {ok}")


if __name__ == "__main__":
    main()
