#!/usr/bin/env python3
"""decode_trace.py -- mpc_decode_image with MPC_TRACE=1 on the reference's .mn fixture and on a synthetic 16 Mpixel frame (tuning aid)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MPC_TRACE"] = "1"


def main():
    import imageexperiments_amd as ia
    from bench import synth_frame
    ctx = ia.create_compression_context(32, 8, 3.5, device=0)
    with open(os.path.join(ROOT, "tests", "golden", "r0c1de5e1t_3_5.mn"), "rb") as f:
        mn = f.read()
    synth = ctx.encode_image(synth_frame(4928, 3264, 12345))
    for name, blob in (("reference .mn (3.7 MB)", mn), ("synthetic 4928x3264 (6.6 MB)", synth)):
        for i in range(4):
            t = time.perf_counter()
            ia.api.decode_image(blob, ctx)
            print(f"{name}: decode_image {1e3 * (time.perf_counter() - t):.2f} ms", file=sys.stderr)


if __name__ == "__main__":
    main()
