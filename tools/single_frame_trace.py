#!/usr/bin/env python3
"""single_frame_trace.py -- one frame at a time through mpc_encode_image_device with MPC_TRACE=1 (latency breakdown, tuning aid)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MPC_TRACE"] = "1"


def main():
    import torch
    import imageexperiments_amd as ia
    from bench import synth_frame, WORKLOADS
    name = sys.argv[1] if len(sys.argv) > 1 else "raise"
    W, H, K, q = WORKLOADS[name][:4]
    ctx = ia.create_compression_context(K, 8, q, device=0)
    rgb = synth_frame(W, H, 12345)
    d = torch.from_numpy(rgb).cuda()
    for i in range(5):
        t = time.perf_counter()
        blob = ctx.encode_images_device([d.data_ptr()], W, H, views=True)[0]
        dt = time.perf_counter() - t
        print(f"single frame {i}: {dt * 1e3:.2f} ms = {W * H / dt / 1e6:.0f} Mpix/s ({len(blob)} bytes)", file=sys.stderr)
    for i in range(3):
        t = time.perf_counter()
        blob = ctx.encode_image(rgb, view=True)
        dt = time.perf_counter() - t
        print(f"single frame from host memory {i}: {dt * 1e3:.2f} ms = {W * H / dt / 1e6:.0f} Mpix/s", file=sys.stderr)


if __name__ == "__main__":
    main()
