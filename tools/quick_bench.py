#!/usr/bin/env python3
"""quick_bench.py -- device-stage timing of one workload (tuning aid; bench.py is the measured benchmark).

    python tools/quick_bench.py [raise|1080p|8k|natural] [reps]
"""
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
    name = sys.argv[1] if len(sys.argv) > 1 else "raise"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    if name == "natural":
        W, H, K, q = 4928, 3264, 32, 3.5
        ctx = ia.create_compression_context(K, 8, q, device=0)
        with open(os.path.join(ROOT, "tests", "golden", "r0c1de5e1t_3_5.mn"), "rb") as f:
            rgb = ia.api.decode_image(f.read(), ctx)
    else:
        W, H, K, q = WORKLOADS[name][:4]
        ctx = ia.create_compression_context(K, 8, q, device=0)
        rgb = synth_frame(W, H, 12345)
    tiles = ((W + 7) // 8) * ((H + 7) // 8)
    d_rgb = torch.from_numpy(rgb).cuda()
    d_counts = torch.zeros((tiles, 3), dtype=torch.int16, device="cuda")
    d_choices = torch.zeros((tiles, 3, K), dtype=torch.int32, device="cuda")
    d_swept = torch.zeros((tiles, 3), dtype=torch.int32, device="cuda")
    ctx.reserve(tiles)
    stream = torch.cuda.current_stream()

    def run():
        ctx.encode_batch_device(d_rgb.data_ptr(), 1, W * H * 3, W, H, W * 3, 0, (H + 7) // 8, d_counts.data_ptr(),
                                d_choices.data_ptr(), 0, d_swept.data_ptr(), stream=stream.cuda_stream)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    steps = int((d_counts.to(torch.int64).clamp(max=K - 1) + 1).sum().item())
    print(f"{name}: {dt * 1e3:.3f} ms/frame, {W * H / dt / 1e6:.1f} Mpix/s, {steps} tile-channel-steps, "
          f"{dt / steps * 1e9 * 1024 * 2.4:.0f} SIMD-cycles per tile-channel-step (at 2.4 GHz), path={os.environ.get('MPC_PATH', 'persistent')}")


if __name__ == "__main__":
    main()
