#!/usr/bin/env python3
"""chain_bench.py -- the kernels BEHIND the pursuit (stream assembly, entropy phases 1 and 2) on their own: one workload's
records are made once, then `mpc_records_to_container_device` runs `reps` times on an otherwise idle device.  Under
`rocprofv3 --kernel-trace --stats` this gives every small kernel's duration without a pursuit beside it (tuning aid: in the
pipeline these chains are what sits in the gap between two pursuits).

    python tools/chain_bench.py [raise|1080p|8k] [reps]
"""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import imageexperiments_amd as ia
    from bench import synth_frame, WORKLOADS
    name = sys.argv[1] if len(sys.argv) > 1 else "raise"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    W, H, K, q = WORKLOADS[name][:4]
    ctx = ia.create_compression_context(K, 8, q, device=0)
    rgb = synth_frame(W, H, 12345)
    tiles = ((W + 7) // 8) * ((H + 7) // 8)
    d_rgb = torch.from_numpy(rgb).cuda()
    d_counts = torch.zeros((tiles, 3), dtype=torch.int16, device="cuda")
    d_choices = torch.zeros((tiles, 3, K), dtype=torch.int32, device="cuda")
    ctx.reserve(tiles)
    stream = torch.cuda.current_stream()
    ctx.encode_batch_device(d_rgb.data_ptr(), 1, W * H * 3, W, H, W * 3, 0, (H + 7) // 8, d_counts.data_ptr(), d_choices.data_ptr(),
                            stream=stream.cuda_stream)
    torch.cuda.synchronize()
    blob = None
    for _ in range(2):
        blob = ctx.records_to_container_device(d_counts.data_ptr(), d_choices.data_ptr(), W, H, stream=stream.cuda_stream)
    t0 = time.perf_counter()
    for _ in range(reps):
        blob = ctx.records_to_container_device(d_counts.data_ptr(), d_choices.data_ptr(), W, H, stream=stream.cuda_stream)
    dt = (time.perf_counter() - t0) / reps
    print(f"{name}: records -> container {dt * 1e3:.3f} ms per frame (host tables included), {len(blob)} bytes, "
          f"sha1 {hashlib.sha1(bytes(blob)).hexdigest()[:16]}")


if __name__ == "__main__":
    main()
