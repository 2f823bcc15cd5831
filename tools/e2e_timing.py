"""End-to-end timings on one GPU, host buffers in and out (steady state: second call of each):
   python tools/e2e_timing.py"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import imageexperiments_amd as ia


def timed(f, reps=3):
    f()
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter()
        r = f()
        best = min(best, time.perf_counter() - t)
    return r, 1e3 * best


for name, (W, H, K, q) in bench.WORKLOADS.items():
    rgb = bench.synth_frame(W, H, 12345)
    ctx = ia.create_compression_context(K, 8, q, device=0)
    (counts, choices, en, sw), t_tiles = timed(lambda: ctx.encode_tiles(rgb))
    blob, t_host = timed(lambda: ia.assemble_streams(W, H, K, 8, ctx.quant, counts, choices.view(np.uint32)))
    blob2, t_image = timed(lambda: ctx.encode_image(rgb, view=True))      # the library's buffer as it is (no Python copy)
    assert bytes(blob) == bytes(blob2)
    frames = [rgb] * 16
    blobs, t_pipe = timed(lambda: ctx.encode_images(frames, views=True), reps=2)       # the library's buffers as they are
    assert bytes(blobs[-1]) == bytes(blob)
    img, t_dec = timed(lambda: ia.decode_image(blob, ctx))
    print(f"{name}: encode_tiles (H2D + pursuit + D2H) {t_tiles:.1f} ms | entropy stage by the host-only route (assemble_streams) {t_host:.1f} ms | "
          f"encode_image (RGB -> container bytes) {t_image:.1f} ms = {W * H / t_image / 1e3:.0f} Mpix/s | encode_images x16 pipelined "
          f"{t_pipe / 16:.1f} ms/frame = {16 * W * H / t_pipe / 1e3:.0f} Mpix/s | {len(blob)} bytes, "
          f"{8 * len(blob) / (W * H):.3f} bpp | decode_image {t_dec:.1f} ms | psnr {ia.calculate_psnr(rgb, img):.2f}")
    ctx.close()
