#!/usr/bin/env python3
"""hbm_bandwidth.py -- measured HBM copy / read bandwidth of the box next to the nominal 8 TB/s, and what rocminfo says."""
import subprocess
import time

import torch

n = 4 << 30
a = torch.empty(n, dtype=torch.uint8, device="cuda")
b = torch.empty(n, dtype=torch.uint8, device="cuda")
a.fill_(1)
for name, fn, bytes_moved in (("copy (read + write)", lambda: b.copy_(a), 2 * n), ("fill (write)", lambda: b.fill_(3), n),
                              ("sum (read)", lambda: a.view(torch.int64).sum(), n)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    reps = 20
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    print(f"{name}: {bytes_moved / dt / 1e12:.2f} TB/s ({n >> 30} GiB buffers, {dt * 1e3:.2f} ms)")
try:
    out = subprocess.run(["/opt/rocm/bin/rocminfo"], capture_output=True, text=True, timeout=60).stdout
    gpu = out[out.index("gfx950") - 2000:] if "gfx950" in out else out
    for key in ("Marketing Name", "Compute Unit", "Max Clock Freq", "Wavefront Size", "Max Waves Per CU", "Cacheline Size"):
        for line in gpu.splitlines():
            if key in line:
                print("rocminfo:", " ".join(line.split()))
                break
except Exception as e:          # noqa: BLE001
    print("rocminfo unavailable:", e)
