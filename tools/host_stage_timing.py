#!/usr/bin/env python3
"""host_stage_timing.py -- the host entropy stage alone (mpc_assemble_symbol_streams) on a frame's real streams, by thread count.
    python tools/host_stage_timing.py [raise|1080p|8k]        (MPCODEC_LIB selects another build for A/B)
The records come from the device encoder; the streams are assembled here with numpy exactly as mp_streams.hip does."""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "raise"
    if len(sys.argv) > 2:                                     # child: time with the thread count of the environment
        import imageexperiments_amd as ia
        d = np.load(sys.argv[2])
        W, H, K = (int(v) for v in d["shape"])
        L = ia.load_library()
        L.mpc_assemble_symbol_streams.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
        quant, counts, symbols, off = (np.ascontiguousarray(d[k]) for k in ("quant", "counts", "symbols", "off"))

        def run():
            out, n = C.POINTER(C.c_uint8)(), C.c_size_t(0)
            assert L.mpc_assemble_symbol_streams(W, H, K, 8, quant.ctypes.data_as(C.POINTER(C.c_double)), counts.ctypes.data, symbols.ctypes.data,
                                                 off.ctypes.data, C.byref(out), C.byref(n)) == 0
            L.mpc_free(C.cast(out, C.c_void_p))
            return n.value
        run()
        run()
        t = time.perf_counter()
        for _ in range(8):
            nb = run()
        print(f"threads {os.environ.get('MPC_HOST_THREADS')}: {(time.perf_counter() - t) / 8 * 1e3:.2f} ms, {nb} bytes, {symbols.size + counts.size} symbols")
        return
    import imageexperiments_amd as ia
    from bench import synth_frame, WORKLOADS
    W, H, K, q = WORKLOADS[name]
    ctx = ia.create_compression_context(K, 8, q, device=0)
    counts, choices, _e, _s = ctx.encode_tiles(synth_frame(W, H, 12345))
    rec = choices.view(np.uint32).reshape(-1, 3, K)
    syms, off = [], [0]
    for ch in range(3):
        for i in range(K):
            live = counts[:, ch] > i
            dd = (rec[live, ch, i] & 0xFFFF).astype(np.uint16)
            cc = (rec[live, ch, i] >> 16).astype(np.int64)
            if i == 0:
                diff = np.diff(np.concatenate([[0], cc]))
                cc = (diff << 1) ^ (diff >> 63)
            for a in (dd, cc.astype(np.uint16)):
                syms.append(a)
                off.append(off[-1] + a.size)
    path = "/tmp/host_stage_streams.npz"
    np.savez(path, shape=np.array([W, H, K]), quant=ctx.quant.astype(np.float64), counts=counts.astype(np.uint16),
             symbols=np.concatenate(syms), off=np.array(off, np.uint64))
    for t in (1, 4, 8, 16):
        env = dict(os.environ, MPC_HOST_THREADS=str(t))
        subprocess.run([sys.executable, os.path.abspath(__file__), name, path], env=env, check=True)


if __name__ == "__main__":
    main()
