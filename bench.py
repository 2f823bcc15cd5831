#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X tile encoder (BASELINE.json metric: encode Mpixels/s at quality 3.5).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload raise|1080p|8k] [--quality Q] [--no-cpu]

A *step* is one pass of the hot path over one batch of synthetic frames that are already resident in HBM:
RGB in HBM -> tile encode (mp_pursuit_kernel, one launch) -> stream assembly and the per-symbol work of the entropy stage on the
device, one code table per stream on the host -> container bytes in host memory (byte-identical to the reference's encodeImage:
after the timed region the containers are hashed against tests/golden/frames.json, `bytes_match_golden`).
N = 1: one frame per step; default workload = the north_star's: 4928x3264 synthetic RGB, K = 32, quality 3.5; consecutive steps
take DISTINCT frames (seeds 12345 + f, a cycle of up to 8).  The K steps of the timed region go through the library's frame
pipeline in one call (mpc_encode_images_device): the host's table building of step i overlaps the device work of step i + 1, as
in any steady-state use.  Beside `value` the line carries SURVEY 8(d)'s PCIe-inclusive figures, measured after the timed region:
`host_to_bytes_Mpix_s` (pageable host RGB -> container bytes through mpc_encode_images, the same frames, pipelined) and
`single_frame_ms` (one mpc_encode_image call).
N > 1: one rank per GPU, launched by torch.distributed.run -- or by this script itself: `python bench.py --gpus N` without a
launcher starts N child processes (fresh interpreters, before anything here touches a GPU) with RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* set, relays rank 0's line and fails if any rank fails.  Weak scaling, N frames per step.  Every frame is row-striped over
the N ranks (rank r encodes tile-row stripe r of every frame in one launch); the stripes' records then travel to the frame's owner
(rank f owns frame f: batched point-to-point over RCCL/xGMI), which puts them into the reference's tile order and produces frame
f's container.  No other collective: see DESIGN.md 7 for why the symbol-histogram all-reduce is not in it.
Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (W, H, K, quality)   -- BASELINE.json configs
    "1080p": (1920, 1080, 8, 3.5),     # configs[1]
    "raise": (4928, 3264, 32, 3.5),    # configs[2] at the quality the metric and the north_star name
    "8k": (7680, 4320, 16, 3.5),       # configs[4]
}
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BF16_MFMA_PEAK_TFLOPS = 2516.8         # MI355X_MICROARCH.md: dense bf16 MFMA = 16 x the 157.3 TF f32 rate
MFMA_FLOP = 2 * 16 * 16 * 32           # one v_mfma_f32_16x16x32_bf16


def synth_frame(W, H, seed):
    """BASELINE.md section 3 generator (std::mt19937(seed), raster order, one draw per pixel), vectorised:
    numpy's MT19937 seeded through the legacy `init_genrand` path yields the same 32-bit stream."""
    bg = np.random.MT19937()
    state = bg.state
    key = np.empty(624, np.uint32)
    key[0] = seed & 0xFFFFFFFF
    for i in range(1, 624):
        key[i] = (1812433253 * (int(key[i - 1]) ^ (int(key[i - 1]) >> 30)) + i) & 0xFFFFFFFF
    state["state"]["key"] = key
    state["state"]["pos"] = 624
    bg.state = state
    draws = bg.random_raw(W * H).astype(np.int64).reshape(H, W)
    n = (draws % 32) - 16
    x = np.arange(W, dtype=np.int64)[None, :]
    y = np.arange(H, dtype=np.int64)[:, None]
    out = np.empty((H, W, 3), np.uint8)
    out[..., 0] = np.clip(x * 255 // W + n, 0, 255)
    out[..., 1] = np.clip(y * 255 // H + n, 0, 255)
    out[..., 2] = np.clip(128 + 3 * n, 0, 255)
    return out


# ---- cpu_baseline: the oracle (plain-C port of the reference's double path), one core and all cores ----------------------
_cpu = {}


def _cpu_init(K, q, shm_name, shape):
    from multiprocessing import shared_memory
    from oracle import oracle_py as O
    shm = shared_memory.SharedMemory(name=shm_name)
    _cpu["ctx"] = O.OracleContext(K, 8, q)
    _cpu["shm"] = shm
    _cpu["rgb"] = np.ndarray(shape, np.uint8, buffer=shm.buf)


def _cpu_work(rng):
    _cpu["ctx"].encode_tiles(_cpu["rgb"], tx_begin=rng[0], tx_end=rng[1])
    return rng[1] - rng[0]


def host_cores():
    """Cores this process may really use: the affinity mask, cut down to the cgroup's CPU quota where there is one (a GPU
    box hands a one-GPU job a share of the host's cores, not all of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return min(n, 64)


def cpu_baseline(W, H, K, q, frame, seconds):
    """Bounded sample of the same frame: tile columns 0..c-1 on ONE core, then tile columns on ALL host cores (one process per
    core, tile-column parallel), each sized for about `seconds` of work from a short calibration."""
    import multiprocessing as mp
    from multiprocessing import shared_memory
    from oracle import oracle_py as O
    O.build(ref=False)
    octx = O.OracleContext(K, 8, q)
    tiles_x, tiles_y = (W + 7) // 8, (H + 7) // 8
    t0 = time.perf_counter()
    octx.encode_tiles(frame, tx_begin=0, tx_end=2)
    per_col = (time.perf_counter() - t0) / 2
    c1 = int(max(2, min(tiles_x, seconds / per_col)))
    t0 = time.perf_counter()
    octx.encode_tiles(frame, tx_begin=0, tx_end=c1)
    dt1 = time.perf_counter() - t0
    one = c1 * tiles_y * 64 / dt1 / 1e6
    cores = host_cores()
    call = int(max(cores, min(tiles_x, cores * seconds / per_col)))
    shm = shared_memory.SharedMemory(create=True, size=frame.nbytes)
    try:
        np.ndarray(frame.shape, np.uint8, buffer=shm.buf)[:] = frame
        step = max(1, call // (cores * 4))
        ranges = [(a, min(a + step, call)) for a in range(0, call, step)]
        with mp.get_context("fork").Pool(cores, initializer=_cpu_init, initargs=(K, q, shm.name, frame.shape)) as pool:
            pool.map(_cpu_work, [(0, 1)] * cores)                          # contexts built, pages touched
            t0 = time.perf_counter()
            pool.map(_cpu_work, ranges, chunksize=1)
            dta = time.perf_counter() - t0
    finally:
        shm.close()
        shm.unlink()
    allc = call * tiles_y * 64 / dta / 1e6
    return {"value": round(allc, 5), "unit": "Mpixels/s", "cores": cores, "kind": "port",
            "sample": f"tile columns 0..{call - 1} of the same {W}x{H} frame over {cores} processes ({call * tiles_y} tiles, {dta:.1f} s)",
            "one_core": {"value": round(one, 5), "cores": 1, "sample": f"tile columns 0..{c1 - 1} ({c1 * tiles_y} tiles, {dt1:.1f} s)"},
            "note": "oracle/mpo_*.c = C restatement of the reference's double path (tile encode only, no entropy stage) without its "
                    "per-step dictionary copy: about twice as fast as the reference itself (0.0707 Mpix/s on one core of the dev "
                    "container, SURVEY 6); the reference is single-threaded, its Eigen/float path is not buildable here"}


def stripe_bounds(tiles_y, n, r):
    """contiguous tile-row stripes, remainder to the first ranks (540 -> 68x4 + 67x4)."""
    base, rem = divmod(tiles_y, n)
    begin = r * base + min(r, rem)
    return begin, begin + base + (1 if r < rem else 0)


def golden_of(workload, q, seed):
    """sha256 / size of the oracle's container for this synthetic frame (tests/golden/frames.json), or None."""
    try:
        with open(os.path.join(ROOT, "tests", "golden", "frames.json")) as f:
            frames = json.load(f)
    except Exception:
        return None
    W, H, K, _ = WORKLOADS[workload]
    for rec in frames.values():
        if (rec.get("kind") == "synthetic" and rec.get("flavour", "double") == "double" and rec["width"] == W and rec["height"] == H
                and rec["K"] == K and abs(rec["quality"] - q) < 1e-9 and rec["seed"] == seed):
            return rec["container_sha256"], rec["container_bytes"]
    return None


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: N fresh child processes, one rank each, started BEFORE this process has touched a
    GPU (it never does: it only relays).  Rank 0's stdout is passed on; a failing rank ends the others and the run."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    live = set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                for o in live:
                    procs[o].terminate()
        time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="raise", choices=sorted(WORKLOADS))
    ap.add_argument("--quality", type=float, default=None, help="bpp allocation (BASELINE configs[2] sweeps 2.0 .. 6.0); default 3.5")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--fast", action="store_true",
                    help="the `...Fast` (float) flavour of the tile path (SURVEY 8f N3): NOT the driver's line -- parity unpinned "
                         "against the reference, bit-identical to oracle/mpo_fast.c only")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="target duration of each CPU sample")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo + --share-device rehearses the N>1 path on a one-GPU box")
    ap.add_argument("--share-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--no-e2e", action="store_true", help="skip the untimed PCIe-inclusive legs (host_to_bytes_Mpix_s, single_frame_ms)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))            # nothing above has initialised a GPU

    import torch
    import torch.distributed as dist
    import imageexperiments_amd as ia
    from imageexperiments_amd import sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit(f"bench.py --gpus {args.gpus} inside a group of {world} ranks")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the hot path has no CPU fallback")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # RCCL over xGMI
        else:
            dist.init_process_group("gloo")

    W, H, K, q = WORKLOADS[args.workload]
    if args.quality is not None:
        q = args.quality
    ctx = ia.create_compression_context(K, 8, q, device=local_rank)
    if args.fast:
        ctx.set_fast(True)
    tiles_x, tiles_y = (W + 7) // 8, (H + 7) // 8
    # N = 1: consecutive steps take distinct frames (a cycle of `frames`); N > 1: the N frames of a step (weak scaling)
    frames = world if world > 1 else max(1, min(args.steps, 8))
    host_frames = np.stack([synth_frame(W, H, 12345 + f) for f in range(frames)])
    d_rgb = torch.from_numpy(host_frames).cuda()
    stream = torch.cuda.current_stream()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world == 1:
        ptrs = [d_rgb[f].data_ptr() for f in range(frames)]

        def run(n):                                          # n steps = n frames through the pipeline, in one call
            return ctx.encode_images_device([ptrs[i % frames] for i in range(n)], W, H, views=True)    # the library's buffers as they are
    else:
        striped = sharding.StripedEncoder(ctx, W, H, frames, world, rank, args.backend)

        def run(n):                                          # n steps, software-pipelined on the rank's stream
            return striped.run(d_rgb, stream, n, views=True)

    # HIP events around every launch of the dominant kernel IN the timed region, on the stream it is launched on (the library's
    # pursuit stream): roofline.achieved below is priced on these durations.  (On during the warm-up too: the events exist then.)
    timed_events = os.environ.get("BENCH_TIMED_EVENTS", "1") != "0"
    if timed_events:
        ctx.kernel_timing(True)
    run(args.warmup)
    fence()
    if timed_events:
        ctx.kernel_timing(True)                              # counters and event list back to zero
    t0 = time.perf_counter()
    containers = run(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    timed_kern_ms = timed_kern_launches = timed_mfma = 0
    if timed_events:
        timed_kern_ms, timed_kern_launches, _ = ctx.read_kernel_timing()
        timed_mfma, _ = ctx.read_kernel_counters()
        ctx.kernel_timing(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    container_bytes = len(containers[-1])

    # ---- untimed: the bytes of the timed region against the oracle's (tests/golden/frames.json) --------------------------------
    # N = 1: container i is frame i % frames (seed 12345 + i % frames); N > 1: every container of this rank is frame `rank`
    checked, matched = 0, 0
    if not args.fast:
        todo = {}
        for i in range(len(containers)):
            todo.setdefault(rank if world > 1 else i % frames, i)
        for f, i in todo.items():
            gold = golden_of(args.workload, q, 12345 + f)
            if gold is None:
                continue
            checked += 1
            blob = np.ascontiguousarray(containers[i])
            matched += int(len(blob) == gold[1] and hashlib.sha256(blob.tobytes()).hexdigest() == gold[0])
        # ... and every repetition of a frame equals its first container
        for i in range(len(containers)):
            first = todo[rank if world > 1 else i % frames]
            if i != first and not np.array_equal(containers[i], containers[first]):
                matched = -1
                break
    if world > 1:
        v = torch.tensor([checked, matched if matched >= 0 else -10 ** 6], dtype=torch.int64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(v)
        checked, matched = int(v[0].item()), int(v[1].item())
    bytes_match = None if checked == 0 else bool(matched == checked)

    # ---- untimed: SURVEY 8(d)'s PCIe-inclusive figures: pageable host RGB -> container bytes ---------------------------------
    host_to_bytes = single_ms = None
    if world == 1 and not args.no_e2e:
        seq = [host_frames[i % frames] for i in range(args.steps)]
        ctx.encode_images(seq[:min(len(seq), 8)], views=True)                 # slots and staging allocated
        t0 = time.perf_counter()
        out = ctx.encode_images(seq, views=True)
        host_to_bytes = W * H * len(seq) / (time.perf_counter() - t0) / 1e6
        del out
        ctx.encode_image(host_frames[0], view=True)
        singles = []
        for _ in range(5):
            t0 = time.perf_counter()
            ctx.encode_image(host_frames[0], view=True)                     # the library's buffer as it is
            singles.append(time.perf_counter() - t0)
        single_ms = sorted(singles)[len(singles) // 2] * 1e3

    # ---- untimed: the device stage alone, with HIP events around every launch of the dominant kernel -------------------------
    row_begin, row_end = stripe_bounds(tiles_y, world, rank)
    stage_frames = world if world > 1 else 1
    tiles = stage_frames * tiles_x * (row_end - row_begin)
    d_counts = torch.zeros((tiles, 3), dtype=torch.int16, device="cuda")
    d_choices = torch.zeros((tiles, 3, K), dtype=torch.int32, device="cuda")
    d_swept = torch.zeros((tiles, 3), dtype=torch.int32, device="cuda")
    dev_steps = max(3, min(args.steps, 10))

    def device_stage():
        ctx.encode_batch_device(d_rgb.data_ptr(), stage_frames, W * H * 3, W, H, W * 3, row_begin, row_end, d_counts.data_ptr(),
                                d_choices.data_ptr(), 0, d_swept.data_ptr(), stream=stream.cuda_stream)
    device_stage()
    torch.cuda.synchronize()
    ctx.kernel_timing(True)
    t0 = time.perf_counter()
    for _ in range(dev_steps):
        device_stage()
    torch.cuda.synchronize()
    dev_elapsed = (time.perf_counter() - t0) / dev_steps
    kern_ms_total, kern_launches, kern_busy_ms = ctx.read_kernel_timing()
    mfma_instr, tc_steps = ctx.read_kernel_counters()
    ctx.kernel_timing(False)
    swept_total = int(d_swept.to(torch.int64).sum().item())

    if rank == 0:
        pixels_per_step = frames * W * H if world > 1 else W * H
        value = pixels_per_step * args.steps / elapsed / 1e6
        launches_per_step = kern_launches // dev_steps
        standalone_flops = mfma_instr * MFMA_FLOP / max(kern_launches, 1)
        standalone_ms = kern_ms_total / max(kern_launches, 1)
        standalone_tflops = standalone_flops / (standalone_ms * 1e-3) / 1e12
        # the roofline's duration: the launches of the timed region (N = 1: the pipeline's, on 7/8 of the CUs with the small
        # kernels of the other frames beside them); the stand-alone leg (all CUs, nothing else on the device) is kept beside it
        if timed_kern_launches > 0:
            flops_per_launch = timed_mfma * MFMA_FLOP / timed_kern_launches
            avg_ms = timed_kern_ms / timed_kern_launches
        else:
            flops_per_launch, avg_ms = standalone_flops, standalone_ms
        mfma_tflops = flops_per_launch / (avg_ms * 1e-3) / 1e12
        mfma_frac = mfma_tflops / BF16_MFMA_PEAK_TFLOPS
        # fabric traffic of the same kernel from separate rocprofv3 --pmc passes (tools/profile_round.sh), newest round first
        traffic = traffic_src = fetch_raw = write_raw = None
        wait_note = None
        for tag in ("r03", "r02"):
            pmc = os.path.join(ROOT, "profiles", f"{tag}_pmc_{args.workload}.json")
            if args.quality is None and world == 1 and not args.fast and os.path.exists(pmc):
                try:
                    rec = json.load(open(pmc)).get("mp_pursuit_kernel", {})
                    fetch_raw, write_raw = rec.get("raw_fetch_bytes_per_launch"), rec.get("write_bytes_per_launch")
                    if fetch_raw and write_raw:
                        traffic = fetch_raw + write_raw
                        traffic_src = "profiles/" + os.path.basename(pmc)
                        sq = os.path.join(ROOT, "profiles", f"{tag}_pmc_sq_{args.workload}.json")
                        if os.path.exists(sq):
                            k = json.load(open(sq))["per_kernel"]["mp_pursuit_kernel"]
                            wait_note = (f"waves parked on memory {100 * k['SQ_WAIT_ANY_share_of_wave_cycles']:.0f} %, issue-stalled "
                                         f"{100 * k['SQ_WAIT_INST_ANY_share_of_wave_cycles']:.0f} %, issuing "
                                         f"{100 * k['SQ_ACTIVE_INST_ANY_share_of_wave_cycles']:.0f} % of their cycles (profiles/{os.path.basename(sq)})")
                        break
                except Exception:
                    traffic = None
        sec = avg_ms * 1e-3
        roof = {
            # The kernel is bound by per-step latency and instruction issue, not by a roof; the exactly measured utilisation is the
            # matrix pipe's.  The contract's two roofs, both reported: frac = executed MFMA flops / dense bf16 peak; the memory side
            # as fabric_frac_* (L2-side request bytes, Infinity-Cache hits included -- NOT HBM bytes).
            "bound": "latency", "nearest_roof": "mfma",
            "achieved": round(mfma_tflops, 2), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(mfma_frac, 4),
            "traffic": traffic, "traffic_source": traffic_src,
            "traffic_note": "FETCH_SIZE + WRITE_SIZE per launch of the same kernel, as counted, from two separate rocprofv3 --pmc passes of "
                            "tools/quick_bench.py at the commit of that file (not this run): the L2's memory-side requests, "
                            "Infinity-Cache hits included; the dictionary tables it re-reads (25 MB of operand tiles, Gram rows) live there",
            "fabric_frac_raw": round(traffic / sec / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
            "fabric_frac_2x": round((2 * fetch_raw + write_raw) / sec / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
            "limited_by": wait_note,
            "kernel": "mp_pursuit_kernel", "kernel_avg_ms": round(avg_ms, 5),
            "kernel_avg_ms_source": ("HIP events around the %d launches of the timed region, on the library's pursuit stream" % timed_kern_launches)
                                    if timed_kern_launches > 0 else "the stand-alone leg",
            "kernel_launches_per_step": launches_per_step,
            "standalone": {"kernel_avg_ms": round(standalone_ms, 5), "achieved": round(standalone_tflops, 2),
                           "frac": round(standalone_tflops / BF16_MFMA_PEAK_TFLOPS, 4),
                           "what": "the same kernel alone on the device, all CUs (untimed leg, HIP events on the launch stream)"},
            "kernel_busy_ms_per_step": round(kern_busy_ms / dev_steps, 4),
            "executed_mfma_flops_per_launch": int(flops_per_launch),
            "tile_channel_steps_per_step": tc_steps // dev_steps,
            "simd_cycles_per_tile_channel_step_at_2p4GHz": round(dev_elapsed / max(tc_steps // dev_steps, 1) * 1024 * 2.4e9, 1),
            # SURVEY 8(d)'s algorithmic figure, kept for reference only: 64 * sizeof(double) per dictionary row the reference
            # correlates, over the device stage's time.  It is an algorithmic speed-up over a literal sweep, not a bandwidth:
            # the rows live in LDS as split-bf16 operands and only one or two per tile-channel-step are touched in double.
            "equivalent_sweep_GBps": round(64 * 8 * swept_total / dev_elapsed / 1e9, 1),
            "note": "dominant kernel = mp_pursuit_kernel (one launch per step: kernel_avg_ms is its average duration in the timed region, "
                    "HIP events on the launch stream; in the N = 1 pipeline it runs on 7/8 of the CUs, the stream assembly and entropy "
                    "kernels of the neighbouring frames on the rest).  achieved = MFMA flops the kernel itself counted (every v_mfma_f32_16x16x32_bf16 executed, 16384 flop each) "
                    "/ that duration, against the dense bf16 peak.  fabric_frac_raw = traffic / duration / 8 TB/s; fabric_frac_2x applies "
                    "the guide's gfx950 correction (2 x FETCH_SIZE, valid for 16 B/lane streams) and is an upper bound.  Neither roof "
                    "binds: per-step latency does, and with every CU at work the device's power limit (stamps build: shader clock "
                    "2.37 GHz on 64 workgroups, about 1.8 GHz on 256, profiles/r03_stamps.txt -- against the MFMA peak at that clock "
                    "frac would be about 0.31; DESIGN.md 3, 9)"}
        line = {
            "metric": "encode Mpixels/s at quality=3.5" if q == 3.5 else f"encode Mpixels/s at quality={q}",
            "value": round(value, 3),
            "unit": "Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.fast else "f64",
            "data": "synthetic",
            "config": {"workload": (f"{frames} x {W}x{H} synthetic RGB per step" if world > 1 else f"{W}x{H} synthetic RGB, {frames} distinct frames in turn")
                                   + f" (mt19937 seed 12345+f), quality {q}, K={K} atoms/tile-channel, 8x8 tiles"
                                   + (f", row-striped over {world} GPUs" if world > 1 else ""),
                       "stage": "frames resident in HBM -> tile encode, stream assembly and the per-symbol work of the entropy stage "
                                "(run lengths, histograms, code writing) on the device; the host builds one code table per stream -> "
                                "container bytes in host memory "
                                + ("(the ...Fast / float flavour: PARITY UNPINNED against the reference's Eigen results; bit-identical to "
                                   "oracle/mpo_fast.c, PSNR / size equivalent to the double path)" if args.fast else
                                   "(byte-identical to the reference's encodeImage)")
                                + ("; stripes' records exchanged between ranks so that rank f produces frame f's container" if world > 1 else
                                   "; the steps of the timed region are pipelined (table building of step i beside the device work of step i+1)"),
                       "container_bytes": container_bytes, "bpp": round(8.0 * container_bytes / (W * H), 4)},
            "bytes_match_golden": bytes_match,
            "golden_containers_checked": checked,
            "hbm_resident_Mpix_s": round(value, 3),
            "host_to_bytes_Mpix_s": round(host_to_bytes, 3) if host_to_bytes else None,
            "single_frame_ms": round(single_ms, 4) if single_ms else None,
            "device_stage_Mpix_s": round(pixels_per_step / world / dev_elapsed / 1e6, 3) if world > 1 else round(pixels_per_step / dev_elapsed / 1e6, 3),
            "device_stage_ms": round(dev_elapsed * 1e3, 4),
            "roofline": roof,
        }
        if not args.no_cpu and world == 1:
            line["cpu_baseline"] = cpu_baseline(W, H, K, q, host_frames[0], args.cpu_seconds)      # the double oracle, either way
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
