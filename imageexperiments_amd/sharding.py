"""Row-stripe sharding of frames over the GPUs of one node (one process per GPU, torch.distributed).

The tile encode is embarrassingly parallel: rank r encodes tile rows [begin_r, end_r) of EVERY frame of a step
(`stripe_bounds`, one launch).  A container, however, is one bit stream per frame in the reference's x-outer / y-inner tile
order: DC difference chains and run-length coding run across stripe boundaries inside every tile column
(CompressedImage.cpp:428-453, 535-537), so a frame's records must meet in one place before they are coded.
With N frames per step (weak scaling) frame f is OWNED by rank f: every rank sends its stripe of frame f to rank f
(`exchange_stripes`: batched point-to-point, RCCL over xGMI on GPUs, gloo in the CPU tests; exact sizes, u16 counts and
u32 records as they are), the owner interleaves the N stripes into frame order and runs stream assembly + entropy stage.
That is the path's only exchange, and every rank does an equal share of the host work.

The symbol-histogram all-reduce the survey sketched is not on this path: a histogram cannot carry what decides the
bytes (first-occurrence order of the symbols for the Huffman ties, run lengths, DC chains), and the frame's owner gets the
histogram of its own frame for free while it codes it.  `histogram_of_records` / `allreduce_histogram` remain for callers
that want global statistics (mpc_histogram_device is their device form).
"""
import numpy as np

HIST_BINS = 8192


def stripe_bounds(tiles_y, world, rank):
    """Contiguous tile-row stripes, remainder to the first ranks (540 rows over 8 -> 68 x4 + 67 x4)."""
    base, rem = divmod(tiles_y, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def interleave_stripes(parts, tiles_x, tiles_y, world):
    """parts[r] = array [tiles_x * rows_r, ...] in the C ABI's stripe order (t = tx*rows_r + ty_local) ->
    whole-frame array [tiles_x * tiles_y, ...] in the reference's order (t = tx*tiles_y + ty).  numpy arrays or torch tensors."""
    tail = tuple(parts[0].shape[1:])
    if isinstance(parts[0], np.ndarray):
        out = np.empty((tiles_x, tiles_y) + tail, parts[0].dtype)
    else:
        import torch
        out = torch.empty((tiles_x, tiles_y) + tail, dtype=parts[0].dtype, device=parts[0].device)
    for r in range(world):
        b, e = stripe_bounds(tiles_y, world, r)
        out[:, b:e] = parts[r].reshape((tiles_x, e - b) + tail)
    return out.reshape((tiles_x * tiles_y,) + tail)


def nccl_dtypes():
    """What torch's NCCL/RCCL process group carries (torch/csrc/distributed/c10d/NCCLUtils.hpp: no 16-bit integers)."""
    import torch
    return {torch.uint8, torch.int8, torch.int32, torch.int64, torch.float16, torch.float32, torch.float64, torch.bfloat16, torch.bool}


def exchange_stripes(dist, sets, tiles_x, tiles_y, via_cpu=False, recv=None):
    """ONE batched point-to-point exchange for all record arrays of a step.
    sets[k][f] = this rank's stripe of frame f of array k (torch tensor [tiles_x * rows_mine, ...]), f = 0 .. world-1.
    Returns parts[k][r] = rank r's stripe of the frame THIS rank owns (frame index = rank), exact sizes; parts[k][rank] is
    sets[k][rank] itself (no copy).  Every tensor must have a dtype RCCL carries (`nccl_dtypes`): 16-bit counts travel as bytes.
    via_cpu: stage through host memory (gloo has no device point-to-point).  recv[k][r]: preallocated receive buffers."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    ok = nccl_dtypes()
    parts, ops = [], []
    for k, mine in enumerate(sets):
        if mine[0].dtype not in ok:
            raise TypeError(f"{mine[0].dtype} cannot travel over RCCL: view it as uint8")
        tail = tuple(mine[0].shape[1:])
        dev = mine[0].device
        got = []
        for r in range(world):
            if r == rank:
                got.append(mine[rank])
                continue
            b, e = stripe_bounds(tiles_y, world, r)
            shape = (tiles_x * (e - b),) + tail
            if recv is not None and not via_cpu:
                buf = recv[k][r]
                assert tuple(buf.shape) == shape and buf.dtype == mine[0].dtype
            else:
                buf = torch.empty(shape, dtype=mine[0].dtype, device="cpu" if via_cpu else dev)
            got.append(buf)
            ops.append(dist.P2POp(dist.isend, (mine[r].cpu() if via_cpu else mine[r]).contiguous(), r))
            ops.append(dist.P2POp(dist.irecv, buf, r))
        parts.append(got)
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if via_cpu:
        parts = [[p if r == rank else p.to(sets[k][0].device) for r, p in enumerate(got)] for k, got in enumerate(parts)]
    return parts


class StripedEncoder:
    """The N > 1 path on a GPU: stripe encode of every frame (one launch), stripe exchange (one batched send / receive over
    RCCL), the stripes copied to their places in the frame this rank owns (mpc_interleave_stripe_device), that frame -> container
    bytes (stream assembly and the per-symbol entropy work on the device, code tables on the host).  `step` does one step
    synchronously on `stream`; `run` pipelines consecutive steps over `stream` (the tile encodes, back to back) and a side stream
    (everything behind a tile encode).  A stream is made torch's current stream while the exchange is enqueued on it (the process
    group orders its transfers against the current stream)."""

    SLOTS = 3            # whole-frame records / container jobs in flight
    STRIPE_SETS = 2      # stripe records: the exchange of step i reads one set while the tile encode of step i + 1 fills the other

    def __init__(self, ctx, width, height, frames, world, rank, backend):
        import torch
        import torch.distributed as dist
        if frames != world:
            raise ValueError("weak scaling: one frame per rank and step")
        self.ctx, self.W, self.H, self.world, self.rank, self.dist = ctx, width, height, world, rank, dist
        self.via_cpu = backend != "nccl"
        self.tiles_x, self.tiles_y = (width + 7) // 8, (height + 7) // 8
        self.begin, self.end = stripe_bounds(self.tiles_y, world, rank)
        per_frame = self.tiles_x * (self.end - self.begin)
        K = ctx.K
        self.counts = [torch.zeros((frames, per_frame, 3), dtype=torch.int16, device="cuda") for _ in range(self.STRIPE_SETS)]
        self.choices = [torch.zeros((frames, per_frame, 3, K), dtype=torch.int32, device="cuda") for _ in range(self.STRIPE_SETS)]
        tiles = self.tiles_x * self.tiles_y
        # whole-frame records of the frame this rank owns: a container job reads them until its `collect`
        self.frame_counts = [torch.zeros((tiles, 3), dtype=torch.int16, device="cuda") for _ in range(self.SLOTS)]
        self.frame_choices = [torch.zeros((tiles, 3, K), dtype=torch.int32, device="cuda") for _ in range(self.SLOTS)]
        self.recv = None
        if not self.via_cpu:
            rows = [stripe_bounds(self.tiles_y, world, r) for r in range(world)]
            self.recv = [[torch.empty((self.tiles_x * (e - b), 6), dtype=torch.uint8, device="cuda") for b, e in rows],
                         [torch.empty((self.tiles_x * (e - b), 3, K), dtype=torch.int32, device="cuda") for b, e in rows]]
        self.side = None                                        # `run`'s second stream, made on first use
        ctx.reserve(frames * per_frame)

    def _encode_stripes(self, d_rgb, stream, stripe_set):
        """stripe encode of every frame of the step, one launch"""
        W, H = self.W, self.H
        self.ctx.encode_batch_device(d_rgb.data_ptr(), self.world, W * H * 3, W, H, W * 3, self.begin, self.end,
                                     self.counts[stripe_set].data_ptr(), self.choices[stripe_set].data_ptr(), stream=stream.cuda_stream)

    def _gather_my_frame(self, stream, stripe_set, slot):
        """stripe exchange, then the stripes copied to their places in slot `slot`'s whole-frame records; on torch's current
        stream, which must be `stream`"""
        W, H = self.W, self.H
        cparts, hparts = exchange_stripes(self.dist, [list(self.counts[stripe_set].view(torch_uint8())), list(self.choices[stripe_set])],
                                          self.tiles_x, self.tiles_y, self.via_cpu, self.recv)
        fc, fh = self.frame_counts[slot], self.frame_choices[slot]
        for r in range(self.world):
            b, e = stripe_bounds(self.tiles_y, self.world, r)
            self.ctx.interleave_stripe_device(cparts[r].data_ptr(), hparts[r].data_ptr(), W, H, b, e, fc.data_ptr(), fh.data_ptr(),
                                              stream=stream.cuda_stream)
        self._held = (cparts, hparts)                           # staged copies of the gloo rehearsal stay alive until the next step
        return fc, fh

    def step(self, d_rgb, stream):
        import torch
        with torch.cuda.stream(stream):
            self._encode_stripes(d_rgb, stream, 0)
            counts, choices = self._gather_my_frame(stream, 0, 0)
            return self.ctx.records_to_container_device(counts.data_ptr(), choices.data_ptr(), self.W, self.H, stream=stream.cuda_stream)

    def run(self, d_rgb, stream, steps, views=False):
        """`steps` steps, software-pipelined over two streams.  `stream`: the tile encodes, one behind the other, on 7/8 of the
        CUs (mpc_context_set_tile_encode_workgroups) so that what the side stream carries runs beside them instead of between
        them.  Side stream: behind tile encode i (an event) the stripe exchange, the interleave, stream assembly + entropy
        phase 1 of this rank's frame; behind those, once the host has built step i - 1's code tables (while the device works on
        step i), its phase 2 + container copy.  Tile encode i + 2 waits (an event) until exchange and interleave i have read
        the stripe set it will overwrite.  Step i - 2's container is collected at step i.  Same containers as `step`, in order."""
        import torch
        slots = self.SLOTS
        if self.side is None:
            self.side = torch.cuda.Stream(device=stream.device)
        side = self.side
        cus = self.ctx.max_waves // 12
        self.ctx.set_tile_encode_workgroups(cus - cus // 8 if cus >= 16 else 0)
        out = []
        read = [None] * self.STRIPE_SETS                        # event: the side stream is through with this stripe set
        encoded = {}

        def enqueue_encode(k):
            ss = k % self.STRIPE_SETS
            if read[ss] is not None:
                stream.wait_event(read[ss])
            self._encode_stripes(d_rgb, stream, ss)
            encoded[k] = torch.cuda.Event()
            encoded[k].record(stream)

        try:
            side.wait_stream(stream)                            # whatever the caller enqueued before this call comes first
            if steps >= 1:
                enqueue_encode(0)
            for i in range(steps):
                if i + 1 < steps:
                    enqueue_encode(i + 1)                       # the next tile encode is queued before the host waits for anything
                side.wait_event(encoded.pop(i))
                with torch.cuda.stream(side):
                    counts, choices = self._gather_my_frame(side, i % self.STRIPE_SETS, i % slots)   # slot i % 3 was collected at step i - 1
                    read[i % self.STRIPE_SETS] = torch.cuda.Event()
                    read[i % self.STRIPE_SETS].record(side)
                    self.ctx.container_job_begin(i % slots, counts.data_ptr(), choices.data_ptr(), self.W, self.H, stream=side.cuda_stream)
                if i >= 1:
                    self.ctx.container_job_tables((i - 1) % slots)
                if i >= 2:
                    out.append(self.ctx.container_job_collect((i - 2) % slots, views))
            if steps >= 1:
                self.ctx.container_job_tables((steps - 1) % slots)
            if steps >= 2:
                out.append(self.ctx.container_job_collect((steps - 2) % slots, views))
            if steps >= 1:
                out.append(self.ctx.container_job_collect((steps - 1) % slots, views))
            stream.wait_stream(side)                            # the caller's stream is behind everything this call enqueued
        except BaseException:
            for sl in range(slots):                             # no job may outlive the records it reads
                try:
                    self.ctx.container_job_cancel(sl)
                except Exception:
                    pass
            raise
        finally:
            self.ctx.set_tile_encode_workgroups(0)
        return out


def torch_uint8():
    import torch
    return torch.uint8


def histogram_of_records(counts, choices, K):
    """numpy form of the device histogram kernel: [(1 + 6K), 8192] uint32; stream 0 = lengths,
    1 + 2K*ch + 2i = deltaId at step i, +1 = intCoeff at step i.  choices: uint32 [T,3,K].  Symbols >= 8192 (possible only
    with custom quantisers far below the data) are not counted, here and on the device."""
    hist = np.zeros((1 + 6 * K, HIST_BINS), np.int64)
    hist[0] = np.bincount(counts.reshape(-1), minlength=HIST_BINS)[:HIST_BINS]
    for ch in range(3):
        for i in range(K):
            m = counts[:, ch] > i
            rec = choices[m, ch, i]
            hist[1 + 2 * K * ch + 2 * i] = np.bincount(rec & 0xFFFF, minlength=HIST_BINS)[:HIST_BINS]
            hist[2 + 2 * K * ch + 2 * i] = np.bincount(rec >> 16, minlength=HIST_BINS)[:HIST_BINS]
    return hist


def allreduce_histogram(dist, hist, device="cpu"):
    """Sum of the ranks' histograms on every rank."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(hist, np.int64)).to(device)
    dist.all_reduce(t)
    return t.cpu().numpy()
