"""Row-stripe sharding of frames over the GPUs of one node (one process per GPU, torch.distributed).

The tile encode is embarrassingly parallel: rank r encodes tile rows [begin_r, end_r) of every frame
(`stripe_bounds`).  The only exchange on the data path is the all-reduce of the per-stream symbol
histograms (RCCL over xGMI on GPUs, gloo in the CPU tests) that feeds the Huffman/Golomb tables.  The
per-tile records themselves are gathered to rank 0 (they are what the container is made of): the DC
difference chains and the run-length variant depend on the reference's x-outer / y-inner tile order
across stripes (CompressedImage.cpp:428-453, 535-537), which no histogram captures.
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
    whole-frame array [tiles_x * tiles_y, ...] in the reference's order (t = tx*tiles_y + ty)."""
    tail = parts[0].shape[1:]
    out = np.empty((tiles_x, tiles_y) + tail, parts[0].dtype)
    for r in range(world):
        b, e = stripe_bounds(tiles_y, world, r)
        out[:, b:e] = parts[r].reshape((tiles_x, e - b) + tail)
    return out.reshape((tiles_x * tiles_y,) + tail)


def histogram_of_records(counts, choices, K):
    """numpy form of the device histogram kernel: [(1 + 6K), 8192] uint32; stream 0 = lengths,
    1 + 2K*ch + 2i = deltaId at step i, +1 = intCoeff at step i.  choices: uint32 [T,3,K]."""
    hist = np.zeros((1 + 6 * K, HIST_BINS), np.int64)
    hist[0] = np.bincount(counts.reshape(-1), minlength=HIST_BINS)[:HIST_BINS]
    for ch in range(3):
        for i in range(K):
            m = counts[:, ch] > i
            rec = choices[m, ch, i]
            hist[1 + 2 * K * ch + 2 * i] = np.bincount(rec & 0xFFFF, minlength=HIST_BINS)[:HIST_BINS]
            hist[2 + 2 * K * ch + 2 * i] = np.bincount(rec >> 16, minlength=HIST_BINS)[:HIST_BINS]
    return hist


def gather_records(dist, counts, choices, tiles_x, tiles_y, K, device="cpu"):
    """All ranks call this with their stripe's records (numpy). Rank 0 gets (counts[T,3], choices[T,3,K]) of the
    whole frame in the reference's tile order; the others get (None, None).  Stripes differ in size, so the
    tensors are padded to the largest stripe for the collective."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    max_rows = max(stripe_bounds(tiles_y, world, r)[1] - stripe_bounds(tiles_y, world, r)[0] for r in range(world))
    pad = tiles_x * max_rows
    n = counts.shape[0]
    c = torch.zeros((pad, 3), dtype=torch.int32, device=device)
    h = torch.zeros((pad, 3, K), dtype=torch.int64, device=device)
    c[:n] = torch.from_numpy(counts.astype(np.int32)).to(device)
    h[:n] = torch.from_numpy(choices.astype(np.int64)).to(device)
    cs = [torch.zeros_like(c) for _ in range(world)] if rank == 0 else None
    hs = [torch.zeros_like(h) for _ in range(world)] if rank == 0 else None
    dist.gather(c, cs, dst=0)
    dist.gather(h, hs, dst=0)
    if rank != 0:
        return None, None
    cparts, hparts = [], []
    for r in range(world):
        b, e = stripe_bounds(tiles_y, world, r)
        m = tiles_x * (e - b)
        cparts.append(cs[r][:m].cpu().numpy().astype(np.uint16))
        hparts.append(hs[r][:m].cpu().numpy().astype(np.uint32))
    return (interleave_stripes(cparts, tiles_x, tiles_y, world), interleave_stripes(hparts, tiles_x, tiles_y, world))


def allreduce_histogram(dist, hist, device="cpu"):
    """Sum of the ranks' histograms on every rank (the path's only data-path collective)."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(hist, np.int64)).to(device)
    dist.all_reduce(t)
    return t.cpu().numpy()
