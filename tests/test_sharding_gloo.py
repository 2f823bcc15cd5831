"""N > 1 path on CPU: world_size-2 gloo processes.  Each rank produces the records of its row stripe (with the
ORACLE standing in for the device encoder -- there is no GPU here; the -m gpu tests prove device == oracle),
the histograms are all-reduced, the records gathered and re-interleaved, and rank 0 builds the container with
the PRODUCT's host entropy stage.  The bytes must equal the oracle's whole-frame encodeImage."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, H, K, q, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from oracle import oracle_py as O
    import imageexperiments_amd as ia
    from imageexperiments_amd import sharding as sh
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rgb = O.synth_frame(W, H, 12345)
    tiles_x, tiles_y = (W + 7) // 8, (H + 7) // 8
    b, e = sh.stripe_bounds(tiles_y, world, rank)
    octx = O.OracleContext(K, 8, q)
    # stripe records in the C ABI's stripe order: t = tx*rows + ty_local
    oc, od, ok, _, _ = octx.encode_tiles(rgb)
    full_c = oc.reshape(tiles_x, tiles_y, 3)
    full_ch = (od.astype(np.uint32) | (ok.astype(np.uint32) << 16)).reshape(tiles_x, tiles_y, 3, K)
    idx = np.arange(K)[None, None, None, :] < full_c[..., None]
    full_ch = np.where(idx, full_ch, 0)
    counts = full_c[:, b:e].reshape(-1, 3)
    choices = full_ch[:, b:e].reshape(-1, 3, K)
    hist = sh.allreduce_histogram(dist, sh.histogram_of_records(counts, choices, K))
    gc, gch = sh.gather_records(dist, counts, choices, tiles_x, tiles_y, K)
    if rank == 0:
        whole = sh.histogram_of_records(full_c.reshape(-1, 3), full_ch.reshape(-1, 3, K), K)
        assert (hist == whole).all(), "all-reduced histogram != whole-frame histogram"
        assert (gc == full_c.reshape(-1, 3)).all()
        blob = ia.assemble_streams(W, H, K, 8, octx.quant, gc, gch)
        ref = octx.encode_image(rgb)
        with open(out_path, "w") as f:
            f.write("ok" if blob == ref else f"bytes differ: {len(blob)} vs {len(ref)}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("size", [(96, 72), (70, 50)])
def test_two_rank_row_stripes_reproduce_whole_frame_bytes(tmp_path, oracle, size):
    import torch.multiprocessing as mp
    W, H = size
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, _free_port(), W, H, 8, 3.5, out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def test_stripe_bounds_cover_and_match_survey():
    from imageexperiments_amd.sharding import stripe_bounds, interleave_stripes
    assert [stripe_bounds(540, 8, r)[1] - stripe_bounds(540, 8, r)[0] for r in range(8)] == [68] * 4 + [67] * 4
    assert [stripe_bounds(408, 8, r)[1] - stripe_bounds(408, 8, r)[0] for r in range(8)] == [51] * 8
    for ty, w in ((135, 8), (9, 2), (5, 8), (1, 1)):
        edges = [stripe_bounds(ty, w, r) for r in range(w)]
        assert edges[0][0] == 0 and edges[-1][1] == ty
        assert all(edges[i][1] == edges[i + 1][0] for i in range(w - 1))
    tx, ty, w = 3, 5, 2
    full = np.arange(tx * ty).reshape(tx, ty)
    parts = [full[:, slice(*stripe_bounds(ty, w, r))].reshape(-1) for r in range(w)]
    assert (interleave_stripes(parts, tx, ty, w) == full.reshape(-1)).all()
