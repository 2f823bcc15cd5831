"""mpc_encode_images_multi: several device lanes driven from ONE process (the C++ side of SURVEY 8e).  A one-GPU box offers one
device, so every lane names device 0 (one context per lane): stripes, peer copies, interleave and the pipelined container jobs are
the real code, only the copies stay on one device.  Every container must equal the oracle's whole-frame encodeImage."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the gpu-marked tests need a real MI355X")
    return torch


@pytest.mark.parametrize("lanes,frames,size,K", [(2, 3, (200, 136), 32), (3, 4, (97, 83), 8), (2, 1, (70, 50), 32), (4, 9, (64, 40), 16)])
def test_lanes_on_one_device_reproduce_the_oracles_bytes(gpu, oracle, lanes, frames, size, K):
    import imageexperiments_amd as ia
    W, H = size
    ctxs = [ia.create_compression_context(K, 8, 3.5, device=0) for _ in range(lanes)]
    rgbs = [oracle.synth_frame(W, H, 4000 + f) for f in range(frames)]
    octx = oracle.OracleContext(K, 8, 3.5)
    for rep in range(2):                                       # twice: a second call allocates and pipelines afresh
        blobs = ia.api.encode_images_multi(ctxs, rgbs)
        assert len(blobs) == frames
        for f in range(frames):
            assert bytes(blobs[f]) == bytes(octx.encode_image(rgbs[f])), (rep, f)
    # the quantiser override of a call reaches every lane
    q = np.ones((3, K))
    blobs = ia.api.encode_images_multi(ctxs, rgbs[:2], quant=q)
    for f in range(min(2, frames)):
        assert bytes(blobs[f]) == bytes(octx.encode_image(rgbs[f], quant=q))


def test_more_lanes_than_tile_rows_is_refused(gpu, oracle):
    import imageexperiments_amd as ia
    ctxs = [ia.create_compression_context(8, 8, 3.5, device=0) for _ in range(3)]
    with pytest.raises(ia.MpcError):
        ia.api.encode_images_multi(ctxs, [oracle.synth_frame(40, 16, 1)])      # two tile rows, three lanes
    with pytest.raises(ia.MpcError):
        ia.api.encode_images_multi([ctxs[0], ctxs[0]], [oracle.synth_frame(40, 40, 1)])   # one context in two lanes


def test_configs3_frames_at_full_size_through_two_lanes(gpu):
    """BASELINE configs[3]'s first frames (4928x3264, K = 32, quality 3.5, seeds 12345 + f) striped over two lanes: golden sha256."""
    import imageexperiments_amd as ia
    import bench
    with open(os.path.join(ROOT, "tests", "golden", "frames.json")) as f:
        gold = json.load(f)
    ctxs = [ia.create_compression_context(32, 8, 3.5, device=0) for _ in range(2)]
    rgbs = [bench.synth_frame(4928, 3264, 12345 + f) for f in range(3)]
    blobs = ia.api.encode_images_multi(ctxs, rgbs, views=True)
    for f, name in enumerate(("raise_k32_q3.5", "batch_frame1_k32_q3.5", "batch_frame2_k32_q3.5")):
        assert len(blobs[f]) == gold[name]["container_bytes"]
        assert hashlib.sha256(np.ascontiguousarray(blobs[f]).tobytes()).hexdigest() == gold[name]["container_sha256"], name
