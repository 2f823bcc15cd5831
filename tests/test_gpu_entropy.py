"""Device-side entropy stage (mp_entropy.hip) against the oracle's writeCompressed (run with -m gpu).

The whole-frame golden hashes (test_gpu_golden_frames.py) already go through it; here it is driven on its own with streams
chosen for its branches (tests/stream_cases.py), and the host route (MPC_HOST_ENTROPY=1) is checked to give the same bytes."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ia():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the gpu-marked tests need a real MI355X (there is no CPU fallback)")
    import imageexperiments_amd as ia
    return ia


def test_adversarial_streams_coded_on_the_device_equal_the_oracle(ia, oracle):
    import stream_cases
    c = stream_cases.make()
    q = stream_cases.quant(c["K"])
    want = oracle.write_compressed(dict(W=c["W"], H=c["H"], K=c["K"], bs=c["bs"], quant=q, lengths=c["counts"], codes=c["as_held"]))
    ctx = ia.create_compression_context(c["K"], 8, 3.5, device=0)
    for attempt in range(2):                       # twice: the second call finds the tables the first one left behind
        got, route = ctx.code_symbol_streams_device(c["W"], c["H"], c["counts"], c["as_coded"], quant=q)
        assert route == 0, "the device route was not taken"
        assert len(got) == len(want)
        assert got == want, f"first differing byte {next(i for i, (a, b) in enumerate(zip(got, want)) if a != b)} (call {attempt})"
    ctx.close()


def test_every_stream_empty(ia, oracle):
    K, W, H = 2, 64, 32
    counts = np.zeros(3 * 8 * 4, np.uint16)
    streams = [np.zeros(0, np.uint16)] * (6 * K)
    q = np.full((3, K), 8.0)
    want = oracle.write_compressed(dict(W=W, H=H, K=K, bs=8, quant=q, lengths=counts, codes=streams))
    ctx = ia.create_compression_context(K, 8, 3.5, device=0)
    got, route = ctx.code_symbol_streams_device(W, H, counts, streams, quant=q)
    assert route == 0 and got == want
    ctx.close()


_CHILD = r"""
import sys, hashlib
sys.path.insert(0, {root!r})
import imageexperiments_amd as ia
from bench import synth_frame
ctx = ia.create_compression_context(8, 8, 3.5, device=0)
frames = [synth_frame(328, 200, 12345 + f) for f in range(9)]
blobs = ctx.encode_images(frames)
print(" ".join(hashlib.sha256(b).hexdigest() for b in blobs))
"""


def test_host_route_gives_the_same_bytes(ia, oracle):
    """MPC_HOST_ENTROPY=1 (the route taken when a stream is outside what the device tables hold) against the device route and
    the oracle, nine pipelined frames (more than the pipeline has slots)."""
    import hashlib
    from bench import synth_frame
    octx = oracle.OracleContext(8, 8, 3.5)
    want = [hashlib.sha256(octx.encode_image(synth_frame(328, 200, 12345 + f))).hexdigest() for f in range(9)]
    # forced from the start; taken behind a completed phase 1 (more distinct symbols than the triple list may hold); device
    # ... and the frame pipeline's other schedules (one side stream per slot, no stream priorities, the pursuits on every CU, deeper lags)
    for env in ({"MPC_HOST_ENTROPY": "1"}, {"MPC_ENTROPY_TRIPLES": "50"}, {"MPC_HOST_ENTROPY": "0"}, {"MPC_SHARED_SIDE_STREAMS": "0"},
                {"MPC_SIDE_PRIORITY": "0", "MPC_SEQ_WORKGROUPS": "0"}, {"MPC_LAG_ASSEMBLY": "3", "MPC_LAG_PHASE2": "4"}):
        r = subprocess.run([sys.executable, "-c", _CHILD.format(root=ROOT)], capture_output=True, text=True, timeout=600,
                           env={**os.environ, **env})
        assert r.returncode == 0, r.stderr
        assert r.stdout.split() == want, env
