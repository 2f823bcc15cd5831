"""Whole-frame byte identity at every BASELINE.json size and quality (run with -m gpu).

tests/golden/frames.json holds, for every BASELINE frame, what the ORACLE produced in the build container
(tools/gen_golden.py: sha256 and size of the container, sha256 of the live records, sum of counts per channel,
length of every stream).  Here the PRODUCT (libmpcodec.so through the C ABI: device tile encode + host entropy
stage) encodes the same pixels and must reproduce all of it -- whole frames, not sampled tile columns.
Inputs are regenerated on the box (synthetic generator; the reference's own .mn decoded by the product's device
decoder; the reference's .jpg decoded by PIL with the decoded pixels' hash pinned)."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

with open(os.path.join(GOLDEN, "frames.json")) as _f:
    FRAMES = json.load(_f)


@pytest.fixture(scope="module")
def ia(request):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the gpu-marked tests need a real MI355X (there is no CPU fallback)")
    import imageexperiments_amd as ia
    return ia


_contexts = {}


def _context(ia, K, q):
    key = (K, q)
    if key not in _contexts:
        _contexts[key] = ia.create_compression_context(K, 8, q, device=0)
    return _contexts[key]


def _pixels(ia, spec):
    if spec["kind"] == "synthetic":
        from bench import synth_frame
        return synth_frame(spec["width"], spec["height"], spec["seed"])
    if spec["kind"] == "mn":
        with open(os.path.join(GOLDEN, "r0c1de5e1t_3_5.mn"), "rb") as f:
            return ia.api.decode_image(f.read(), _context(ia, 32, 3.5))
    if spec["kind"] == "jpg":
        from PIL import Image
        return np.ascontiguousarray(np.asarray(Image.open(os.path.join(GOLDEN, "r0c1de5e1t.jpg")).convert("RGB")))
    raise ValueError(spec["kind"])


def _records_sha(counts, choices, K):
    rec = choices.view(np.uint32).reshape(counts.shape[0], 3, K)
    step = np.arange(K)[None, None, :]
    rec = np.where(step < counts[:, :, None], rec, 0).astype(np.uint32)
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(counts, np.uint16).tobytes())
    h.update(np.ascontiguousarray(rec).tobytes())
    return h.hexdigest()


@pytest.mark.parametrize("name", sorted(FRAMES))
def test_whole_frame_bytes_equal_the_oracles(ia, name):
    spec = FRAMES[name]
    K, q = spec["K"], spec["quality"]
    rgb = _pixels(ia, spec)
    assert rgb.shape == (spec["height"], spec["width"], 3)
    if hashlib.sha256(rgb.tobytes()).hexdigest() != spec["rgb_sha256"]:
        if spec["kind"] == "jpg":
            pytest.skip("this box's libjpeg decodes Data/r0c1de5e1t.jpg to other pixels than the build container's")
        pytest.fail("input pixels differ from the ones the golden hashes were made from")
    ctx = _context(ia, K, q)
    # "fast": the `...Fast` (float) flavour, golden hashes by oracle/mpo_fast.c (a definition of the float mode: parity unpinned
    # against the reference's Eigen results; what is checked is that the device reproduces the definition on whole frames)
    ctx.set_fast(spec.get("flavour") == "fast")
    # records first (device stage alone), then the container (device stage + host entropy stage)
    counts, choices, _energy, swept = ctx.encode_tiles(rgb)
    assert [int(counts[:, ch].sum()) for ch in range(3)] == spec["sum_counts"]
    assert int(swept.astype(np.int64).sum()) == spec["swept_rows"]
    assert _records_sha(counts, choices, K) == spec["records_sha256"]
    blob = ctx.encode_image(rgb)
    assert len(blob) == spec["container_bytes"]
    assert hashlib.sha256(blob).hexdigest() == spec["container_sha256"]
    streams = ia.api.read_compressed(blob)
    lengths = [len(streams["lengths"])] + [len(c) for c in streams["codes"]]
    assert lengths == spec["stream_lengths"]
    ctx.set_fast(False)
