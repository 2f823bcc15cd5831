"""bench.py's driver contract around the N > 1 path.

CPU: `python bench.py --gpus 2` without a launcher starts two ranks itself; with no GPU both ranks refuse to run (the hot path has
no CPU fallback) and the parent must relay the failure instead of hanging or printing a line.
GPU (-m gpu): BASELINE configs[3]'s frames (4928x3264, K = 32, quality 3.5, seeds 12345 + f) through the striped path at full size:
two ranks (gloo, both on device 0: a one-GPU box cannot run two RCCL ranks) stripe both frames, exchange the stripes, and each
produces its frame's container -- whose sha256 must equal the oracle's whole-frame golden (tests/golden/frames.json:
raise_k32_q3.5 and batch_frame1_k32_q3.5)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def test_self_launch_relays_a_failing_rank():
    env = dict(os.environ)
    env.pop("RANK", None)
    env["HIP_VISIBLE_DEVICES"] = ""                           # no GPU for the children wherever this runs
    env["CUDA_VISIBLE_DEVICES"] = ""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu",
                        "--backend", "gloo", "--share-device"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    assert "needs a GPU" in p.stderr
    assert not p.stdout.strip().startswith("{")


@pytest.mark.gpu
def test_striped_path_at_full_size_reproduces_the_golden_containers():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the gpu-marked tests need a real MI355X")
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-device", "--steps", "3",
                        "--warmup", "1", "--no-cpu"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 3
    assert line["golden_containers_checked"] == 2             # rank 0: seed 12345, rank 1: seed 12346
    assert line["bytes_match_golden"] is True
    assert line["config"]["container_bytes"] == 6631961       # rank 0's frame
