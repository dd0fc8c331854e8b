"""bench.py --gpus N (N > 1) from a plain shell starts itself under torch.distributed.run as a child process."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "6",
                           "--warmup", "2", "--no-cpu-baseline", *extra], capture_output=True, text=True, timeout=timeout,
                          env=env, cwd=ROOT)


@pytest.mark.skipif(torch.cuda.is_available(), reason="the GPU variant below covers a box with a GPU")
def test_self_launch_spawns_one_rank_per_gpu_and_forwards_the_exit_code():
    r = _run()
    # no GPU here: a rank stops with the product's "needs an MI355X" message (the elastic agent tears the other one down
    # as soon as the first has failed, so it may not get to print its own), and the parent returns that failure
    assert r.returncode != 0
    assert r.stderr.count("bench.py needs an MI355X") >= 1, r.stderr[-2000:]
    assert "torch.distributed" in r.stderr or "ChildFailedError" in r.stderr  # it did go through torch.distributed.run


@pytest.mark.gpu
def test_self_launch_two_ranks_print_one_json_line():
    r = _run()
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 2 * j["config"]["batch_per_gpu"]
    assert j["config"]["backend"] == "gloo" and j["config"]["rccl_ranks"] == 0 and j["value"] > 0
    assert j["effective_warmup_steps"] >= j["warmup"]
