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
    # the first node run must explain itself: where a data-parallel step's time goes, per rank (round 5)
    d = j["dp_breakdown"]
    parts = d["chain_ms"] + d["exchange_ms"] + d["adam_ms"] + d["other_ms"]
    assert abs(parts - d["traced_ms_per_step"]) <= 0.05 * d["traced_ms_per_step"]
    assert d["exchange"]["form"] in ("one-bucket", "two-part") and d["exchange"]["bucket_bytes"] == 4 * 2 * (384 * 384 * 2 + 2 * 384)
    assert d["exchange"]["backend"] == "gloo" and len(d["per_rank"]) == 2 and d["k1_placement"].startswith("under the exchange")
    assert j["calibration"]["exchange_form"] == d["exchange"]["form"] and "data-parallel" in j["calibration"]["regime"]


def _bench_module():
    import importlib.util

    spec = importlib.util.spec_from_file_location("tt_bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_dp_breakdown_components_sum_to_the_traced_step():
    """bench.breakdown(): chain + exchange + adam + other = the traced step, from the trainer's per-step event quadruples
    (fake events with known times here; the first traced step is dropped, an incomplete one ignored)."""
    class Ev:
        def __init__(self, t):
            self.t = t

        def elapsed_time(self, other):
            return other.t - self.t

    trace, t = [], 0.0
    for i in range(6):
        row = [Ev(t), Ev(t + 0.110), Ev(t + 0.110 + 0.060), Ev(t + 0.110 + 0.060 + 0.007)]
        trace.append(row)
        t += 0.190
    trace.append([Ev(t), None, None, None])  # a step whose events were never completed
    d = _bench_module().breakdown(trace)
    assert d["steps"] == 4 and set(d) >= {"chain_ms", "exchange_ms", "adam_ms", "other_ms", "traced_ms_per_step"}
    assert abs(d["chain_ms"] - 0.110) < 1e-9 and abs(d["exchange_ms"] - 0.060) < 1e-9 and abs(d["adam_ms"] - 0.007) < 1e-9
    assert abs(d["chain_ms"] + d["exchange_ms"] + d["adam_ms"] + d["other_ms"] - d["traced_ms_per_step"]) < 1e-9
    assert abs(d["traced_ms_per_step"] - 0.190) < 1e-9
    assert _bench_module().breakdown(trace[:2]) is None


def test_watchdog_ends_a_hung_rank_with_its_state_and_a_nonzero_code():
    """A rank that never gets through rendezvous / its first collective must not hang the node run: the watchdog prints
    the rank's state and exits 3 (a fresh child process here; nothing is exec'ed)."""
    code = ("import sys, time; sys.argv=['bench.py']; sys.path.insert(0, %r)\n"
            "import importlib.util\n"
            "spec = importlib.util.spec_from_file_location('b', %r); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n"
            "with b.watchdog(0.5, 'first collective', 1):\n"
            "    time.sleep(30)\n") % (ROOT, os.path.join(ROOT, "bench.py"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stderr[-500:])
    assert "rank 1" in r.stderr and "first collective" in r.stderr and "MASTER_ADDR" in r.stderr
    ok = subprocess.run([sys.executable, "-c", code.replace("time.sleep(30)", "pass")], capture_output=True, text=True, timeout=120)
    assert ok.returncode == 0
