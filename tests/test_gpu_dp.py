"""GPU: data-parallel TripletTrainer with the real HIP engine — two ranks (sharing the one
GPU of the test box, gloo transport) on shards of the global batches must reproduce the
single-rank trainer on the global batches (and therefore the reference's train_epoch)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import two_towers_oracle as orc  # noqa: E402

pytestmark = pytest.mark.gpu
V, D, P, B, S, NB = 5000, 384, 64, 130, 48, 3


def _setup():
    table = orc.synthetic_table(V, D, seed=3) * 30
    params = orc.init_params(D, P, seed=4)
    batches = [orc.synthetic_batch(B, S, V, seed=70 + i, ragged=True) for i in range(NB)]
    gb = [(torch.stack([r[0] for r in b]).int(), None, torch.stack([r[1].sum(1) for r in b]).int()) for b in batches]
    return table, params, batches, gb


def _worker(rank, world, port, q):
    try:
        from two_towers_overlords_amd.training import TripletTrainer, shard_batch

        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        table, params, _, gb = _setup()
        tr = TripletTrainer(table=table.cuda(), params={k: v.cuda() for k, v in params.items()}, max_batch=B,
                            max_seq_len=S, margin=0.3, lr=1e-3)
        shards, sizes = [], []
        for b in gb:
            sh, n = shard_batch(tuple(None if t is None else t.cuda() for t in b), rank, world)
            shards.append(sh)
            sizes.append(n)
        avg = tr.train_epoch(shards, global_batch_sizes=sizes)
        if rank == 0:
            q.put(("ok", avg, {k: v.cpu().numpy().copy() for k, v in tr.named_parameters().items()}))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback

        q.put(("error", f"rank {rank}: {e}\n{traceback.format_exc()}", None))
        raise


def test_two_ranks_equal_one_rank_and_oracle():
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = 29600 + os.getpid() % 1000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get()
    for p in procs:
        p.join(180)
    assert out[0] == "ok", out[1]
    assert all(p.exitcode == 0 for p in procs)
    _, avg2, params2 = out

    table, params, batches, _ = _setup()
    st = orc.AdamState(lr=1e-3)
    ref_avg, _ = orc.train_epoch(table, batches, params, st, 0.3)
    assert abs(avg2 - ref_avg) <= 1e-5
    from conftest import assert_params_close

    for k in orc.PARAM_KEYS:
        assert_params_close(params2[k], params[k].numpy(), 1e-3, NB)
