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


def _ranks(worker, world, args=(), salt=0):
    """Every multi-process test of this file goes through tests/mp_util.run_ranks: the result queue is read with a
    timeout and ranks still alive after it are killed, so a rank that dies or hangs in native code / a collective
    fails ITS test instead of stalling the pytest session."""
    from mp_util import free_port, run_ranks

    return run_ranks(worker, world, args, timeout=420, join=60, port=free_port(29300, 2500, salt))


def test_two_ranks_equal_one_rank_and_oracle():
    _, avg2, params2 = _ranks(_worker, 2, salt=1)

    table, params, batches, _ = _setup()
    st = orc.AdamState(lr=1e-3)
    ref_avg, _ = orc.train_epoch(table, batches, params, st, 0.3)
    assert abs(avg2 - ref_avg) <= 1e-5
    from conftest import assert_params_close

    for k in orc.PARAM_KEYS:
        assert_params_close(params2[k], params[k].numpy(), 1e-3, NB)


# --------------------------------------------------------------------------- f4 in-batch negatives
def _inbatch_setup():
    Vv, Dd, Pp, Bb, Ss, nb = 3000, 128, 48, 64, 24, 3
    table = orc.synthetic_table(Vv, Dd, seed=13) * 30
    params = orc.init_params(Dd, Pp, seed=14)
    batches = [orc.synthetic_batch(Bb, Ss, Vv, seed=170 + i, ragged=True)[:2] for i in range(nb)]
    g = torch.Generator().manual_seed(15)
    qids = [torch.randint(0, Bb // 2, (Bb,), generator=g) for _ in range(nb)]
    return table, params, batches, qids, (Vv, Dd, Pp, Bb, Ss, nb)


def _inbatch_reference(Bg):
    table, params, batches, qids, (_, _, _, Bb, _, nb) = _inbatch_setup()
    st = orc.AdamState(lr=1e-3)
    losses = []
    for (qb, pb), qid in zip(batches, qids):
        leaves = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
        q = orc.encode(table, qb[0][:Bg], qb[1][:Bg], leaves, "query_tower")
        d = orc.encode(table, pb[0][:Bg], pb[1][:Bg], leaves, "document_tower")
        loss = orc.inbatch_triplet_loss(q, d, 0.2, None, qid[:Bg], qid[:Bg])
        loss.backward()
        orc.adam_step(params, {k: v.grad for k, v in leaves.items()}, st)
        losses.append(loss.item())
    return sum(losses) / nb, params


def _inbatch_worker(rank, world, port, Bg, q):
    try:
        from two_towers_overlords_amd.training import TripletBatch, TripletTrainer, shard_batch

        if world > 1:
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
            dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        table, params, batches, qids, (_, _, _, Bb, Ss, _) = _inbatch_setup()
        tr = TripletTrainer(table=table.cuda(), params={k: v.cuda() for k, v in params.items()},
                            max_batch=(Bg + world - 1) // world, max_seq_len=Ss, margin=0.2, lr=1e-3, loss="inbatch")
        shards, sizes = [], []
        for (qb, pb), qid in zip(batches, qids):  # the global batch (first Bg rows), then this rank's contiguous shard
            gb = TripletBatch(ids=torch.stack([qb[0][:Bg], pb[0][:Bg]]).int().cuda(),
                              lens=torch.stack([qb[1][:Bg].sum(1), pb[1][:Bg].sum(1)]).int().cuda(),
                              query_ids=qid[:Bg].cuda())
            sh, n = shard_batch(gb, rank, world)
            assert sh.query_ids.shape[0] == sh.size
            shards.append(sh)
            sizes.append(n)
        avg = tr.train_epoch(shards, global_batch_sizes=sizes)
        if rank == 0:
            q.put(("ok", avg, {k: v.cpu().numpy().copy() for k, v in tr.named_parameters().items()}))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback

        q.put(("error", f"rank {rank}: {e}\\n{traceback.format_exc()}", None))
        raise


@pytest.mark.parametrize("world,Bg", [(1, 64), (2, 64), (2, 63)])
def test_inbatch_negatives_trainer_vs_oracle(world, Bg):
    """In-batch-negatives loss through the trainer: one rank, and two ranks exchanging document
    embeddings + query ids (all-gather) and the documents' gradient (all-reduce) — equal shards
    and uneven ones (63 = 32 + 31: zero-padded exchange blocks) — against the oracle's autograd."""
    _, avg, params = _ranks(_inbatch_worker, world, (Bg,), salt=2 + world + Bg)
    ref_avg, ref_params = _inbatch_reference(Bg)
    assert abs(avg - ref_avg) <= 1e-5
    from conftest import assert_params_close

    for k in orc.PARAM_KEYS:
        assert_params_close(params[k], ref_params[k].numpy(), 1e-3, 3)


# --------------------------------------------------------------------------- run_training under data parallelism
def _rt_items():
    import random

    rng = random.Random(3)
    V, items = 300, []
    for qid in range(80):
        qt = [rng.randrange(1, V) for _ in range(rng.randint(3, 6))]
        for _ in range(rng.randint(1, 3)):
            doc = [rng.choice(qt) if rng.random() < 0.6 else rng.randrange(1, V) for _ in range(rng.randint(6, 20))]
            items.append({"query_id": qid, "query": " ".join(map(str, qt)), "positive": " ".join(map(str, doc))})
    return V, items


def _rt_worker(rank, world, port, cache, q):
    try:
        from test_gpu_eval import IntTokenizer
        from two_towers_overlords_amd.model import EmbeddingOnlyEncoder
        from two_towers_overlords_amd.training import run_training

        if world > 1:
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
            dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        V, items = _rt_items()
        table = orc.synthetic_table(V, 64, seed=9) * 30
        torch.manual_seed(100 + rank)  # every rank initialises its projections differently, like unseeded nn.Linear
        m = run_training(num_epochs=2, batch_size=33, learning_rate=1e-2, max_samples=10 ** 6, projection_dim=32, margin=0.3,
                         use_wandb=False, accumulation_steps=1, train_dataset=items, tokenizer=IntTokenizer(),
                         pretrained_model=EmbeddingOnlyEncoder(table), seed=5 if rank == 0 else None, feature_cache=cache)
        flat = torch.cat([p.detach().reshape(-1).cpu() for p in m.trainable_parameters()])
        init = None
        if rank == 0:
            q.put(("ok", flat.numpy().copy(), init))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback

        q.put(("error", f"rank {rank}: {e}\\n{traceback.format_exc()}", None))
        raise


@pytest.mark.parametrize("cache", [False, True])
def test_run_training_two_ranks_equal_one_rank(cache):
    """run_training launched on two ranks (global batches built identically from a broadcast seed, each rank trains its
    contiguous shard, replicas start from rank 0's projections) ends with the parameters of the one-rank run that
    starts from the same projections."""
    res = {}
    for world in (1, 2):
        res[world] = _ranks(_rt_worker, world, (cache,), salt=40 + world + (10 if cache else 0))[1]
    from conftest import assert_params_close

    # both runs start from rank 0's torch.manual_seed(100) initialisation and see the same triplets
    assert_params_close(res[2], res[1], 1e-2, 2 * 5, tight=5e-5)


def _rccl_one_rank_worker(rank, world, port, q):
    try:
        from two_towers_overlords_amd.training import TripletTrainer

        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        table, params, _, gb = _setup()
        dev_batches = [tuple(None if t is None else t.cuda() for t in b) for b in gb]
        out = []
        for exchange, split in ((False, False), (True, False), (True, True)):
            tr = TripletTrainer(table=table.cuda(), params={k: v.cuda() for k, v in params.items()}, max_batch=B,
                                max_seq_len=S, margin=0.3, lr=1e-3, exchange=exchange, split_exchange=split)
            assert (tr.bucket is not None) == exchange and tr._split == split
            losses = [tr.train_epoch(dev_batches) for _ in range(3)]
            out.append((losses, tr.flat.cpu().numpy().copy()))
        q.put(("ok", out, None))
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback

        q.put(("error", f"{e}\\n{traceback.format_exc()}", None))
        raise


def test_data_parallel_step_over_rccl_single_rank_is_the_single_replica_step():
    """The data-parallel step's own code path (slab reduce -> flat bucket -> RCCL all-reduce issued on the caller's
    stream, the next gather enqueued right before it -> Adam from the bucket) on a ONE-rank NCCL(RCCL) group, where the
    exchange is an identity: bitwise the single-replica step.  Exercises the real RCCL backend and its stream ordering
    against the engine's raw-stream launches (the multi-rank arithmetic is covered over gloo above)."""
    out = _ranks(_rccl_one_rank_worker, 1, salt=77)
    (l0, f0), (l1, f1), (l2, f2) = out[1]
    assert l0 == l1 == l2
    assert np.array_equal(f0, f1)
    assert np.array_equal(f0, f2)  # the two-part exchange (split_exchange=True): its early part on a second stream


def _dp_calibrate_worker(rank, world, port, q):
    try:
        from two_towers_overlords_amd.training import TrainerConfig, TripletTrainer

        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        table, params, _, gb = _setup()
        dev_batches = [tuple(None if t is None else t.cuda() for t in b) for b in gb]
        out = {}
        for name, split in (("fixed", False), ("auto", None)):
            tr = TripletTrainer(table=table.cuda(), params={k: v.cuda() for k, v in params.items()}, max_batch=B, max_seq_len=S,
                                config=TrainerConfig(margin=0.3, lr=1e-3, exchange=True, split_exchange=split, accumulation_steps=2))
            tr.train_epoch(dev_batches)                      # leaves an un-applied gradient in the bucket (3 batches, k = 2)
            before = [t.clone() for t in (tr.flat, tr.exp_avg, tr.exp_avg_sq, tr.bucket)]
            state = (tr.step_count, tr._micro, tr._carry)
            cal = tr.calibrate(dev_batches[0], steps=6, rounds=1)
            same = all(torch.equal(a, b) for a, b in zip(before, (tr.flat, tr.exp_avg, tr.exp_avg_sq, tr.bucket)))
            out[name] = (cal, same, state == (tr.step_count, tr._micro, tr._carry), tr._split,
                         [tr.train_epoch(dev_batches) for _ in range(2)], tr.flat.cpu().numpy().copy())
        q.put(("ok", out, None))
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback

        q.put(("error", f"{e}\n{traceback.format_exc()}", None))
        raise


def test_calibrate_times_the_data_parallel_step_and_leaves_the_run_untouched():
    """Round 5: with a gradient exchange calibrate()'s dry steps are the data-parallel step itself (slab reduce into a
    scratch bucket, the gather under the real RCCL collective) and, where the caller left the exchange form open
    (split_exchange=None), both forms are candidates.  Nothing of the run moves: parameters, Adam moments, step counters
    and an un-applied accumulation window in the bucket; the epochs that follow are the same whatever was picked."""
    out = _ranks(_dp_calibrate_worker, 1, salt=91)[1]
    cal, same, counters, split, losses_f, flat_f = out["fixed"]
    assert cal["regime"].startswith("data-parallel") and cal["exchange_form"] == "one-bucket" and not split
    assert same and counters and not any("two-part" in k for k in cal["ms_per_step_by_fence"])
    cal_a, same_a, counters_a, split_a, losses_a, flat_a = out["auto"]
    assert same_a and counters_a and any("two-part" in k for k in cal_a["ms_per_step_by_fence"])
    assert cal_a["exchange_form"] == ("two-part" if split_a else "one-bucket")
    assert losses_f == losses_a and np.array_equal(flat_f, flat_a)   # both exchange forms sum the same elements


# --------------------------------------------------------------------------- two-part exchange, two ranks
def _split_worker(rank, world, port, backend, q):
    try:
        from two_towers_overlords_amd.training import TripletTrainer, shard_batch

        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dev = rank if backend == "nccl" else 0
        torch.cuda.set_device(dev)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        table, params, _, gb = _setup()
        res = {}
        for split in (False, True):
            tr = TripletTrainer(table=table.cuda(), params={k: v.cuda() for k, v in params.items()}, max_batch=B,
                                max_seq_len=S, margin=0.3, lr=1e-3, split_exchange=split)
            shards, sizes = zip(*(shard_batch(tuple(None if t is None else t.cuda() for t in b), rank, world) for b in gb))
            losses = [tr.train_epoch(list(shards), global_batch_sizes=list(sizes)) for _ in range(2)]
            res[split] = (losses, tr.flat.cpu().numpy().copy(), tr.replica_divergence())
        if rank == 0:
            q.put(("ok", res, None))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback

        q.put(("error", f"rank {rank}: {e}\n{traceback.format_exc()}", None))
        raise


def _run_split(backend):
    return _ranks(_split_worker, 2, (backend,), salt=90 + (17 if backend == "nccl" else 0))[1]


def _check_split(res):
    (l0, f0, d0), (l1, f1, d1) = res[False], res[True]
    assert d0 == 0.0 and d1 == 0.0          # replicas in step
    assert l0 == l1 and np.array_equal(f0, f1)  # two replicas: a + b either way, bit for bit
    table, params, batches, _ = _setup()
    st = orc.AdamState(lr=1e-3)
    for _ in range(2):
        ref_avg, _ = orc.train_epoch(table, batches, params, st, 0.3)
    assert abs(l1[-1] - ref_avg) <= 1e-5


def test_split_exchange_two_ranks_is_bitwise_the_one_bucket_exchange():
    """split_exchange=True: the two-part exchange with the real HIP engine (dW2 / db2 reduced and exchanged on a second stream
    behind the engine's `dw2` event while dH -> dW1 still run), two ranks sharing the box's GPU over gloo."""
    _check_split(_run_split("gloo"))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: one RCCL rank per GPU")
def test_two_ranks_over_rccl_both_exchange_forms():
    """The same over RCCL (nccl backend), one rank per GPU — runs wherever two GPUs are visible."""
    _check_split(_run_split("nccl"))


# --------------------------------------------------------------------------- the drop-in route under a process group
def _route_worker(rank, world, port, q):
    """Each rank runs the reference's literal call — train_epoch(model, batches, TripletLoss, torch.optim.Adam) — on ITS OWN
    batches under an initialised process group: the route must take it (round 5 declined) and must not exchange anything."""
    try:
        from conftest import load_golden
        from two_towers_overlords_amd.model import EmbeddingOnlyEncoder, TripletLoss, TwoTowersModel
        from two_towers_overlords_amd.training import train_epoch

        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        g = load_golden("step_mid.npz")
        T = torch.from_numpy
        P_ = g["init/query_tower.projection.0.bias"].shape[0]
        m = TwoTowersModel(projection_dim=P_, tokenizer=lambda *a, **k: None, pretrained_model=EmbeddingOnlyEncoder(T(g["table"])),
                           encoder_mode="embed")
        sd = m.state_dict()
        for k in orc.PARAM_KEYS:
            sd[k] = T(g["init/" + k])
        m.load_state_dict(sd)
        m = m.cuda()
        nb = int(g["n_batches"])
        order = list(range(nb)) if rank == 0 else list(reversed(range(nb)))  # the ranks see DIFFERENT data: nothing may be averaged
        batches = [tuple((T(g[f"batch{i}/ids_{r}"]).cuda(), T(g[f"batch{i}/mask_{r}"]).cuda()) for r in "qpn") for i in order]
        opt = torch.optim.Adam(m.parameters(), lr=float(g["lr"]))
        avg = train_epoch(m, batches, TripletLoss(margin=float(g["margin"])), opt, log_wandb=False)
        routed = getattr(opt, "_tt_route", None) is not None and opt._tt_route.trainer.world == 1
        out = [None] * world
        dist.all_gather_object(out, (avg, routed, {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items() if ".projection." in k}))
        if rank == 0:
            q.put(("ok", out, None))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback

        q.put(("error", f"rank {rank}: {e}\n{traceback.format_exc()}", None))
        raise


def test_train_epoch_route_under_a_process_group_trains_each_process_on_its_own():
    from conftest import assert_params_close, load_golden

    _, out, _ = _ranks(_route_worker, 2, salt=7)
    g = load_golden("step_mid.npz")
    (avg0, routed0, p0), (avg1, routed1, p1) = out
    assert routed0 and routed1
    assert abs(avg0 - float(g["epoch/avg_loss"])) <= 1e-5           # rank 0 saw the reference's order: the reference's epoch
    for k in orc.PARAM_KEYS:
        assert_params_close(p0[k], g["final/" + k], float(g["lr"]), int(g["n_batches"]))
    assert any(np.abs(p0[k] - p1[k]).max() > 1e-6 for k in orc.PARAM_KEYS)  # rank 1 trained on its own order: other weights
