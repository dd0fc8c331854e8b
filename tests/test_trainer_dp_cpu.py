"""Data-parallel / accumulation logic of TripletTrainer on CPU: world_size-2 gloo
processes (and a single process) with the oracle-backed engine must reproduce the
oracle's single-replica train_epoch on the GLOBAL batches."""
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

V, D, P, B, S, NB = 200, 16, 12, 33, 9, 4  # B odd: ranks get 17 + 16 triplets


def _setup():
    table = orc.synthetic_table(V, D, seed=3) * 50
    params = orc.init_params(D, P, seed=4)
    batches = [orc.synthetic_batch(B, S, V, seed=70 + i, ragged=True) for i in range(NB)]
    return table, params, batches


def _global_batches(batches):
    return [(torch.stack([r[0] for r in b]), torch.stack([r[1] for r in b]), None) for b in batches]


def _reference(accum=1):
    table, params, batches = _setup()
    st = orc.AdamState(lr=1e-3)
    if accum == 1:
        avg, _ = orc.train_epoch(table, batches, params, st, 0.3)
    else:
        avg, _, _ = orc.train_epoch_accum(table, batches, params, st, 0.3, accum)
    return avg, params


def _worker(rank, world, port, accum, q):
    try:
        _worker_body(rank, world, port, accum, q)
    except Exception as e:  # surface the failure instead of leaving the parent waiting on the queue
        import traceback

        q.put(("error", f"rank {rank}: {e}\n{traceback.format_exc()}", None))
        raise


def _worker_body(rank, world, port, accum, q):
    from oracle_engine import OracleEngine
    from two_towers_overlords_amd.training import TripletTrainer, shard_batch

    torch.set_num_threads(1)
    if world > 1:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    table, params, batches = _setup()
    tr = TripletTrainer(table=table, params=params, max_batch=B, max_seq_len=S, margin=0.3, lr=1e-3,
                        accumulation_steps=accum, engine=OracleEngine(table, P, B, S))
    shards, sizes = [], []
    for gb in _global_batches(batches):
        sh, n = shard_batch(gb, rank, world)
        shards.append(sh)
        sizes.append(n)
    avg = tr.train_epoch(shards, global_batch_sizes=sizes)
    if rank == 0:
        q.put((avg, {k: v.detach().numpy().copy() for k, v in tr.named_parameters().items()}, tr.step_count))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _run(world, accum, worker=None):
    from mp_util import free_port, run_ranks

    worker = worker or _worker
    port = free_port(salt=7 * world + accum + (0 if worker is _worker else 31))
    return run_ranks(worker, world, (accum,), timeout=240, join=60, port=port)


@pytest.mark.parametrize("world,accum", [(1, 1), (2, 1), (2, 2), (1, 3)])
def test_sharded_epoch_equals_single_replica(world, accum):
    avg, params, steps = _run(world, accum)
    ref_avg, ref_params = _reference(accum)
    assert steps == NB // accum
    assert abs(avg - ref_avg) < 1e-6
    for k in orc.PARAM_KEYS:
        np.testing.assert_allclose(params[k], ref_params[k].numpy(), atol=2e-6, rtol=0)


def _worker_unsynced(rank, world, port, accum, q):
    """Ranks build DIFFERENT initial projections (what nn.Linear's per-process default init gives,
    backend/model.py:33-38): the trainer must start every replica from rank 0's, and a global batch
    smaller than the world (an empty shard on rank 1) must not stall the step's all-reduce."""
    try:
        from oracle_engine import OracleEngine
        from two_towers_overlords_amd.training import TripletTrainer, shard_batch

        torch.set_num_threads(1)
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        table, params, batches = _setup()
        mine = {k: v.clone() for k, v in params.items()} if rank == 0 else orc.init_params(D, P, seed=900 + rank)
        tr = TripletTrainer(table=table, params=mine, max_batch=B, max_seq_len=S, margin=0.3, lr=1e-3,
                            accumulation_steps=1, engine=OracleEngine(table, P, B, S))
        div0 = tr.replica_divergence()
        gbs = _global_batches(batches)
        gbs.append(tuple(None if t is None else t[:, :1].contiguous() for t in gbs[0]))  # one triplet for two ranks
        shards, sizes = [], []
        for gb in gbs:
            sh, n = shard_batch(gb, rank, world)
            shards.append(sh)
            sizes.append(n)
        assert shards[-1][0].shape[1] == (1 if rank == 0 else 0)
        avg = tr.train_epoch(shards, global_batch_sizes=sizes)
        div1 = tr.replica_divergence()
        if rank == 0:
            q.put((avg, {k: v.detach().numpy().copy() for k, v in tr.named_parameters().items()}, (div0, div1)))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:
        import traceback

        q.put(("error", f"rank {rank}: {e}\n{traceback.format_exc()}", None))
        raise


def test_replicas_start_from_rank0_and_empty_shard_takes_part():
    avg, params, (div0, div1) = _run(2, 1, worker=_worker_unsynced)
    assert div0 == 0.0 and div1 == 0.0  # in step after construction and after the epoch
    table, ref_params, batches = _setup()
    one = tuple([r[0][:1], r[1][:1]] for r in batches[0])
    st = orc.AdamState(lr=1e-3)
    ref_avg, _ = orc.train_epoch(table, batches + [one], ref_params, st, 0.3)
    assert abs(avg - ref_avg) < 1e-6
    for k in orc.PARAM_KEYS:
        np.testing.assert_allclose(params[k], ref_params[k].numpy(), atol=2e-6, rtol=0)


def test_shard_batch_carries_query_ids():
    from two_towers_overlords_amd.training import TripletBatch, shard_batch

    tb = TripletBatch(ids=torch.arange(2 * 7 * 3).view(2, 7, 3), lens=torch.ones(2, 7).int(),
                      query_ids=torch.arange(7) * 10)
    parts = [shard_batch(tb, r, 3)[0] for r in range(3)]
    assert [p.size for p in parts] == [3, 2, 2]
    assert torch.equal(torch.cat([p.query_ids for p in parts]), tb.query_ids)
    assert all(p.query_ids.shape[0] == p.size for p in parts)


def test_shard_batch_partition():
    from two_towers_overlords_amd.training import shard_batch

    ids = torch.arange(3 * 10 * 2).view(3, 10, 2)
    lens = torch.arange(30).view(3, 10).int()
    got = [shard_batch((ids, None, lens), r, 4) for r in range(4)]
    assert [g[0][0].shape[1] for g in got] == [3, 3, 2, 2] and all(g[1] == 10 for g in got)
    assert torch.equal(torch.cat([g[0][0] for g in got], dim=1), ids)
    assert torch.equal(torch.cat([g[0][2] for g in got], dim=1), lens)


def test_neg_index_and_feature_cache_equal_three_encodes():
    """roles-2 batches (negatives by in-batch index) and frozen-feature-cache batches give the
    same epoch as the reference-shaped three-encode batches (trainer logic, oracle engine)."""
    from oracle_engine import OracleEngine
    from two_towers_overlords_amd.training import TripletBatch, TripletTrainer

    torch.set_num_threads(1)
    table, params, _ = _setup()
    g = torch.Generator().manual_seed(11)
    runs = {}
    q_ids = torch.randint(1, V, (NB, B, S), generator=g)
    p_ids = torch.randint(1, V, (NB, B, S), generator=g)
    lens = torch.randint(1, S + 1, (NB, 2, B), generator=g).int()
    neg = torch.stack([torch.tensor([(i + 1 + int(k)) % B for i, k in enumerate(torch.randint(0, B - 1, (B,), generator=g))])
                       for _ in range(NB)]).int()
    for mode in ("ids3", "index", "cache"):
        tr = TripletTrainer(table=table, params={k: v.clone() for k, v in params.items()}, max_batch=B, max_seq_len=S,
                            margin=0.3, lr=1e-3, engine=OracleEngine(table, P, B, S))
        if mode == "cache":
            flat_ids = torch.cat([q_ids.reshape(-1, S), p_ids.reshape(-1, S)])
            flat_lens = torch.cat([lens[:, 0].reshape(-1), lens[:, 1].reshape(-1)])
            mask = (torch.arange(S).unsqueeze(0) < flat_lens.unsqueeze(1)).long()
            tr.feature_cache = orc.gather_pool_normalize(table, flat_ids, mask)  # what build_feature_cache computes
        batches = []
        for i in range(NB):
            ids2 = torch.stack([q_ids[i], p_ids[i]])
            if mode == "cache":
                rows = torch.stack([torch.arange(B) + i * B, torch.arange(B) + NB * B + i * B]).int()
                batches.append(TripletBatch(cache_rows=rows, neg_index=neg[i]))
            else:
                tb = TripletBatch(ids=ids2, lens=lens[i], neg_index=neg[i])
                batches.append(tb.expand_negatives() if mode == "ids3" else tb)
        avg = tr.train_epoch(batches)
        runs[mode] = (avg, {k: v.clone() for k, v in tr.named_parameters().items()})
    for mode in ("index", "cache"):
        assert abs(runs[mode][0] - runs["ids3"][0]) < 1e-6
        for k in orc.PARAM_KEYS:
            np.testing.assert_allclose(runs[mode][1][k].numpy(), runs["ids3"][1][k].numpy(), atol=2e-6, rtol=0)


def test_trainer_config_object_and_keyword_overrides():
    """TrainerConfig (round 5): every knob of TripletTrainer in one dataclass; keywords given beside it override its
    fields (so every older call site keeps working), an unknown keyword is a TypeError like a bad argument, and the two
    spellings train identically."""
    from oracle_engine import OracleEngine
    from two_towers_overlords_amd.training import TrainerConfig, TripletTrainer

    torch.set_num_threads(1)
    table, params, batches = _setup()
    gb = [(torch.stack([r[0] for r in b]), torch.stack([r[1] for r in b]), None) for b in batches]
    mk = lambda **kw: TripletTrainer(table=table, params={k: v.clone() for k, v in params.items()}, max_batch=B,
                                     max_seq_len=S, engine=OracleEngine(table, P, B, S), **kw)
    a = mk(margin=0.25, lr=2e-3, accumulation_steps=2)
    b = mk(config=TrainerConfig(margin=0.9, lr=2e-3, accumulation_steps=2), margin=0.25)   # the keyword wins
    assert (b.config.margin, b.config.lr, b.config.accumulation_steps, b.margin) == (0.25, 2e-3, 2, 0.25)
    assert a.config == b.config and TrainerConfig().loss == "triplet" and TrainerConfig().split_exchange is False
    assert a.train_epoch(gb) == b.train_epoch(gb)
    for k in orc.PARAM_KEYS:
        assert torch.equal(a.named_parameters()[k], b.named_parameters()[k])
    with pytest.raises(TypeError):
        mk(learning_rate=1e-3)
    with pytest.raises(ValueError):
        mk(loss="softmax")


@pytest.mark.parametrize("accum", [1, 3])
def test_trainer_state_dict_resumes_exactly(accum):
    """Two epochs straight == one epoch, state_dict() -> a NEW trainer -> load_state_dict(), one more epoch: parameters,
    Adam moments, step counters and (with accumulation) the carried gradient bucket all continue bit for bit."""
    from oracle_engine import OracleEngine
    from two_towers_overlords_amd.training import TripletTrainer

    torch.set_num_threads(1)
    table, params, batches = _setup()
    gbs = _global_batches(batches)

    def trainer(p):
        return TripletTrainer(table=table, params={k: v.clone() for k, v in p.items()}, max_batch=B, max_seq_len=S,
                              margin=0.3, lr=1e-3, accumulation_steps=accum, engine=OracleEngine(table, P, B, S))

    a = trainer(params)
    la = [a.train_epoch(gbs), a.train_epoch(gbs)]
    b = trainer(params)
    lb = [b.train_epoch(gbs)]
    sd = b.state_dict()
    c = trainer(orc.init_params(D, P, seed=999))  # different weights: everything must come from the checkpoint
    c.load_state_dict(sd)
    lb.append(c.train_epoch(gbs))
    assert la == lb and a.step_count == c.step_count and a._micro == c._micro
    assert torch.equal(a.flat, c.flat) and torch.equal(a.exp_avg, c.exp_avg) and torch.equal(a.exp_avg_sq, c.exp_avg_sq)


def _worker_split(rank, world, port, accum, q):
    """One epoch with the exchange in one bucket, then the same epoch from the same start with the two-part (split_exchange=True)
    exchange (second-layer gradients first, first-layer gradients after): same per-element sums, so with two replicas
    the parameters must be bitwise equal.  The last global batch holds one triplet: rank 1's shard is empty."""
    try:
        from oracle_engine import OracleEngine
        from two_towers_overlords_amd.training import TripletTrainer, shard_batch

        torch.set_num_threads(1)
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        table, params, batches = _setup()
        gbs = _global_batches(batches)
        gbs.append(tuple(None if t is None else t[:, :1].contiguous() for t in gbs[0]))
        gbs.append(gbs[1])
        out = {}
        for split in (False, True):
            tr = TripletTrainer(table=table, params={k: v.clone() for k, v in params.items()}, max_batch=B, max_seq_len=S,
                                margin=0.3, lr=1e-3, accumulation_steps=accum, engine=OracleEngine(table, P, B, S),
                                split_exchange=split)
            assert tr._split == split
            shards, sizes = zip(*(shard_batch(gb, rank, world) for gb in gbs))
            avg = tr.train_epoch(list(shards), global_batch_sizes=list(sizes))
            out[split] = (avg, tr.flat.clone(), tr.exp_avg.clone(), tr.exp_avg_sq.clone(), tr.step_count)
        same = all(torch.equal(a, b) for a, b in zip(out[False][1:4], out[True][1:4]))
        if rank == 0:
            q.put((same, (out[False][0], out[True][0]), (out[False][4], out[True][4])))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:
        import traceback

        q.put(("error", f"rank {rank}: {e}\n{traceback.format_exc()}", None))
        raise


@pytest.mark.parametrize("accum", [1, 2])
def test_split_exchange_is_bitwise_the_one_bucket_exchange(accum):
    same, avgs, steps = _run(2, accum, worker=_worker_split)
    assert same and avgs[0] == avgs[1] and steps[0] == steps[1] == 6 // accum


def _accum_golden():
    from conftest import load_golden

    g = load_golden("step_accum.npz")
    T = torch.from_numpy
    table = T(g["table"])
    params = {k: T(g["init/" + k]).clone() for k in orc.PARAM_KEYS}
    gbs = [(torch.stack([T(g[f"batch{i}/ids_{r}"]) for r in "qpn"]), torch.stack([T(g[f"batch{i}/mask_{r}"]) for r in "qpn"]),
            None) for i in range(int(g["n_batches"]))]
    return g, table, params, gbs


def _worker_accum_golden(rank, world, port, accum, q):
    try:
        from oracle_engine import OracleEngine
        from two_towers_overlords_amd.training import TripletTrainer, shard_batch

        torch.set_num_threads(1)
        if world > 1:
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
            dist.init_process_group("gloo", rank=rank, world_size=world)
        g, table, params, gbs = _accum_golden()
        Bg, Sg = gbs[0][0].shape[1:]
        tr = TripletTrainer(table=table, params=params, max_batch=Bg, max_seq_len=Sg, margin=float(g["margin"]),
                            lr=float(g["lr"]), accumulation_steps=accum, engine=OracleEngine(table, params[orc.PARAM_KEYS[1]].shape[0], Bg, Sg))
        shards, sizes = zip(*(shard_batch(gb, rank, world) for gb in gbs))
        res = []
        for _ in range(int(g["n_epochs"])):
            avg = tr.train_epoch(list(shards), global_batch_sizes=list(sizes))
            res.append((avg, {k: v.detach().numpy().copy() for k, v in tr.named_parameters().items()}, tr.step_count))
        if rank == 0:
            q.put(("ok", res, None))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
    except Exception as e:
        import traceback

        q.put(("error", f"rank {rank}: {e}\n{traceback.format_exc()}", None))
        raise


@pytest.mark.parametrize("world", [1, 2])
def test_accumulation_across_epochs_matches_the_reference_train_epoch_optimized(world):
    """The trainer's accumulation logic (oracle-backed engine; 1 replica and 2 gloo replicas with 10 + 9 shards)
    against tests/golden/step_accum.npz — the reference's own `train_epoch_optimized` in fp32 over two epochs of
    five batches, k = 2: steps after batches 1 and 3 of each epoch, epoch 1's fifth gradient applied together with
    epoch 2's first two (backend/training.py:93-102)."""
    g = _accum_golden()[0]
    _, res, _ = _run(world, int(g["accumulation_steps"]), worker=_worker_accum_golden)
    assert [r[2] for r in res] == [2, 4] and int(g["optimizer_steps"]) == 4
    from conftest import assert_params_close

    for e, (avg, params, steps) in enumerate(res):
        assert abs(avg - float(g["epoch/avg_loss"][e])) < 2e-6
        for k in orc.PARAM_KEYS:
            assert_params_close(params[k], g[f"epoch{e}/" + k], float(g["lr"]), steps, tight=2e-6)
