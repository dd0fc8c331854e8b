#!/usr/bin/env python3
"""
Generates tests/golden/*.npz by running the REFERENCE's own code
(/root/reference/backend/{model,training,data}.py) in the build container.

Run (container only — /root/reference does not exist on the GPU box, and
nothing under tests/ reads it at test time):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Only inputs and expected outputs are written (data); no reference source is
copied.  The fetching constructors (`from_pretrained`, `load_dataset`) are
bypassed with `__new__` + `nn.Module.__init__`, and every arithmetic line of
the reference (`AveragePoolingTower.forward/_mean_pooling`, `TwoTowersModel.
encode_*`, `TripletLoss`, `training.train_epoch`, `TripletDataLoader.
create_triplets`) is executed unchanged.

Version note: fixtures come from torch {torch.__version__} / transformers as
installed here; the reference pins torch 2.7.1 / transformers 4.52.4.  The ops
on the path (embedding, mul, sum, clamp, div, normalize, linear, relu,
cosine_similarity, mean, Adam) are stable across those versions.
"""

import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/backend")
sys.dont_write_bytecode = True

import model as ref_model  # noqa: E402  (reference)
import training as ref_training  # noqa: E402  (reference)
import data as ref_data  # noqa: E402  (reference)

from oracle import two_towers_oracle as orc  # noqa: E402  (only for the closed-form / seeded INPUT generators)

torch.set_num_threads(1)  # deterministic reduction order for the fixtures


class _EmbedOnly(torch.nn.Module):
    """`embed` encoder mode: pretrained model reduced to the word-embedding table."""

    def __init__(self, table):
        super().__init__()
        self.emb = torch.nn.Embedding.from_pretrained(table.clone(), freeze=True)

    @property
    def device(self):
        return self.emb.weight.device

    def forward(self, input_ids=None, **kw):
        return (self.emb(input_ids),)


class _StubTokenizer:
    """Returns prepared int64 tensors for a tag string (texts[0])."""

    def __init__(self):
        self.registry = {}

    def __call__(self, texts, **kw):
        ids, mask = self.registry[texts[0]]
        assert len(texts) == ids.shape[0]
        return {"input_ids": ids.clone(), "attention_mask": mask.clone()}


def build_ref_tower(pretrained, tok, D, P, w1, b1, w2, b2):
    t = ref_model.AveragePoolingTower.__new__(ref_model.AveragePoolingTower)
    torch.nn.Module.__init__(t)
    t.tokenizer = tok
    t.pretrained_model = pretrained
    t.embedding_dim = D
    for p in t.pretrained_model.parameters():
        p.requires_grad = False
    t.projection = torch.nn.Sequential(torch.nn.Linear(D, P), torch.nn.ReLU(), torch.nn.Linear(P, P))
    with torch.no_grad():
        t.projection[0].weight.copy_(w1)
        t.projection[0].bias.copy_(b1)
        t.projection[2].weight.copy_(w2)
        t.projection[2].bias.copy_(b2)
    return t


def build_ref_model(table, params, tok):
    D = table.shape[1]
    P = params["query_tower.projection.0.bias"].shape[0]
    m = ref_model.TwoTowersModel.__new__(ref_model.TwoTowersModel)
    torch.nn.Module.__init__(m)
    for tower in ("query_tower", "document_tower"):
        k = tower + ".projection."
        setattr(
            m,
            tower,
            build_ref_tower(
                _EmbedOnly(table), tok, D, P,
                params[k + "0.weight"], params[k + "0.bias"], params[k + "2.weight"], params[k + "2.bias"],
            ),
        )
    return m


def trainable_state(m):
    return {k: v.detach().clone() for k, v in m.state_dict().items() if ".projection." in k}


def np_(x):
    return x.detach().cpu().numpy()


def register_batch(tok, batch, tag):
    names = []
    for role, (ids, mask) in zip("qpn", batch):
        key = f"{tag}:{role}"
        tok.registry[key] = (ids, mask)
        names.append([key] * ids.shape[0])
    return tuple(names)


def full_step_fixture(name, table, batches, params, margin, lr):
    """Reference forward + TripletLoss + backward on batch 0, then reference
    train_epoch over all batches with torch.optim.Adam(model.parameters(), lr)."""
    tok = _StubTokenizer()
    out = {"table": np_(table), "margin": np.float64(margin), "lr": np.float64(lr), "n_batches": np.int64(len(batches))}
    for k, v in params.items():
        out["init/" + k] = np_(v)

    # --- single forward/backward on batch 0 through the reference classes
    m = build_ref_model(table, params, tok)
    crit = ref_model.TripletLoss(margin=margin)
    texts = [register_batch(tok, b, f"b{i}") for i, b in enumerate(batches)]
    for i, b in enumerate(batches):
        for role, (ids, mask) in zip("qpn", b):
            out[f"batch{i}/ids_{role}"] = np_(ids)
            out[f"batch{i}/mask_{role}"] = np_(mask)
    tq, tp, tn = texts[0]
    q = m.encode_queries(tq)
    p = m.encode_documents(tp)
    n = m.encode_documents(tn)
    for t in (q, p, n):
        t.retain_grad()
    loss = crit(q, p, n)
    loss.backward()
    out["fwd/q"], out["fwd/p"], out["fwd/n"] = np_(q), np_(p), np_(n)
    out["fwd/loss"] = np_(loss)
    out["fwd/pos_dist"] = np_(crit.cosine_distance(q, p))
    out["fwd/neg_dist"] = np_(crit.cosine_distance(q, n))
    out["bwd/dq"], out["bwd/dp"], out["bwd/dn"] = np_(q.grad), np_(p.grad), np_(n.grad)
    for k, v in m.named_parameters():
        if v.requires_grad:
            out["grad/" + k] = np_(v.grad)
    # pooled-normalised intermediates via the reference's own lines (model.py:51-56)
    for role, (ids, mask) in zip("qpn", batches[0]):
        tower = m.query_tower if role == "q" else m.document_tower
        with torch.no_grad():
            o = tower.pretrained_model(input_ids=ids, attention_mask=mask)
        pooled = tower._mean_pooling(o, mask)
        out[f"fwd/pooled_{role}"] = np_(torch.nn.functional.normalize(pooled, p=2, dim=1))

    # --- reference train_epoch over all batches, fresh model
    m2 = build_ref_model(table, params, tok)
    opt = torch.optim.Adam(m2.parameters(), lr=lr)
    losses = []

    class _Spy(ref_model.TripletLoss):
        def forward(self, a, p_, n_):
            l = super().forward(a, p_, n_)
            losses.append(float(l.detach()))
            return l

    avg = ref_training.train_epoch(m2, texts, _Spy(margin=margin), opt, log_wandb=False)
    out["epoch/avg_loss"] = np.float64(avg)
    out["epoch/loss_trace"] = np.array(losses, dtype=np.float64)
    for k, v in trainable_state(m2).items():
        out["final/" + k] = np_(v)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(f"{name}: loss0={float(loss):.9f} avg={avg:.9f} trace={losses}")
    return out


def g_kat0():
    table, batch, params, margin = orc.kat0_inputs()
    return full_step_fixture("kat0.npz", table, [batch, batch, batch], params, margin, lr=1e-3)


def g_step_mid():
    """Mid-size seeded case: ragged lengths, B not a multiple of any tile, 3 distinct batches."""
    V, D, P, B, S = 500, 48, 40, 33, 17
    table = orc.synthetic_table(V, D, seed=11) * 50.0  # O(1) entries
    params = orc.init_params(D, P, seed=12)
    batches = [orc.synthetic_batch(B, S, V, seed=100 + i, ragged=True) for i in range(3)]
    return full_step_fixture("step_mid.npz", table, batches, params, margin=0.3, lr=1e-3)


def g_step_minilm_shape():
    """MiniLM-shaped D=384, P=64 (BASELINE configs[0] dims), small V/B so the file stays small."""
    V, D, P, B, S = 300, 384, 64, 16, 24
    table = orc.synthetic_table(V, D, seed=21) * 50.0
    params = orc.init_params(D, P, seed=22)
    batches = [orc.synthetic_batch(B, S, V, seed=200 + i, ragged=True) for i in range(2)]
    return full_step_fixture("step_d384.npz", table, batches, params, margin=0.3, lr=1e-3)


def g_pool_edges():
    """K1 edge cases through the reference's `_mean_pooling` + F.normalize:
    ragged prefix masks, S=1-long row, an all-padding row (clamp 1e-9 path ->
    zero vector -> 1e-12 normalise path), a non-prefix (holey) mask, repeated ids."""
    g = torch.Generator().manual_seed(5)
    V, D, B, S = 97, 24, 7, 9
    table = torch.randn(V, D, generator=g)
    ids = torch.randint(0, V, (B, S), generator=g)
    lens = torch.tensor([9, 1, 0, 5, 9, 3, 2])
    mask = (torch.arange(S).unsqueeze(0) < lens.unsqueeze(1)).long()
    mask[4] = torch.tensor([1, 0, 1, 1, 0, 0, 1, 0, 1])  # holey mask
    ids[5] = 13  # same row repeated
    t = ref_model.AveragePoolingTower.__new__(ref_model.AveragePoolingTower)
    emb = torch.nn.functional.embedding(ids, table)
    pooled = ref_model.AveragePoolingTower._mean_pooling(t, (emb,), mask)
    out = torch.nn.functional.normalize(pooled, p=2, dim=1)
    np.savez_compressed(
        os.path.join(HERE, "pool_edges.npz"),
        table=np_(table), ids=np_(ids), mask=np_(mask), hidden=np_(emb), pooled=np_(pooled), out=np_(out),
    )
    print("pool_edges.npz: row2 (all pad) =", out[2, :3].tolist())


def g_loss_edges():
    """K3 edge cases through the reference TripletLoss: a zero anchor row and a
    zero negative row (cosine eps 1e-8 clamp path), inactive hinge rows, P not a
    multiple of 64."""
    g = torch.Generator().manual_seed(6)
    B, P = 11, 37
    q = torch.randn(B, P, generator=g)
    p = torch.randn(B, P, generator=g)
    n = torch.randn(B, P, generator=g)
    q[3] = 0
    n[5] = 0
    p[7] = q[7] * 2.0  # pos_dist 0 -> inactive at margin 0.3 if neg far
    n[7] = -q[7]
    q[8] = q[8] * 1e-10  # tiny-norm anchor: below eps
    crit = ref_model.TripletLoss(margin=0.3)
    qq, pp, nn_ = (t.clone().requires_grad_(True) for t in (q, p, n))
    loss = crit(qq, pp, nn_)
    loss.backward()
    np.savez_compressed(
        os.path.join(HERE, "loss_edges.npz"),
        q=np_(q), p=np_(p), n=np_(n), margin=np.float64(0.3), loss=np_(loss),
        pos_dist=np_(crit.cosine_distance(q, p)), neg_dist=np_(crit.cosine_distance(q, n)),
        dq=np_(qq.grad), dp=np_(pp.grad), dn=np_(nn_.grad),
    )
    print("loss_edges.npz: loss =", float(loss))


def g_bert_mode():
    """`bert` encoder mode on a tiny locally-built BertConfig (no fetch): the
    reference forward with a real BertModel between gather and pool.  The fixture
    stores the encoder's last_hidden_state (the dense [B,S,D] input of the pool
    kernel) and the tower output."""
    from transformers import BertConfig, BertModel

    torch.manual_seed(7)
    cfg = BertConfig(vocab_size=120, hidden_size=32, num_hidden_layers=2, num_attention_heads=4,
                     intermediate_size=64, max_position_embeddings=32)
    bert = BertModel(cfg).eval()
    D, P, B, S = 32, 16, 5, 12
    params = orc.init_params(D, P, seed=31)
    g = torch.Generator().manual_seed(8)
    ids = torch.randint(0, 120, (B, S), generator=g)
    lens = torch.tensor([12, 7, 1, 9, 4])
    mask = (torch.arange(S).unsqueeze(0) < lens.unsqueeze(1)).long()
    ids = ids * mask
    tok = _StubTokenizer()
    tok.registry["x"] = (ids, mask)
    k = "query_tower.projection."
    t = build_ref_tower(bert, tok, D, P, params[k + "0.weight"], params[k + "0.bias"], params[k + "2.weight"], params[k + "2.bias"])
    out = t(["x"] * B)
    with torch.no_grad():
        hidden = bert(input_ids=ids, attention_mask=mask)[0]
    np.savez_compressed(
        os.path.join(HERE, "bert_mode.npz"),
        ids=np_(ids), mask=np_(mask), hidden=np_(hidden), out=np_(out),
        w1=np_(params[k + "0.weight"]), b1=np_(params[k + "0.bias"]), w2=np_(params[k + "2.weight"]), b2=np_(params[k + "2.bias"]),
    )
    print("bert_mode.npz: out[0,:3] =", out[0, :3].tolist())


def g_bert_minilm():
    """`bert` encoder mode at the shape the reference actually loads (sentence-transformers/all-MiniLM-L6-v2: vocab 30522,
    hidden 384, 6 layers, 12 heads, intermediate 1536, max positions 512 — SURVEY.md Appendix A; 22,713,216 parameters),
    with seeded random weights (the pretrained ones are not in the image): the reference's AveragePoolingTower.forward,
    unchanged, with that encoder between gather and pool, ragged texts.  The weights are NOT stored (91 MB): the test
    rebuilds the same module from the seed and checks its hidden states against the stored ones first."""
    from transformers import BertConfig, BertModel

    torch.manual_seed(9)
    cfg = BertConfig(vocab_size=30522, hidden_size=384, num_hidden_layers=6, num_attention_heads=12, intermediate_size=1536,
                     max_position_embeddings=512)
    bert = BertModel(cfg).eval()
    assert sum(p.numel() for p in bert.parameters()) == 22_713_216
    D, P, B, S = 384, 64, 6, 48
    params = orc.init_params(D, P, seed=41)
    g = torch.Generator().manual_seed(10)
    ids = torch.randint(0, 30522, (B, S), generator=g)
    lens = torch.tensor([48, 31, 1, 17, 40, 9])
    mask = (torch.arange(S).unsqueeze(0) < lens.unsqueeze(1)).long()
    ids = ids * mask
    tok = _StubTokenizer()
    tok.registry["x"] = (ids, mask)
    k = "document_tower.projection."
    t = build_ref_tower(bert, tok, D, P, params[k + "0.weight"], params[k + "0.bias"], params[k + "2.weight"], params[k + "2.bias"])
    out = t(["x"] * B)
    with torch.no_grad():
        hidden = bert(input_ids=ids, attention_mask=mask)[0]
    np.savez_compressed(
        os.path.join(HERE, "bert_minilm.npz"),
        ids=np_(ids), mask=np_(mask), hidden=np_(hidden), out=np_(out),
        w1=np_(params[k + "0.weight"]), b1=np_(params[k + "0.bias"]), w2=np_(params[k + "2.weight"]), b2=np_(params[k + "2.bias"]),
    )
    print("bert_minilm.npz: out[0,:3] =", out[0, :3].tolist())


def g_triplets():
    """`TripletDataLoader.create_triplets` (data.py:113-137) on fake items with
    seeded `random`: expected negative index per item for several seeds.  Cases 4-5
    hold items whose passage is the empty string: data.py:126 (`if negative and ...`)
    never takes such a passage as a negative."""
    dl = ref_data.TripletDataLoader.__new__(ref_data.TripletDataLoader)
    out = {}
    cases = ((8, 3, 0, 0), (32, 5, 1, 0), (64, 40, 2, 0), (5, 2, 3, 0), (24, 6, 4, 5), (9, 3, 5, 3))
    for case, (n_items, n_q, seed, n_empty) in enumerate(cases):
        rq = random.Random(1000 + case)
        qids = [rq.randrange(n_q) for _ in range(n_items)]
        if len(set(qids)) == 1:
            qids[0] = (qids[0] + 1) % n_q
        empty = set(rq.sample(range(n_items), n_empty))
        docs = ["" if i in empty else f"doc{i}" for i in range(n_items)]
        batch = [{"query": f"q{qids[i]}", "positive": docs[i], "query_id": qids[i]} for i in range(n_items)]
        random.seed(seed)
        queries, positives, negatives = dl.create_triplets(batch)
        assert all(negatives), "the reference never returns an empty negative"
        neg_idx = [int(s[3:]) for s in negatives]
        assert positives == docs
        out[f"case{case}/query_ids"] = np.array(qids, dtype=np.int64)
        out[f"case{case}/seed"] = np.int64(seed)
        out[f"case{case}/neg_idx"] = np.array(neg_idx, dtype=np.int64)
        out[f"case{case}/positive_nonempty"] = np.array([bool(d) for d in docs])
    np.savez_compressed(os.path.join(HERE, "triplets.npz"), **out)
    print("triplets.npz written")


class _IntTokenizer:
    """Deterministic stand-in for the WordPiece tokenizer: a text is a space-separated list of
    token ids; `padding=True` pads to the longest text of the call with 0 / mask 0."""

    def __call__(self, texts, **kw):
        rows = [[int(w) for w in t.split()] for t in texts]
        S = max(len(r) for r in rows)
        ids = torch.zeros(len(rows), S, dtype=torch.int64)
        mask = torch.zeros(len(rows), S, dtype=torch.int64)
        for i, r in enumerate(rows):
            ids[i, : len(r)] = torch.tensor(r)
            mask[i, : len(r)] = 1
        return {"input_ids": ids, "attention_mask": mask}


class _EvalDataset:
    """map-style {"query_id","query","positive"} + get_unique_passages(), like MSMarcoDataset."""

    def __init__(self, items):
        self.items = items
        self.docs = list(dict.fromkeys(it["positive"] for it in items))

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]

    def get_unique_passages(self):
        return self.docs


def g_eval():
    """`evaluate_model` (training.py:136-380) of the reference itself, CPU, on a synthetic corpus:
    NDCG@{1,5,10} for the all-documents mode (candidate_pool_size=-1) and the per-query pool mode
    (pool larger than the irrelevant set, so no hash-order-dependent sampling is involved)."""
    rng = random.Random(77)
    V, D, P = 400, 48, 32
    table = orc.synthetic_table(V, D, seed=41) * 50.0
    params = orc.init_params(D, P, seed=42)
    n_queries, items = 40, []
    for qid in range(n_queries):
        q_tokens = [rng.randrange(1, V) for _ in range(rng.randint(3, 8))]
        query = " ".join(map(str, q_tokens))
        for _ in range(rng.randint(1, 6)):
            # a relevant passage shares some tokens with its query
            doc = [rng.choice(q_tokens) if rng.random() < 0.5 else rng.randrange(1, V) for _ in range(rng.randint(5, 30))]
            items.append({"query_id": 1000 + qid, "query": query, "positive": " ".join(map(str, doc))})
    items.append(dict(items[3]))  # an exact duplicate pair -> tied scores in the all-documents mode
    items[-1]["query_id"], items[-1]["query"] = items[20]["query_id"], items[20]["query"]
    ds = _EvalDataset(items)
    model = build_ref_model(table, params, _IntTokenizer())
    out = {
        "table": np_(table),
        "items_query_id": np.array([it["query_id"] for it in items], dtype=np.int64),
        "items_query": np.array([it["query"] for it in items]),
        "items_positive": np.array([it["positive"] for it in items]),
    }
    for k, v in params.items():
        out["init/" + k] = np_(v)
    cases = [  # (seed, sample_size, min_query_groups, candidate_pool_size)
        (5, 60, 12, -1),
        (6, 80, 15, 10_000),
        (7, 30, 40, -1),  # asks for more query groups than one sample holds -> the multiplier loop runs
    ]
    out["cases"] = np.array(cases, dtype=np.int64)
    for c, (seed, ss, mq, pool) in enumerate(cases):
        random.seed(seed)
        res = ref_training.evaluate_model(model, ds, sample_size=ss, min_query_groups=mq, candidate_pool_size=pool,
                                          comprehensive=True, log_wandb=False, wandb_prefix="t_", batch_size=64)
        for k, v in res.items():
            out[f"case{c}/{k}"] = np.float64(v)
        random.seed(seed)
        out[f"case{c}/simple_ndcg10"] = np.float64(
            ref_training.evaluate_model(model, ds, sample_size=ss, min_query_groups=mq, candidate_pool_size=pool,
                                        comprehensive=False, batch_size=64))
        print(f"eval case {c}: ndcg@10 {res['t_ndcg_10']:.6f} ndcg@1 {res['t_ndcg_1']:.6f} queries {res['t_queries_evaluated']}")
    np.savez_compressed(os.path.join(HERE, "eval.npz"), **out)


def g_accum():
    """`training.train_epoch_optimized` (training.py:66-133) of the reference itself, run UNCHANGED on the CPU in fp32:
    `torch.set_autocast_dtype("cpu", torch.float32)` makes `autocast("cpu")` a no-op ("target dtype is not supported.
    Disabling autocast": Linear stays fp32) and `GradScaler("cpu", enabled=False)` scales by 1, so what is left is the
    accumulation arithmetic: loss / k, `optimizer.step()` + `zero_grad()` when (batch_idx + 1) % k == 0 with batch_idx
    restarting every epoch, the gradient of an epoch's unpaired last batch neither applied nor cleared (it is still
    in `.grad` when the next epoch starts), average of the un-scaled per-batch losses.  TWO epochs of FIVE batches
    with k = 2: epoch 1 steps after batches 1 and 3 and leaves batch 4's gradient behind; epoch 2's first step
    (after ITS batch 1) therefore applies three micro-gradients.  model.train() is called by the function; the
    embed-only encoder has no dropout, so the quirk SURVEY.md §3.1 notes does not bite here."""
    import warnings

    V, D, P, B, S = 400, 40, 24, 19, 13
    k, n_epochs, n_batches = 2, 2, 5
    table = orc.synthetic_table(V, D, seed=51) * 50.0
    params = orc.init_params(D, P, seed=52)
    batches = [orc.synthetic_batch(B, S, V, seed=300 + i, ragged=True) for i in range(n_batches)]
    margin, lr = 0.3, 1e-3
    tok = _StubTokenizer()
    out = {"table": np_(table), "margin": np.float64(margin), "lr": np.float64(lr), "n_batches": np.int64(n_batches),
           "n_epochs": np.int64(n_epochs), "accumulation_steps": np.int64(k)}
    for kk, v in params.items():
        out["init/" + kk] = np_(v)
    texts = [register_batch(tok, b, f"b{i}") for i, b in enumerate(batches)]
    for i, b in enumerate(batches):
        for role, (ids, mask) in zip("qpn", b):
            out[f"batch{i}/ids_{role}"] = np_(ids)
            out[f"batch{i}/mask_{role}"] = np_(mask)
    m = build_ref_model(table, params, tok)
    opt = torch.optim.Adam(m.parameters(), lr=lr)
    losses = []

    class _Spy(ref_model.TripletLoss):
        def forward(self, a, p_, n_):
            l = super().forward(a, p_, n_)
            assert l.dtype == torch.float32 and a.dtype == torch.float32  # autocast really is off
            losses.append(float(l.detach()))
            return l

    torch.set_autocast_dtype("cpu", torch.float32)
    scaler = torch.GradScaler("cpu", enabled=False)
    avgs = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # "In CPU autocast, but the target dtype is not supported. Disabling autocast."
        for e in range(n_epochs):
            avgs.append(ref_training.train_epoch_optimized(m, texts, _Spy(margin=margin), opt, torch.device("cpu"), scaler,
                                                           accumulation_steps=k, log_wandb=False))
            for kk, v in trainable_state(m).items():
                out[f"epoch{e}/" + kk] = np_(v)
    out["epoch/avg_loss"] = np.array(avgs, dtype=np.float64)
    out["epoch/loss_trace"] = np.array(losses, dtype=np.float64)
    out["optimizer_steps"] = np.int64(next(iter(opt.state.values()))["step"].item())
    # what is still in .grad after the last epoch: the carried (un-applied) gradient of its 5th batch, already / k
    for kk, v in m.named_parameters():
        if v.requires_grad:
            out["carry/" + kk] = np_(v.grad)
    np.savez_compressed(os.path.join(HERE, "step_accum.npz"), **out)
    print(f"step_accum.npz: avgs={avgs} optimizer steps={int(out['optimizer_steps'])} trace={losses}")


if __name__ == "__main__":
    if sys.argv[1:] == ["triplets"]:  # regenerate only the sampling fixture
        g_triplets()
        sys.exit(0)
    if sys.argv[1:] == ["accum"]:  # regenerate only the accumulation fixture
        g_accum()
        sys.exit(0)
    if sys.argv[1:] == ["bert_minilm"]:  # only the MiniLM-L6-shaped bert-mode fixture (round 5)
        g_bert_minilm()
        sys.exit(0)
    r = g_kat0()
    g_step_mid()
    g_step_minilm_shape()
    g_pool_edges()
    g_loss_edges()
    g_bert_mode()
    g_bert_minilm()
    g_triplets()
    g_eval()
    g_accum()
    with open(os.path.join(HERE, "VERSIONS.txt"), "w") as f:
        import transformers

        f.write(f"torch {torch.__version__}\ntransformers {transformers.__version__}\nnumpy {np.__version__}\n")
        f.write("reference pins: torch 2.7.1, transformers 4.52.4 (backend/uv.lock)\n")
