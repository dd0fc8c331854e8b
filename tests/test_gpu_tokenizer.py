"""The input boundary with a REAL tokenizer: `transformers.BertTokenizerFast` built offline from
a vocab file this test writes, called exactly like backend/model.py:43-45 (`tokenizer(texts, padding=True,
truncation=True, return_tensors="pt", max_length=512)`).  Its int64 `input_ids` / `attention_mask` must reach the
gather bit for bit, batches must be padded to the longest text of the call like `padding=True` pads them, and the
string-level entry points (`TwoTowersModel.forward`, `TokenTripletLoader`, `run_training`) must agree with the oracle
on those very ids."""
import os
import random
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import two_towers_oracle as orc  # noqa: E402

pytestmark = pytest.mark.gpu

WORDS = ("what is the capital of france paris river long a an and run walk city country where who when how many people "
         "live in on at by for with from to europe asia africa america ocean sea lake mountain tall deep old new first last "
         "big small red blue green king queen president war peace year day night food water bread rice fish").split()
PIECES = ["##s", "##ing", "##ed", "##er", "##ly", "##est", ".", "?", ",", "!"]


@pytest.fixture(scope="module")
def tok(tmp_path_factory):
    transformers = pytest.importorskip("transformers")
    d = tmp_path_factory.mktemp("vocab")
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + [f"[unused{i}]" for i in range(20)] + WORDS + PIECES
    (d / "vocab.txt").write_text("\n".join(vocab) + "\n")
    t = transformers.BertTokenizerFast.from_pretrained(str(d))  # a local directory: nothing is fetched
    assert t.is_fast and t.vocab_size == len(vocab)
    return t


def _texts(n, rng, lo=1, hi=14):
    out = []
    for _ in range(n):
        ws = [rng.choice(WORDS) + rng.choice(["", "", "s", "ing", "ed"]) for _ in range(rng.randint(lo, hi))]
        out.append(" ".join(ws).capitalize() + rng.choice(["?", ".", ""]))
    return out


def _model(tok, P, seed):
    from two_towers_overlords_amd.model import EmbeddingOnlyEncoder, TwoTowersModel

    table = orc.synthetic_table(tok.vocab_size, 64, seed=seed) * 30
    torch.manual_seed(seed)
    return TwoTowersModel(P, tokenizer=tok, pretrained_model=EmbeddingOnlyEncoder(table)).cuda(), table


def test_real_tokenizer_ids_reach_the_gather_bit_exactly_and_forward_matches_the_oracle(tok):
    from two_towers_overlords_amd import ops

    rng = random.Random(1)
    queries, docs = _texts(37, rng, 1, 8), _texts(37, rng, 3, 40)
    model, table = _model(tok, 48, 3)
    for texts, tower, name in ((queries, model.query_tower, "query_tower"), (docs, model.document_tower, "document_tower")):
        enc = tok(texts, padding=True, truncation=True, return_tensors="pt", max_length=512)  # backend/model.py:43-45
        ids, mask = enc["input_ids"], enc["attention_mask"]
        assert ids.dtype == torch.int64 and mask.dtype == torch.int64
        assert ids.shape[1] == int(mask.sum(1).max())  # padding=True: the longest text of THIS call, no more
        assert (ids[:, 0] == tok.cls_token_id).all() and ((ids == tok.sep_token_id).sum(1) == 1).all()  # [CLS] .. [SEP] are pooled too
        echo = ops.embed_gather(table.cuda(), ids.cuda())           # the rows K1 reads for these int64 ids ...
        assert torch.equal(echo.cpu(), table[ids])                  # ... are the reference's gather, bit for bit
        got = tower(texts)                                          # strings -> tokenizer -> .to(device) -> K1 -> K2
        p = {k: v.detach().cpu() for k, v in tower.projection.state_dict().items()}
        ref = orc.tower_forward(table, ids, mask, p["0.weight"], p["0.bias"], p["2.weight"], p["2.bias"])
        assert (got.cpu() - ref).abs().max().item() <= 1e-5, name
    q, d = model(queries, docs)
    assert torch.equal(q, model.encode_queries(queries)) and torch.equal(d, model.encode_documents_batched(docs, batch_size=16))


def test_real_tokenizer_through_the_token_loader_packed_files_and_run_training(tok, tmp_path):
    from two_towers_overlords_amd.data import TokenTripletLoader
    from two_towers_overlords_amd.model import EmbeddingOnlyEncoder
    from two_towers_overlords_amd.training import run_training

    rng = random.Random(2)
    n = 96
    qtext = _texts(24, rng, 1, 8)
    items = [{"query_id": i % 24, "query": qtext[i % 24], "positive": t} for i, t in enumerate(_texts(n, rng, 3, 30))]
    ld = TokenTripletLoader(items, tok, batch_size=32, device="cuda", seed=4, sampler="reference", emit="ids3")
    # tokenised once, rows = the tokenizer's own ids (int32 copies of its int64 output), lengths = its mask sums
    for texts, ids_t, lens_t in (([it["query"] for it in items], ld.q_ids, ld.q_lens), ([it["positive"] for it in items], ld.p_ids, ld.p_lens)):
        enc = tok(texts, padding=True, truncation=True, return_tensors="pt", max_length=512)
        assert torch.equal(ids_t.cpu().long(), enc["input_ids"]) and torch.equal(lens_t.cpu().long(), enc["attention_mask"].sum(1))
    for ids, lens in ld:  # pad="batch": every batch is cut to its own longest row, like padding=True per call
        assert ids.shape[2] == int(lens.max()) and ids.dtype == torch.int32
    # PackedTokens round trip (f3): ids.bin + offsets.bin and back
    prefix = str(tmp_path / "marco")
    ld.save_packed(prefix)
    ld2 = TokenTripletLoader.from_packed(prefix, batch_size=32, device="cuda", seed=4, sampler="reference", emit="ids3")
    assert torch.equal(ld2.q_ids, ld.q_ids) and torch.equal(ld2.p_ids, ld.p_ids) and torch.equal(ld2.p_lens, ld.p_lens)
    assert ld2.query_ids == ld.query_ids
    ld3 = TokenTripletLoader(items, tok, batch_size=32, device="cuda", seed=4, sampler="reference", emit="ids3")
    for a, b in zip(ld3, ld2):  # same seed, same draws (both loaders fresh)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return len(items)

        def __getitem__(self, i):
            return items[i]

    table = orc.synthetic_table(tok.vocab_size, 64, seed=5) * 30
    flat = {}
    for cache in (True, False):
        torch.manual_seed(0)
        m = run_training(num_epochs=2, batch_size=32, learning_rate=1e-2, max_samples=10 ** 6, projection_dim=32, margin=0.3,
                         use_wandb=False, accumulation_steps=1, train_dataset=DS(), tokenizer=tok,
                         pretrained_model=EmbeddingOnlyEncoder(table), seed=5, feature_cache=cache)
        flat[cache] = torch.cat([p.detach().reshape(-1).cpu() for p in m.trainable_parameters()])
        assert torch.isfinite(flat[cache]).all()
    assert (flat[True] - flat[False]).abs().max() <= 2e-5


class _DS(torch.utils.data.Dataset):
    def __init__(self, items):
        self.items = items

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


def _marco_like(n_queries, n, rng):
    qtext = _texts(n_queries, rng, 1, 8)
    return [{"query_id": i % n_queries, "query": qtext[i % n_queries], "positive": t} for i, t in enumerate(_texts(n, rng, 3, 40))]


def test_string_batches_are_tokenised_once_and_train_like_the_reference_loop(tok, monkeypatch):
    """`train_epoch(model, batches of strings, TripletLoss, torch.optim.Adam)` — the reference's literal
    call (backend/training.py:26-63,436) — on the engine with the route's text store: every text goes through the
    tokenizer ONCE (epoch 1), epochs 2 and 3 make no tokenizer call at all, and the three epochs are the module loop's
    (which tokenises three lists per batch like backend/model.py:43-48) — losses, parameters, Adam moments."""
    from two_towers_overlords_amd.model import TripletLoss
    from two_towers_overlords_amd.training import train_epoch

    rng = random.Random(7)
    items = _marco_like(40, 160, rng)
    B = 32
    batches = []
    for s in range(0, len(items), B):
        chunk = items[s: s + B]
        neg = [(i * 7 + 3) % len(chunk) for i in range(len(chunk))]
        batches.append(([c["query"] for c in chunk], [c["positive"] for c in chunk], [chunk[j]["positive"] for j in neg]))
    crit = TripletLoss(margin=0.3)
    calls = {"n": 0, "texts": 0}
    real = type(tok).__call__
    from two_towers_overlords_amd import textstore

    rows_of = textstore.hf_token_rows

    def counting(tok_, texts, *a, **kw):   # every path behind a transformers tokenizer asks for its ids here (list form)
        calls["n"] += 1
        calls["texts"] += len(texts)
        return rows_of(tok_, texts, *a, **kw)

    monkeypatch.setattr(textstore, "hf_token_rows", counting)
    out = {}
    for mode in ("modules", "engine-inline", "engine-cached"):
        model, _ = _model(tok, 48, 3)
        opt = torch.optim.Adam(model.parameters(), lr=1e-2)
        calls.update(n=0, texts=0)
        per_epoch, losses = [], []
        for _ in range(3):
            losses.append(train_epoch(model, batches, crit, opt, log_wandb=False, fused=(False if mode == "modules" else None),
                                      text_cache=(None if mode == "engine-cached" else False)))
            per_epoch.append(calls["texts"])
        assert (getattr(opt, "_tt_route", None) is not None) == (mode != "modules")
        out[mode] = (losses, [p.detach().clone() for p in model.trainable_parameters()],
                     [opt.state[p]["exp_avg"].clone() for p in model.trainable_parameters()], per_epoch)
        if mode == "engine-cached":
            store = opt._tt_route.text_store
            distinct = len({c["query"] for c in items}) + len({c["positive"] for c in items})
            assert per_epoch == [distinct, distinct, distinct]  # nothing tokenised after the first epoch, nothing twice
            st = store.stats()
            assert st["rows"] == distinct == st["texts_tokenized"] and st["width"] % 64 == 0
            # the rows hold the tokenizer's own ids, TT_PAD_ID behind them
            enc = real(tok, [items[5]["positive"]], padding=True, truncation=True, return_tensors="pt", max_length=512)
            row = store._by_text[id(tok)][items[5]["positive"]]
            got = store.ids[row].cpu()
            n = enc["input_ids"].shape[1]
            assert torch.equal(got[:n].long(), enc["input_ids"][0]) and bool((got[n:] == -(2 ** 31)).all())
        else:
            assert per_epoch[2] == 3 * per_epoch[0] == 9 * len(items)  # the reference's three tokenizer calls per batch
    ref = out["modules"]
    for mode in ("engine-inline", "engine-cached"):
        got = out[mode]
        for a, b in zip(ref[0], got[0]):
            assert abs(a - b) <= 1e-5
        for x, y in zip(ref[1], got[1]):
            assert (x - y).abs().max().item() <= 2e-5 * max(1.0, x.abs().max().item())
        for x, y in zip(ref[2], got[2]):
            assert (x - y).abs().max().item() <= 1e-5 * max(1.0, x.abs().max().item())


def test_triplet_dataloader_batches_carry_their_dataset_rows_through_the_text_store(tok):
    """This package's `TripletDataLoader` (backend/data.py:90-152: same shuffle, same draws) yields lists of strings that
    also say which dataset item each string is; the route turns them into store rows with index arithmetic — no per-text
    look-up — and the run is the module loop's under the same seeds.  The store grows in rows and in width while gathers
    are in flight (it starts at 4096 x 64; positives here reach past 64 tokens)."""
    from two_towers_overlords_amd.data import TextColumn, TripletDataLoader
    from two_towers_overlords_amd.model import TripletLoss
    from two_towers_overlords_amd.training import train_epoch

    rng = random.Random(11)
    items = _marco_like(60, 300, rng)
    long_doc = " ".join(rng.choice(WORDS) for _ in range(150))
    items[17]["positive"] = long_doc
    ds = _DS(items)
    crit = TripletLoss(margin=0.3)
    out = {}
    for fused in (False, None):
        model, _ = _model(tok, 32, 5)
        opt = torch.optim.Adam(model.parameters(), lr=1e-2)
        random.seed(21)
        torch.manual_seed(21)
        dl = TripletDataLoader(ds, batch_size=64, num_workers=0)
        q, p, n = next(iter(dl))
        assert isinstance(q, TextColumn) and isinstance(q, list) and q.field == "query" and n.field == "positive"
        assert all(items[r]["query"] == t for r, t in zip(q.rows, q)) and all(items[r]["positive"] == t for r, t in zip(n.rows, n))
        random.seed(22)
        torch.manual_seed(22)
        losses = [train_epoch(model, dl, crit, opt, log_wandb=False, fused=fused) for _ in range(3)]
        out[fused] = (losses, [p_.detach().clone() for p_ in model.trainable_parameters()])
        if fused is None:
            store = opt._tt_route.text_store
            st = store.stats()
            assert st["rows"] == 2 * len(items) == st["texts_tokenized"]  # one row per dataset field, filled once
            assert not store._by_text and st["width"] >= 128              # no dict look-ups; widened for the long passage
    for a, b in zip(out[False][0], out[None][0]):
        assert abs(a - b) <= 1e-5
    for x, y in zip(out[False][1], out[None][1]):
        assert (x - y).abs().max().item() <= 2e-5 * max(1.0, x.abs().max().item())


def test_string_batches_through_accumulation_and_an_engine_rebuild_mid_epoch(tok):
    """`train_epoch_optimized(model, string batches, ..., accumulation_steps=2)` (backend/training.py:66-133) on the engine
    with the text store, over batches whose sizes GROW mid-epoch (16, 32, 40, 8: the engine is rebuilt for the larger ones
    while the store, the optimizer state and the un-applied gradient window carry over) — against the module loop that
    tokenises every batch like the reference; two epochs, so the odd batch's gradient crosses the epoch boundary."""
    from two_towers_overlords_amd.model import TripletLoss
    from two_towers_overlords_amd.training import train_epoch_optimized

    rng = random.Random(13)
    items = _marco_like(30, 136, rng)
    sizes, batches, s = [16, 32, 40, 8, 40], [], 0
    for n in sizes:
        chunk = items[s: s + n]
        s += n
        neg = [(i * 5 + 1) % n for i in range(n)]
        batches.append(([c["query"] for c in chunk], [c["positive"] for c in chunk], [chunk[j]["positive"] for j in neg]))
    crit = TripletLoss(margin=0.3)
    out = {}
    for fused in (False, None):
        model, _ = _model(tok, 40, 9)
        opt = torch.optim.Adam(model.parameters(), lr=5e-3)
        losses = [train_epoch_optimized(model, batches, crit, opt, torch.device("cuda"), None, accumulation_steps=2,
                                        log_wandb=False, fused=fused) for _ in range(2)]
        grads = [None if p.grad is None else p.grad.clone() for p in model.trainable_parameters()]
        out[fused] = (losses, [p.detach().clone() for p in model.trainable_parameters()], grads,
                      {int(opt.state[p]["step"]) for p in model.trainable_parameters()})
        if fused is None:
            route = opt._tt_route
            assert route.trainer.engine.max_B == 40 and route.text_store.stats()["texts_tokenized"] == route.text_store.stats()["rows"]
    assert out[False][3] == out[None][3] == {4}     # two optimizer steps per epoch (batch_idx restarts); epoch 1's fifth gradient is applied with epoch 2's first window
    for a, b in zip(out[False][0], out[None][0]):
        assert abs(a - b) <= 1e-5
    for x, y in zip(out[False][1], out[None][1]):
        assert (x - y).abs().max().item() <= 2e-5 * max(1.0, x.abs().max().item())
    for x, y in zip(out[False][2], out[None][2]):
        assert (x is None) == (y is None)
        if x is not None:
            assert (x - y).abs().max().item() <= 1e-5 * max(1.0, x.abs().max().item())
