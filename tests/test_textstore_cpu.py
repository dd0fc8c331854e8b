"""CPU: the host logic behind string batches of the drop-in route (round 6) — the text store's row bookkeeping and
growth, the list-form tokenizer call against the reference's call, which optimizers the route accepts, and the O(1)
in-batch negative draw against the reference's literal list form."""
import random

import numpy as np
import pytest
import torch

from two_towers_overlords_amd.data import TextColumn, TripletDataLoader, sample_negative_indices
from two_towers_overlords_amd.textstore import PAD32, TextRowStore, hf_token_rows, is_hf_tokenizer, per_text_tokenizer

WORDS = "alpha beta gamma delta epsilon zeta eta theta iota kappa lambda mu nu xi omicron pi rho sigma tau".split()


@pytest.fixture(scope="module")
def tok(tmp_path_factory):
    transformers = pytest.importorskip("transformers")
    d = tmp_path_factory.mktemp("vocab")
    (d / "vocab.txt").write_text("\n".join(["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + WORDS + ["##s", "."]) + "\n")
    return transformers.BertTokenizerFast.from_pretrained(str(d))


def _texts(n, rng, lo, hi):
    return [" ".join(rng.choice(WORDS) for _ in range(rng.randint(lo, hi))) + rng.choice(["", ".", "s"]) for _ in range(n)]


def test_list_form_tokenizer_call_gives_the_reference_calls_ids(tok):
    rng = random.Random(3)
    texts = _texts(57, rng, 1, 40) + [" ".join(["alpha"] * 700)]          # one text beyond max_length: truncated to 512 like the reference
    ref = tok(texts, padding=True, truncation=True, return_tensors="pt", max_length=512)   # backend/model.py:43-45
    rows, lens = hf_token_rows(tok, texts)
    assert rows.dtype == np.int32 and rows.shape == tuple(ref["input_ids"].shape) and rows.shape[1] == 512
    assert np.array_equal(lens, ref["attention_mask"].sum(1).numpy())
    keep = ref["attention_mask"].numpy().astype(bool)
    assert np.array_equal(rows[keep], ref["input_ids"].numpy()[keep]) and (rows[~keep] == PAD32).all()
    assert is_hf_tokenizer(tok) and per_text_tokenizer(tok)
    assert not per_text_tokenizer(lambda texts, **kw: None)

    class Declared:
        tt_per_text = True

    assert per_text_tokenizer(Declared()) and not is_hf_tokenizer(Declared())


def test_text_store_rows_growth_and_both_lookups(tok):
    rng = random.Random(5)
    calls = []

    def tokenize(texts):
        calls.append(len(texts))
        return hf_token_rows(tok, texts)

    store = TextRowStore("cpu", max_S=512, rows=8, width=64)
    a = _texts(20, rng, 1, 10)
    ra = store.rows_of_texts("k", a + a[:5], tokenize)                    # duplicates inside one call: one row each
    assert calls == [len(set(a))] and list(ra[:5]) == list(ra[20:]) and store.ids.shape[0] >= len(set(a))
    assert np.array_equal(store.rows_of_texts("k", a, tokenize), ra[:20]) and calls == [len(set(a))]    # second time: no tokenizer
    long_text = " ".join(["beta"] * 100)
    rl = store.rows_of_texts("k", [long_text], tokenize)
    assert store.ids.shape[1] == 128                                       # widened in steps of 64, old rows intact
    for t, r in zip(a, ra):
        ref = tok([t], padding=True, truncation=True, return_tensors="pt", max_length=512)["input_ids"][0]
        got = store.ids[r]
        assert torch.equal(got[: len(ref)].long(), ref) and bool((got[len(ref):] == PAD32).all())
    assert int((store.ids[rl[0]] != PAD32).sum()) == 102
    with pytest.raises(ValueError):
        TextRowStore("cpu", max_S=64).rows_of_texts("k", [long_text], tokenize)

    # TextColumn lists: row = arithmetic on the dataset index; queries and positives of one dataset do not collide
    items = [{"query_id": i % 7, "query": q, "positive": p} for i, (q, p) in enumerate(zip(_texts(30, rng, 1, 6), _texts(30, rng, 5, 30)))]
    calls.clear()
    idx = np.array([3, 9, 3, 29, 0], dtype=np.int64)
    q = TextColumn([items[i]["query"] for i in idx], items, "query", idx)
    p = TextColumn([items[i]["positive"] for i in idx], items, "positive", idx)
    rq, rp = store.rows_of_column(q, tokenize), store.rows_of_column(p, tokenize)
    assert calls == [4, 4] and rq[0] == rq[2] and len(set(rq) & set(rp)) == 0 and (rp - rq == len(items)).all()
    assert np.array_equal(store.rows_of_column(q, tokenize), rq) and calls == [4, 4]
    for col, rows in ((q, rq), (p, rp)):
        for t, r in zip(col, rows):
            n = len(tok([t])["input_ids"][0])
            assert int((store.ids[r] != PAD32).sum()) == n
    with pytest.raises(ValueError):
        store.rows_of_column(TextColumn(["x"], items, "query", np.array([30])), tokenize)      # not an index of its dataset
    st = store.stats()
    assert st["rows"] == len(set(a)) + 1 + 8 and st["texts_tokenized"] == st["rows"]


def test_triplet_dataloader_columns_name_their_dataset_rows():
    rng = random.Random(9)
    items = [{"query_id": i // 3, "query": f"q{i // 3}", "positive": f"passage {i}"} for i in range(50)]

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return len(items)

        def __getitem__(self, i):
            return items[i]

    ds = DS()
    random.seed(1)
    torch.manual_seed(1)
    seen = 0
    for q, p, n in TripletDataLoader(ds, batch_size=16, num_workers=0):
        assert type(q) is TextColumn and isinstance(q, list) and q.source is ds and (q.field, p.field, n.field) == ("query", "positive", "positive")
        assert [items[r]["query"] for r in q.rows] == list(q) and [items[r]["positive"] for r in p.rows] == list(p)
        assert [items[r]["positive"] for r in n.rows] == list(n) and np.array_equal(q.rows, p.rows)
        assert all(items[a]["query_id"] != items[b]["query_id"] for a, b in zip(q.rows, n.rows))     # backend/data.py:124-128
        seen += len(q)
    assert seen == len(items)


def test_constant_time_negative_draw_is_the_references_list_draw():
    """`rng.choice([k for k in range(n) if k != i])` (backend/data.py:132-137) and `r = rng._randbelow(n - 1); r + (r >= i)`
    consume the generator identically: the same draws for any seed, with rejections (equal query ids, empty positives)."""
    def literal(query_ids, rng, ok=None):
        out = []
        for i in range(len(query_ids)):
            while True:
                j = rng.choice([k for k in range(len(query_ids)) if k != i])
                if (ok is None or ok[j]) and query_ids[j] != query_ids[i]:
                    out.append(j)
                    break
        return out

    for seed in range(5):
        g = random.Random(100 + seed)
        qids = [g.randrange(6) for _ in range(40)]
        ok = [g.random() > 0.2 for _ in range(40)]
        assert sample_negative_indices(qids, random.Random(seed), ok) == literal(qids, random.Random(seed), ok)
        random.seed(seed)
        a = sample_negative_indices(qids)                                   # the module-level generator, like the reference
        random.seed(seed)
        assert a == literal(qids, random)
    with pytest.raises(IndexError):
        sample_negative_indices([1])


def test_which_optimizers_the_engine_route_takes():
    from two_towers_overlords_amd.training import FusedAdam, _adam_form, _step_value

    w = [torch.nn.Parameter(torch.zeros(3))]
    assert _adam_form(torch.optim.Adam(w, lr=1e-3)) == "torch"                # backend/training.py:436
    assert _adam_form(torch.optim.Adam(w, lr=1e-3, betas=(0.8, 0.9), eps=1e-6)) == "torch"
    assert _adam_form(FusedAdam(w, lr=1e-3)) == "fused"
    for kw in ({"weight_decay": 0.1}, {"amsgrad": True}, {"maximize": True}, {"capturable": True}, {"differentiable": True}):
        assert _adam_form(torch.optim.Adam(w, lr=1e-3, **kw)) is None, kw
    assert _adam_form(torch.optim.AdamW(w, lr=1e-3)) is None and _adam_form(torch.optim.SGD(w, lr=1e-3)) is None
    assert _adam_form(torch.optim.Adam([{"params": w}, {"params": [torch.nn.Parameter(torch.zeros(2))]}], lr=1e-3)) is None
    s = _step_value("torch", 7)
    assert isinstance(s, torch.Tensor) and s.dim() == 0 and s.device.type == "cpu" and float(s) == 7.0
    assert _step_value("fused", 7) == 7 and isinstance(_step_value("fused", 7), int)
    # what torch itself keeps there (so that a state_dict written by the route is a stock Adam's)
    opt = torch.optim.Adam(w, lr=1e-3)
    w[0].grad = torch.ones(3)
    opt.step()
    native = opt.state[w[0]]["step"]
    assert type(native) is type(_step_value("torch", 1)) and native.dtype == _step_value("torch", 1).dtype and native.device == s.device


def test_tower_tokenize_rebuilds_the_reference_calls_tensors(tok):
    """`AveragePoolingTower.tokenize` behind a right-padding transformers tokenizer: the same keys, dtypes and values as
    `tokenizer(texts, padding=True, truncation=True, return_tensors="pt", max_length=512)` (backend/model.py:43-45)."""
    from two_towers_overlords_amd.model import AveragePoolingTower, EmbeddingOnlyEncoder

    rng = random.Random(4)
    texts = _texts(33, rng, 1, 30) + [" ".join(["mu"] * 600)]
    tower = AveragePoolingTower(projection_dim=8, tokenizer=tok, pretrained_model=EmbeddingOnlyEncoder(torch.zeros(tok.vocab_size, 16)),
                                encoder_mode="embed")
    got = tower.tokenize(texts)
    ref = tok(texts, padding=True, truncation=True, return_tensors="pt", max_length=512)
    assert set(got) == set(ref.keys())
    for k in ref.keys():
        assert got[k].dtype == ref[k].dtype and torch.equal(got[k], ref[k]), k
    tok.padding_side = "left"                     # a left-padding tokenizer keeps the reference's literal call
    try:
        left = tower.tokenize(texts[:5])
        ref = tok(texts[:5], padding=True, truncation=True, return_tensors="pt", max_length=512)
        assert all(torch.equal(left[k], ref[k]) for k in ref.keys()) and int(left["attention_mask"][0, 0]) in (0, 1)
    finally:
        tok.padding_side = "right"


def test_backend_tokenizer_path_is_self_checked_and_falls_back(tmp_path):
    """`textstore._backend_ids` calls the Rust backend directly (transformers' per-text Python costs more than the tokenizer),
    after checking ONCE per tokenizer object that the backend's ids are the public call's; a backend that disagrees, or
    raises, is never used again and the public call's ids are returned."""
    transformers = pytest.importorskip("transformers")
    (tmp_path / "vocab.txt").write_text("\n".join(["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + WORDS) + "\n")
    rng = random.Random(8)
    texts = _texts(40, rng, 1, 12)

    def fresh():
        return transformers.BertTokenizerFast.from_pretrained(str(tmp_path))

    good = fresh()
    want = good(texts, padding=True, truncation=True, return_tensors="pt", max_length=512)
    rows, lens = hf_token_rows(good, texts)
    assert good._tt_backend_ids is True
    keep = want["attention_mask"].numpy().astype(bool)
    assert np.array_equal(rows[keep], want["input_ids"].numpy()[keep]) and np.array_equal(lens, keep.sum(1))

    class Proxy:
        def __init__(self, real, mode):
            self.real, self.mode = real, mode

        def encode_batch(self, batch, **kw):
            if self.mode == "raise":
                raise RuntimeError("backend unavailable")
            out = self.real.encode_batch(batch, **kw)

            class E:
                def __init__(self, ids):
                    self.ids = ids

            return [E(list(e.ids[:-1]) + [1]) for e in out]   # last id wrong

    for mode in ("wrong", "raise"):
        base = fresh()

        class Broken(type(base)):
            @property
            def backend_tokenizer(self):
                return Proxy(super().backend_tokenizer, mode)

        base.__class__ = Broken
        r2, l2 = hf_token_rows(base, texts)
        assert base._tt_backend_ids is False and np.array_equal(r2, rows) and np.array_equal(l2, lens)
        r3, _ = hf_token_rows(base, texts[:5])      # and stays on the public call
        assert np.array_equal(r3[:, : rows.shape[1]][r3[:, : rows.shape[1]] != PAD32], rows[:5][rows[:5] != PAD32])
