"""Host-side triplet batching (mirror of backend/data.py:90-152) against the reference's own draws."""
import random

import torch

from conftest import load_golden
from two_towers_overlords_amd.data import TokenTripletLoader, TripletDataLoader, sample_negative_indices


def test_negative_sampling_matches_reference_draws():
    g = load_golden("triplets.npz")
    for case in range(6):  # cases 4-5: some passages are empty strings (backend/data.py:126 never draws those)
        qids = g[f"case{case}/query_ids"].tolist()
        random.seed(int(g[f"case{case}/seed"]))
        ok = g[f"case{case}/positive_nonempty"].tolist()
        assert sample_negative_indices(qids, positive_nonempty=ok) == g[f"case{case}/neg_idx"].tolist()
        if all(ok):
            random.seed(int(g[f"case{case}/seed"]))
            assert sample_negative_indices(qids) == g[f"case{case}/neg_idx"].tolist()


class _DS(torch.utils.data.Dataset):
    def __init__(self, n, nq):
        r = random.Random(0)
        self.items = [{"query_id": r.randrange(nq), "query": "", "positive": ""} for _ in range(n)]
        for i, it in enumerate(self.items):
            it["query"] = " ".join(["q%d" % it["query_id"]] * (1 + it["query_id"] % 3))
            it["positive"] = " ".join(["d%d" % i] * (2 + i % 5))

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


def _tok(texts, padding=True, truncation=True, return_tensors="pt", max_length=512):
    rows = [[1 + (hash(w) % 97) for w in t.split()][:max_length] for t in texts]
    S = max(len(r) for r in rows)
    ids = torch.zeros(len(rows), S, dtype=torch.int64)
    mask = torch.zeros(len(rows), S, dtype=torch.int64)
    for i, r in enumerate(rows):
        ids[i, : len(r)] = torch.tensor(r)
        mask[i, : len(r)] = 1
    return {"input_ids": ids, "attention_mask": mask}


def test_string_loader_semantics():
    ds = _DS(23, 6)
    dl = TripletDataLoader(ds, batch_size=8, num_workers=0)
    seen = 0
    for q, p, n in dl:
        assert len(q) == len(p) == len(n)
        for qi, pi, ni in zip(q, p, n):
            # the negative is some in-batch item's positive, from a different query
            j = p.index(ni)
            assert q[j] != qi and ni != pi
        seen += len(q)
    assert seen == 23 and len(dl) == 3  # drop_last is not set (data.py:102-111)


def test_token_loader_negatives_are_in_batch_positives():
    ds = _DS(23, 6)
    tl = TokenTripletLoader(ds, _tok, batch_size=8, device="cpu", seed=5)
    total = 0
    for ids, lens in tl:
        B = ids.shape[1]
        total += B
        assert ids.dtype == torch.int32 and lens.shape == (3, B)
        S = ids.shape[2]
        assert int(lens.max()) == S  # padded to the longest row of the batch
        for i in range(B):
            nrow = ids[2, i, : lens[2, i]]
            matches = [j for j in range(B) if lens[1, j] == lens[2, i] and torch.equal(ids[1, j, : lens[1, j]], nrow)]
            assert matches and i not in matches
            # positions past the length are padding zeros
            assert int(ids[2, i, lens[2, i]:].abs().sum()) == 0
    assert total == 23


def test_token_loader_emit_modes_describe_the_same_triplets():
    """`index` / `cache` batches carry the same triplets as `ids3` under the same seed."""
    ds = _DS(23, 6)
    a = list(TokenTripletLoader(ds, _tok, batch_size=8, device="cpu", seed=5, emit="ids3"))
    b = list(TokenTripletLoader(ds, _tok, batch_size=8, device="cpu", seed=5, emit="index"))
    ld = TokenTripletLoader(ds, _tok, batch_size=8, device="cpu", seed=5, emit="cache")
    c = list(ld)
    all_ids, all_lens = ld.all_texts()
    assert all_ids.shape[0] == 2 * 23
    for (ids3, lens3), bi, bc in zip(a, b, c):
        e = bi.expand_negatives()
        S = e.ids.shape[2]
        assert torch.equal(e.lens, lens3)
        assert torch.equal(e.ids, ids3[:, :, :S]) and int(ids3[:, :, S:].abs().sum()) == 0
        assert torch.equal(bc.neg_index, bi.neg_index)
        ec = bc.expand_negatives()
        for r in range(3):  # cache rows point at the same token rows
            rows = ec.cache_rows[r].long()
            assert torch.equal(all_lens[rows], lens3[r])
            for i in range(rows.numel()):
                L = int(lens3[r, i])
                assert torch.equal(all_ids[rows[i], :L], ids3[r, i, :L])


def test_device_sampler_distribution():
    from two_towers_overlords_amd.data import sample_negative_indices_device

    g = torch.Generator().manual_seed(1)
    qids = torch.tensor([0, 0, 1, 2, 2, 2, 3, 0])
    counts = torch.zeros(8, 8)
    for _ in range(4000):
        neg = sample_negative_indices_device(qids, g)
        assert torch.all(qids[neg] != qids)           # never the same query (hence never itself)
        counts[torch.arange(8), neg] += 1
    # uniform over the admissible items of each row (backend/data.py:124-128's rejection loop)
    for i in range(8):
        adm = (qids != qids[i])
        p = counts[i][adm] / counts[i].sum()
        assert counts[i][~adm].sum() == 0 and float((p - 1.0 / adm.sum()).abs().max()) < 0.03
    # degenerate batches: the reference raises IndexError for B = 1 and never returns when all ids are equal
    import pytest
    with pytest.raises(IndexError):
        sample_negative_indices_device(torch.tensor([5]), g)
    with pytest.raises(RuntimeError):
        sample_negative_indices_device(torch.tensor([5, 5, 5]), g)


def test_packed_token_files_round_trip(tmp_path):
    from two_towers_overlords_amd.data import PackedTokens

    ds = _DS(23, 6)
    a = TokenTripletLoader(ds, _tok, batch_size=8, device="cpu", seed=5, emit="index")
    prefix = str(tmp_path / "marco")
    a.save_packed(prefix)
    b = TokenTripletLoader.from_packed(prefix, batch_size=8, device="cpu", seed=5, emit="index")
    assert b.query_ids == a.query_ids and torch.equal(b.q_lens, a.q_lens) and torch.equal(b.p_lens, a.p_lens)
    assert torch.equal(b.q_ids, a.q_ids[:, : b.q_ids.shape[1]]) and torch.equal(b.p_ids, a.p_ids[:, : b.p_ids.shape[1]])
    for x, y in zip(a, b):  # same seed -> the same batches from the packed files
        assert torch.equal(x.ids, y.ids) and torch.equal(x.lens, y.lens) and torch.equal(x.neg_index, y.neg_index)
    pk = PackedTokens.load(prefix + ".p")
    assert len(pk) == 23 and pk.max_len == int(a.p_lens.max())
    # device sampler: same triplet structure, different draws
    c = TokenTripletLoader.from_packed(prefix, batch_size=8, device="cpu", seed=5, emit="index", sampler="device")
    for tb in c:
        qid = torch.tensor(c.query_ids)
        assert tb.neg_index.dtype == torch.int32 and tb.ids.shape[0] == 2


def test_device_sampler_loader_needs_no_per_batch_shapes_and_keeps_triplet_structure():
    """sampler="device" (default pad="max"): every batch is padded to the collection's longest row (so nothing
    about a batch has to be read back to size it), negatives are in-batch positives of a different query,
    the three emit modes describe the same triplets, and a batch without two distinct queries raises on the
    host (the reference would spin forever in create_triplets, backend/data.py:124-128)."""
    import pytest

    ds = _DS(23, 6)
    mk = lambda emit: TokenTripletLoader(ds, _tok, batch_size=8, device="cpu", seed=5, emit=emit, sampler="device")
    a, b = list(mk("ids3")), list(mk("index"))
    ld = mk("cache")
    c = list(ld)
    assert ld.pad == "max"
    r3, r2 = list(mk("rows3")), list(mk("rows2"))
    all_ids, all_lens = ld.all_texts()
    for (ids3, lens3), x3, x2, bc in zip(a, r3, r2, c):  # row forms name the same texts (store = all_texts())
        assert x3.token_rows.dtype == torch.int32 and x3.token_rows.shape == (3, ids3.shape[1]) and x3.neg_index is None
        assert torch.equal(all_ids[x3.token_rows.long()], ids3) and torch.equal(all_lens[x3.token_rows.long()], lens3)
        assert torch.equal(x2.token_rows, bc.cache_rows) and torch.equal(x2.neg_index, bc.neg_index)
        assert torch.equal(x2.expand_negatives().token_rows, x3.token_rows)
    Smax = max(ld.q_ids.shape[1], ld.p_ids.shape[1])
    qid = torch.tensor(ld.query_ids)
    total = 0
    for (ids3, lens3), bi, bc in zip(a, b, c):
        B = ids3.shape[1]
        total += B
        assert ids3.shape == (3, B, Smax) and bi.ids.shape == (2, B, Smax)
        e = bi.expand_negatives()
        assert torch.equal(e.ids, ids3) and torch.equal(e.lens, lens3) and torch.equal(bc.neg_index, bi.neg_index)
        items = bc.cache_rows[0].long()                       # dataset indices of the batch
        assert bool((qid[items][bi.neg_index.long()] != qid[items]).all())
        assert int((ids3 * (torch.arange(Smax).view(1, 1, -1) >= lens3.unsqueeze(-1))).abs().sum()) == 0
    assert total == 23

    class _One(_DS):
        def __init__(self):
            super().__init__(6, 1)

    with pytest.raises(RuntimeError):
        list(TokenTripletLoader(_One(), _tok, batch_size=4, device="cpu", seed=1, sampler="device"))


def test_loader_shards_are_the_columns_of_the_global_batches():
    """shard=(rank, world): every rank sees the same global batches (same seed) and gets its contiguous columns of
    each, negatives as a third row; the ragged tail batch may leave a rank with nothing."""
    import pytest

    from two_towers_overlords_amd.training import shard_sizes

    ds = _DS(26, 7)  # batches of 8, 8, 8, 2
    for sampler in ("device", "reference"):
        for emit in ("rows2", "cache"):
            glob = list(TokenTripletLoader(ds, _tok, batch_size=8, device="cpu", seed=5, emit=emit, sampler=sampler))
            parts = [list(TokenTripletLoader(ds, _tok, batch_size=8, device="cpu", seed=5, emit=emit, sampler=sampler,
                                             shard=(r, 3))) for r in range(3)]
            sizes = TokenTripletLoader(ds, _tok, batch_size=8, device="cpu", seed=5, emit=emit, sampler=sampler,
                                       shard=(0, 3)).global_batch_sizes()
            assert sizes == [8, 8, 8, 2]
            for bi, g in enumerate(glob):
                full = g.expand_negatives()
                rows = full.cache_rows if emit == "cache" else full.token_rows
                got = [p[bi].cache_rows if emit == "cache" else p[bi].token_rows for p in parts]
                assert all(x.shape[0] == 3 and p[bi].neg_index is None for x, p in zip(got, parts))
                assert [x.shape[1] for x in got] == shard_sizes(sizes[bi], 3)[0]
                assert torch.equal(torch.cat(got, dim=1), rows)
    assert parts[2][3].size == 0  # 2 items over 3 ranks
    with pytest.raises(ValueError):
        TokenTripletLoader(ds, _tok, batch_size=8, device="cpu", seed=5, emit="ids3", shard=(0, 2))


def test_empty_passages_are_never_negatives():
    """backend/data.py:126 (`if negative and ...`): an item whose passage is the empty string is never the negative —
    string loader, reference sampler and device sampler alike; the device sampler stays uniform over the rest."""
    from two_towers_overlords_amd.data import sample_negative_indices_device

    ds = _DS(40, 7)
    empties = {3, 8, 21, 22, 37}
    for i in empties:
        ds.items[i]["positive"] = ""
    dl = TripletDataLoader(ds, batch_size=16, num_workers=0)
    for q, p, n in dl:
        assert all(n)
    for sampler in ("reference", "device"):
        tl = TokenTripletLoader(ds, lambda t, **k: _tok([x or "pad" for x in t], **k), batch_size=16, device="cpu", seed=3,
                                sampler=sampler, emit="index")
        assert tl.p_nonempty is not None and sum(not x for x in tl.p_nonempty) == len(empties)
        empty_rows = {tuple(tl.p_ids[i].tolist()) for i in empties}
        for _ in range(3):
            for b in tl:
                negs = b.ids[1][b.neg_index.long()]
                assert not any(tuple(r.tolist()) in empty_rows for r in negs)
    q = torch.tensor([0, 0, 1, 1, 2, 2, 3, 3])
    ok = torch.tensor([1, 0, 1, 1, 0, 1, 1, 0]).bool()
    g = torch.Generator().manual_seed(0)
    counts = torch.zeros(8, 8)
    for _ in range(3000):
        n = sample_negative_indices_device(q, g, positive_nonempty=ok)
        counts[torch.arange(8), n] += 1
    for i in range(8):
        adm = [j for j in range(8) if ok[j] and q[j] != q[i]]
        assert counts[i].sum() == counts[i, adm].sum()
        assert (counts[i, adm] / 3000 - 1 / len(adm)).abs().max() < 0.04


def test_peek_does_not_advance_the_loaders_random_streams():
    """run_training calibrates on loader.peek(): epochs after it must be the ones a fresh loader yields."""
    ds = _DS(37, 6)
    for sampler, emit in (("reference", "ids3"), ("device", "rows3"), ("device", "index")):
        a = TokenTripletLoader(ds, _tok, batch_size=8, device="cpu", seed=9, sampler=sampler, emit=emit)
        b = TokenTripletLoader(ds, _tok, batch_size=8, device="cpu", seed=9, sampler=sampler, emit=emit)
        first = a.peek()
        flat = lambda x: [t for t in (x if isinstance(x, tuple) else (x.ids, x.lens, x.neg_index, x.token_rows)) if t is not None]
        for _ in range(2):
            ea, eb = list(a), list(b)
            assert len(ea) == len(eb)
            for x, y in zip(ea, eb):
                assert all(torch.equal(u, v) for u, v in zip(flat(x), flat(y)))
        assert all(torch.equal(u, v) for u, v in zip(flat(first), flat(next(iter(TokenTripletLoader(
            ds, _tok, batch_size=8, device="cpu", seed=9, sampler=sampler, emit=emit))))))
