"""Host-side mirror of the triplet batching of backend/data.py:90-152, plus the
token-level form the device path consumes.

`TripletDataLoader` keeps the reference's semantics exactly: shuffle, batch, and
for every item draw an in-batch negative uniformly from the *other* items
(`random.choice([j for j != i])`, rejected while the query_id is equal); the
negative text is that item's *positive*.  Because negatives are in-batch
positives, the token-level loader tokenises every query and
positive ONCE and forms each batch's negatives by an index gather of the
positives' token rows — no third tokenisation, identical ids.

MSMarcoDataset (backend/data.py:13-87) is a network fetch + string wrangling
and is out of scope; any map-style dataset of {"query_id","query","positive"}
works here, as it does for the reference's loader.
"""
from __future__ import annotations

import random
from typing import Callable, Iterator, List, Optional, Sequence, Tuple

import torch
from torch.utils.data import DataLoader

Triplet = Tuple[List[str], List[str], List[str]]


def sample_negative_indices(query_ids: Sequence[int], rng=random,
                            positive_nonempty: Optional[Sequence[bool]] = None) -> List[int]:
    """backend/data.py:113-137 on indices: neg_idx[i] = j, drawn like
    `_get_random_negative` (same `choice` call sequence, so the same seed gives the
    same draws as the reference) and rejected while query_ids[j] == query_ids[i] or
    item j's positive is an empty string (data.py:126, `if negative and ...`;
    `positive_nonempty[j]` False).  Like the reference this never terminates if no
    other item qualifies and raises IndexError for a batch of one."""
    n = len(query_ids)
    out = []
    # `rng.choice([k for k in range(n) if k != i])` is `seq[rng._randbelow(n - 1)]` (CPython's Random.choice) over a list
    # whose r-th entry is r + (r >= i): the same draw from the same generator state without building the list — O(B) per
    # batch where the reference's loop is O(B^2) (0.4 s of Python per 2048-item batch).  A generator without
    # `_randbelow` (not a random.Random) takes the literal form.
    below = getattr(getattr(rng, "_inst", rng), "_randbelow", None) if n > 1 else None  # the `random` module draws from `_inst`
    for i in range(n):
        while True:
            if below is not None:
                r = below(n - 1)
                j = r + (r >= i)
            else:
                j = rng.choice([k for k in range(n) if k != i])
            if (positive_nonempty is None or positive_nonempty[j]) and query_ids[j] != query_ids[i]:
                out.append(j)
                break
    return out


def sample_negative_indices_device(query_ids: torch.Tensor, generator: Optional[torch.Generator] = None,
                                   check: bool = True, positive_nonempty: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Same distribution as `sample_negative_indices` (uniform over the in-batch items of a
    different query whose positive is not empty — what backend/data.py:113-137's rejection loop
    converges to), drawn for the whole batch at once on `query_ids`' device, exactly and without
    rejection: sort the batch by query id with the empty-positive items moved to the end, so the
    admissible items are the first n_ok sorted positions minus ONE contiguous run [a_i, b_i) (the
    non-empty items of item i's own query); draw u uniformly from the n_ok - (b_i - a_i)
    admissible positions and skip over the run.
    `query_ids` is [B] (one batch) or [n_batches, B] (independent batches, e.g. a whole epoch in
    one call), `positive_nonempty` a bool tensor of the same shape (None: all non-empty); the
    result has the same shape and holds in-batch indices.
    O(B log B) tensor work against the reference's O(B^2) Python list building per batch, and no
    host synchronisation when `check=False` (the caller has then verified on the host that every
    item of every batch has an admissible negative).  Not the same *draws* as the reference under
    a shared seed.  With `check=True` raises like the reference fails: IndexError for a batch of
    one (`random.choice([])`), RuntimeError when some item has no admissible negative (the
    reference never returns)."""
    B = query_ids.shape[-1]
    if B < 2:
        raise IndexError("cannot draw an in-batch negative from a batch of one")  # reference: random.choice([])
    q = query_ids.reshape(-1, B)
    perm, run_lo, run, admissible = _admissible_runs(q, positive_nonempty)
    if check and int(admissible.min()) <= 0:
        raise RuntimeError("an item of the batch has no admissible negative (every other item shares its query_id "
                           "or has an empty positive)")
    u = torch.rand(q.shape, generator=generator, device=q.device, dtype=torch.float64)
    pick = torch.minimum((u * admissible).long(), (admissible - 1).clamp_(min=0))
    pick = torch.where(pick < run_lo, pick, pick + run)
    return perm.gather(1, pick.clamp_(max=B - 1)).reshape(query_ids.shape)


def _admissible_runs(q: torch.Tensor, positive_nonempty: Optional[torch.Tensor]):
    """For batches q [n, B] of query ids: the sort permutation with empty-positive items moved to the end, and per item
    the run [run_lo, run_lo + run) of sorted positions it must NOT draw (non-empty items of its own query) and the number
    of admissible positions among the n_ok non-empty ones.  O(B log B) per batch, no [B, B] intermediate."""
    B = q.shape[1]
    if positive_nonempty is None:
        key, n_ok = q, B
    else:
        ok = positive_nonempty.reshape(-1, B).to(torch.bool)
        key = torch.where(ok, q, torch.full_like(q, torch.iinfo(q.dtype).max))  # empty positives sort behind every real id
        n_ok = ok.sum(1, keepdim=True)
    sorted_k, perm = torch.sort(key, dim=1)
    run_lo = torch.searchsorted(sorted_k, q, right=False)
    run = torch.searchsorted(sorted_k, q, right=True) - run_lo
    return perm, run_lo, run, n_ok - run


class PackedTokens:
    """On-disk form of a tokenised text collection: `<prefix>.ids.bin` — all
    token ids back to back, int32; `<prefix>.offsets.bin` — int64 [N+1] start offsets.  Written
    once (tokenisation is the expensive host step), memory-mapped afterwards."""

    def __init__(self, ids, offsets):
        self.ids, self.offsets = ids, offsets

    def __len__(self) -> int:
        return len(self.offsets) - 1

    @property
    def max_len(self) -> int:
        import numpy as np

        return int(np.diff(self.offsets).max()) if len(self) else 0

    @staticmethod
    def write(prefix: str, input_ids: torch.Tensor, lens: torch.Tensor) -> "PackedTokens":
        import numpy as np

        ids = input_ids.cpu().numpy()
        ln = lens.cpu().numpy().astype(np.int64)
        offsets = np.zeros(len(ln) + 1, dtype=np.int64)
        np.cumsum(ln, out=offsets[1:])
        flat = np.empty(int(offsets[-1]), dtype=np.int32)
        for i, n in enumerate(ln):
            flat[offsets[i]: offsets[i + 1]] = ids[i, :n]
        flat.tofile(prefix + ".ids.bin")
        offsets.tofile(prefix + ".offsets.bin")
        return PackedTokens.load(prefix)

    @staticmethod
    def load(prefix: str) -> "PackedTokens":
        import numpy as np

        offsets = np.fromfile(prefix + ".offsets.bin", dtype=np.int64)
        ids = np.memmap(prefix + ".ids.bin", dtype=np.int32, mode="r") if offsets[-1] > 0 else np.zeros(0, np.int32)
        return PackedTokens(ids, offsets)

    def padded(self, device) -> Tuple[torch.Tensor, torch.Tensor]:
        """(ids [N, max_len] int32 zero-padded, lens [N] int32) on `device`."""
        import numpy as np

        ln = np.diff(self.offsets)
        out = np.zeros((len(self), max(self.max_len, 1)), dtype=np.int32)
        for i, n in enumerate(ln):
            out[i, :n] = self.ids[self.offsets[i]: self.offsets[i + 1]]
        return torch.from_numpy(out).to(device), torch.from_numpy(ln.astype(np.int32)).to(device)


class TextColumn(list):
    """A `list[str]` that also says where its strings come from: text i is field `field` ("query" | "positive") of item
    `rows[i]` (an int64 numpy array) of the map-style dataset `source`.  To every consumer of backend/data.py:139-152's batches it IS the list
    of strings; the drop-in `train_epoch` route reads the rows and skips both the tokenizer and a per-text look-up for
    texts it has stored before (textstore.py)."""

    __slots__ = ("source", "field", "rows")

    def __init__(self, texts, source, field: str, rows):
        super().__init__(texts)
        self.source, self.field, self.rows = source, field, rows


class _WithRow(torch.utils.data.Dataset):
    """dataset[i] plus the index i itself (as "_row"): default collation turns it into a tensor beside query_id."""

    def __init__(self, dataset):
        self.dataset = dataset

    def __len__(self) -> int:
        return len(self.dataset)

    def __getitem__(self, i):
        item = dict(self.dataset[i])
        item["_row"] = int(i)
        return item


class TripletDataLoader:
    """Creates (queries, positives, negatives) string batches.  Mirrors backend/data.py:90-152.  The three lists are
    `TextColumn`s: plain lists of strings that also carry the dataset index behind every string."""

    def __init__(self, dataset, batch_size: int = 1024, num_workers: int = 4, device: Optional[torch.device] = None):
        self.dataset = dataset
        self.batch_size = batch_size
        self.dataloader = DataLoader(
            _WithRow(dataset),
            batch_size=batch_size,
            shuffle=True,
            num_workers=num_workers,
            pin_memory=False,  # the items are Python strings; nothing to pin (data.py:106-108 pins for cuda)
            persistent_workers=True if num_workers > 0 else False,
        )

    def create_triplets(self, batch: list) -> Triplet:
        queries = [item["query"] for item in batch]
        positives = [item["positive"] for item in batch]
        neg_idx = sample_negative_indices([int(item["query_id"]) for item in batch],
                                          positive_nonempty=[bool(item["positive"]) for item in batch])
        negatives = [batch[j]["positive"] for j in neg_idx]
        if all("_row" in item for item in batch):
            import numpy as np

            rows = np.fromiter((item["_row"] for item in batch), dtype=np.int64, count=len(batch))
            return (TextColumn(queries, self.dataset, "query", rows), TextColumn(positives, self.dataset, "positive", rows),
                    TextColumn(negatives, self.dataset, "positive", rows[np.asarray(neg_idx, dtype=np.int64)]))
        return queries, positives, negatives

    def __iter__(self) -> Iterator[Triplet]:
        for batch in self.dataloader:
            rows = batch["_row"].tolist()
            qid = batch["query_id"]
            qid = qid.tolist() if isinstance(qid, torch.Tensor) and not qid.is_floating_point() else qid  # int(0-d tensor) per item is slow
            batch_list = [
                {"query": batch["query"][i], "positive": batch["positive"][i], "query_id": qid[i], "_row": rows[i]}
                for i in range(len(batch["query"]))
            ]
            yield self.create_triplets(batch_list)

    def __len__(self) -> int:
        return len(self.dataloader)


class _Rows:
    """Padded token rows handed to TokenTripletLoader.from_tensors."""

    def __init__(self, q_ids, q_lens, p_ids, p_lens, query_ids):
        self.q_ids, self.q_lens, self.p_ids, self.p_lens = q_ids, q_lens, p_ids, p_lens
        self.query_ids = query_ids.tolist() if hasattr(query_ids, "tolist") else list(query_ids)


class TokenTripletLoader:
    """Device-side triplet batches for the fused engine: yields
    (ids [3,B,S] int32, lens [3,B] int32) with rows ordered (query, positive,
    negative) like backend/training.py:40-42.

    Every query and positive is tokenised once up front (`tokenizer(texts,
    padding=True, truncation=True, max_length=512)` like model.py:43-45, any
    callable returning `input_ids` + `attention_mask`), packed as int32 rows +
    lengths, and kept on `device`.  Per batch only an index permutation and the
    in-batch negative indices are produced on the host (reference semantics:
    shuffle=True, drop_last=False, negatives via `sample_negative_indices`).
    """

    def __init__(self, dataset, tokenizer: Optional[Callable], batch_size: int, device, max_length: int = 512,
                 seed: Optional[int] = None, shuffle: bool = True, emit: str = "ids3", sampler: str = "reference",
                 packed: Optional[Tuple["PackedTokens", "PackedTokens", Sequence[int]]] = None,
                 pad: Optional[str] = None, shard: Optional[Tuple[int, int]] = None):
        """sampler: "reference" — `sample_negative_indices`, the reference's draws under the same seed
                    (O(B^2) Python per batch, like backend/data.py:132-137); "device" — the same
                    distribution drawn with tensor ops (`sample_negative_indices_device`), no host
                    synchronisation per batch.
        packed: (queries, positives, query_ids) as PackedTokens instead of tokenising `dataset`.
        emit: "ids3"  -> (ids [3,B,S], lens [3,B])                      three gathers per batch
                 "index" -> TripletBatch(ids [2,B,S], lens, neg_index)     negatives by in-batch index
                 "cache" -> TripletBatch(cache_rows [2,B], neg_index)      rows of a frozen-feature cache
                            laid out [queries (N); positives (N)] — build it with
                            `trainer.build_feature_cache(*loader.all_texts())`
                 "rows3" -> TripletBatch(token_rows [3,B])                 three gathers per batch, the ids read
                            through a row indirection into `trainer.set_token_store(*loader.all_texts())`
                            (same layout as the cache: query i -> row i, positive i -> row N + i)
                 "rows2" -> TripletBatch(token_rows [2,B], neg_index)      the same with negatives by index
                 With sampler="device" the rows / cache forms cost the host nothing per batch but a tensor
                 view: the whole epoch's shuffle, negative draws and row triples are a dozen device ops.
        shard: (rank, world) — data parallel: every rank builds the SAME global batches (same seed) and this
                    loader yields only rank's contiguous shard of each (`training.shard_sizes`); row emits
                    ("cache", "rows3", "rows2") only, negatives always as a third row (an in-batch index does
                    not survive the cut).  `global_batch_sizes()` gives the sizes `train_epoch` wants beside it.
        pad: "batch" — S = the longest row of the batch, what `padding=True` gives each call of
                    backend/model.py:43-45 (costs one device->host read per batch); "max" — S = the
                    longest row of the collection (no read-back; K1 never fetches past a row's length,
                    so the extra padding is free).  Default: "batch" with the reference sampler,
                    "max" with the device sampler."""
        if emit not in ("ids3", "index", "cache", "rows3", "rows2"):
            raise ValueError("emit must be 'ids3', 'index', 'cache', 'rows3' or 'rows2'")
        if sampler not in ("reference", "device"):
            raise ValueError("sampler must be 'reference' or 'device'")
        if pad not in (None, "batch", "max"):
            raise ValueError("pad must be 'batch' or 'max'")
        if shard is not None and emit not in ("cache", "rows3", "rows2"):
            raise ValueError("shard= needs a row emit ('cache', 'rows3' or 'rows2')")
        self.shard = None if shard is None or shard[1] <= 1 else (int(shard[0]), int(shard[1]))
        self.emit = emit
        self.sampler = sampler
        self.pad = pad or ("max" if sampler == "device" else "batch")
        self.batch_size = batch_size
        self.device = torch.device(device)
        self.shuffle = shuffle
        self.rng = random.Random(seed) if seed is not None else random
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(seed if seed is not None else random.getrandbits(60))
        p_nonempty = None
        if isinstance(packed, _Rows):
            self.query_ids = [int(x) for x in packed.query_ids]
            self.q_ids, self.q_lens = packed.q_ids.to(self.device, torch.int32), packed.q_lens.to(self.device, torch.int32)
            self.p_ids, self.p_lens = packed.p_ids.to(self.device, torch.int32), packed.p_lens.to(self.device, torch.int32)
        elif packed is not None:
            pq, pp, qids = packed
            self.query_ids = [int(x) for x in qids]
            self.q_ids, self.q_lens = pq.padded(self.device)
            self.p_ids, self.p_lens = pp.padded(self.device)
        else:
            n = len(dataset)
            items = [dataset[i] for i in range(n)]
            self.query_ids = [int(it["query_id"]) for it in items]
            p_nonempty = [bool(it["positive"]) for it in items]  # data.py:126: an empty passage is never a negative
            self.q_ids, self.q_lens = self._tokenize(tokenizer, [it["query"] for it in items], max_length)
            self.p_ids, self.p_lens = self._tokenize(tokenizer, [it["positive"] for it in items], max_length)
        self._qid_t = torch.tensor(self.query_ids, dtype=torch.int64, device=self.device)
        # per item: may its positive serve as a negative?  (always, unless built from strings with empty passages)
        self.p_nonempty = None if all(p_nonempty or [True]) else list(p_nonempty)
        self._ok_t = None if self.p_nonempty is None else torch.tensor(self.p_nonempty, dtype=torch.bool, device=self.device)
        self._padded = None  # (q_ids, p_ids) padded to one common S (pad="max")

    def _tokenize(self, tokenizer, texts, max_length):
        from .textstore import hf_token_rows, is_hf_tokenizer

        if is_hf_tokenizer(tokenizer) and len(texts):  # the same ids without transformers' tensor conversion
            rows, lens = hf_token_rows(tokenizer, texts, max_length)
            ids = torch.from_numpy(rows)
            ids = torch.where(ids == -(2 ** 31), torch.zeros((), dtype=torch.int32), ids)  # this loader's rows are zero-padded + lens
            return ids.to(self.device), torch.from_numpy(lens).to(self.device)
        tok = tokenizer(texts, padding=True, truncation=True, return_tensors="pt", max_length=max_length)
        ids, mask = tok["input_ids"], tok["attention_mask"]
        lens = mask.sum(1)
        if int(lens.max()) < ids.shape[1]:  # a tokenizer may pad beyond the longest row
            ids, mask = ids[:, : max(int(lens.max()), 1)], mask[:, : max(int(lens.max()), 1)]
        if not torch.equal(mask, (torch.arange(mask.shape[1]).unsqueeze(0) < lens.unsqueeze(1)).to(mask.dtype)):
            raise ValueError("TokenTripletLoader expects prefix (right-padded) attention masks")
        return ids.to(torch.int32).to(self.device), lens.to(torch.int32).to(self.device)

    def __len__(self) -> int:
        return (len(self.query_ids) + self.batch_size - 1) // self.batch_size

    def global_batch_sizes(self):
        """Sizes of the epoch's (global) batches, in order — what `TripletTrainer.train_epoch(...,
        global_batch_sizes=)` needs when this loader yields per-rank shards."""
        n, B = len(self.query_ids), self.batch_size
        return [B] * (n // B) + ([n % B] if n % B else [])

    def save_packed(self, prefix: str) -> None:
        """Write `<prefix>.q.*`, `<prefix>.p.*` (PackedTokens) and `<prefix>.query_ids.bin` (int64)."""
        import numpy as np

        PackedTokens.write(prefix + ".q", self.q_ids, self.q_lens)
        PackedTokens.write(prefix + ".p", self.p_ids, self.p_lens)
        np.asarray(self.query_ids, dtype=np.int64).tofile(prefix + ".query_ids.bin")

    @classmethod
    def from_packed(cls, prefix: str, batch_size: int, device, **kw) -> "TokenTripletLoader":
        import numpy as np

        qids = np.fromfile(prefix + ".query_ids.bin", dtype=np.int64)
        return cls(None, None, batch_size, device,
                   packed=(PackedTokens.load(prefix + ".q"), PackedTokens.load(prefix + ".p"), qids), **kw)

    @classmethod
    def from_tensors(cls, q_ids: torch.Tensor, q_lens: torch.Tensor, p_ids: torch.Tensor, p_lens: torch.Tensor,
                     query_ids, batch_size: int, device, **kw) -> "TokenTripletLoader":
        """Already tokenised, padded rows: q_ids [N,Sq] / p_ids [N,Sp] int32, lengths [N] int32,
        query_ids [N] (one per item).  No copy is made of tensors already on `device`."""
        self = cls.__new__(cls)
        cls.__init__(self, None, None, batch_size, device, packed=_Rows(q_ids, q_lens, p_ids, p_lens, query_ids), **kw)
        return self

    def peek(self):
        """The first batch an iteration started now would yield, WITHOUT consuming anything: the host and device
        random streams (`rng`, `gen`) are put back afterwards, so a later epoch shuffles and draws its negatives
        exactly as if this call had not happened (run_training calibrates on it before epoch 1)."""
        host_state = self.rng.getstate()
        dev_state = self.gen.get_state()
        try:
            return next(iter(self))
        finally:
            self.rng.setstate(host_state)
            self.gen.set_state(dev_state)

    def all_texts(self):
        """(ids [2N, S], lens [2N]) of every query then every positive — the input of the feature cache."""
        S = max(self.q_ids.shape[1], self.p_ids.shape[1])
        pad = lambda t: torch.nn.functional.pad(t, (0, S - t.shape[1]))
        return torch.cat([pad(self.q_ids), pad(self.p_ids)]).contiguous(), torch.cat([self.q_lens, self.p_lens]).contiguous()

    def _common(self):
        """(q_ids, p_ids) zero-padded to one common S, built once (pad="max")."""
        if self._padded is None:
            S = max(self.q_ids.shape[1], self.p_ids.shape[1])
            pad = lambda t: t if t.shape[1] == S else torch.nn.functional.pad(t, (0, S - t.shape[1])).contiguous()
            self._padded = (pad(self.q_ids), pad(self.p_ids))
        return self._padded

    def _epoch_negatives(self, order_t):
        """sampler="device": the in-batch negative indices of EVERY batch of the epoch in one vectorised
        draw — full batches as one [n_batches, B] call, the ragged last batch on its own.  Returns
        (neg_full [n_full, B] | None, neg_tail [r] | None), int64 in-batch indices on the device.
        One host read per EPOCH (is some batch without two distinct query ids?), none per batch."""
        n, B = order_t.numel(), self.batch_size
        n_full, r = divmod(n, B)
        if r == 1 or (B == 1 and n_full):
            raise IndexError("cannot draw an in-batch negative from a batch of one")
        qt = self._qid_t[order_t]
        okt = None if self._ok_t is None else self._ok_t[order_t]
        full = qt[: n_full * B].view(n_full, B) if n_full else None
        tail = qt[n_full * B:] if r else None
        ok_full = None if okt is None or not n_full else okt[: n_full * B].view(n_full, B)
        ok_tail = None if okt is None or not r else okt[n_full * B:]
        bad = torch.zeros((), dtype=torch.bool, device=self.device)
        if okt is None:
            if n_full:
                bad = bad | (full.min(1).values == full.max(1).values).any()
            if r:
                bad = bad | (tail.min() == tail.max())
        else:  # every item must find a non-empty positive of another query: count them exactly (sort-based, O(B log B))
            for qq, oo in ((full, ok_full), (tail.view(1, -1) if r else None, ok_tail.view(1, -1) if r else None)):
                if qq is not None:
                    bad = bad | (_admissible_runs(qq, oo)[3] <= 0).any()
        if bool(bad):  # the reference would spin forever in create_triplets (backend/data.py:124-128)
            raise RuntimeError("an item of some batch has no admissible negative (every other item shares its query_id"
                               + ("" if okt is None else " or has an empty positive") + ")")
        neg_full = sample_negative_indices_device(full, self.gen, check=False, positive_nonempty=ok_full) if n_full else None
        neg_tail = sample_negative_indices_device(tail, self.gen, check=False, positive_nonempty=ok_tail) if r else None
        return neg_full, neg_tail

    def __iter__(self):
        n, B = len(self.query_ids), self.batch_size
        if self.sampler == "device":  # shuffle on the device too: nothing of the epoch's set-up is O(n) host work
            order_t = (torch.randperm(n, generator=self.gen, device=self.device) if self.shuffle
                       else torch.arange(n, device=self.device))
            order = None
        else:
            order = list(range(n))
            if self.shuffle:
                self.rng.shuffle(order)
            order_t = torch.tensor(order, dtype=torch.int64, device=self.device)  # one host->device copy per epoch
        n_full = n // B
        neg_full = neg_tail = rows_full = None
        by_rows = self.emit in ("cache", "rows3", "rows2")
        if self.sampler == "device":
            neg_full, neg_tail = self._epoch_negatives(order_t)
            if by_rows and n_full:  # the whole epoch's row triples at once: [n_full, roles, B] int32
                it_all = order_t[: n_full * B].view(n_full, B)
                parts = [it_all, it_all + n]
                if self.emit == "rows3" or self.shard is not None:
                    parts.append(it_all.gather(1, neg_full) + n)
                rows_full = torch.stack(parts, dim=1).to(torch.int32)
                neg_full32 = neg_full.to(torch.int32)
                if self.shard is not None:  # this rank's columns of every full batch, cut once per epoch
                    from .training import shard_sizes

                    sizes, offs = shard_sizes(B, self.shard[1])
                    lo = offs[self.shard[0]]
                    rows_full = rows_full[:, :, lo: lo + sizes[self.shard[0]]]
                rows_full = rows_full.contiguous()
        if by_rows:
            from .training import TripletBatch, shard_sizes

            three = self.emit == "rows3" or self.shard is not None

            def as_batch(rows, ni):
                if three:
                    ni = None
                if self.emit == "cache":
                    return TripletBatch(cache_rows=rows, neg_index=ni)
                return TripletBatch(token_rows=rows, neg_index=ni)
        for bi, s in enumerate(range(0, n, B)):
            if rows_full is not None and bi < n_full:  # per batch: two tensor views, nothing else
                yield as_batch(rows_full[bi], neg_full32[bi])
                continue
            it = order_t[s: s + B]
            nb = it.shape[0]
            if self.sampler == "device":
                neg_t = neg_full[bi] if bi < n_full else neg_tail
            else:
                sel = order[s: s + B]
                neg_t = torch.tensor(sample_negative_indices(
                    [self.query_ids[i] for i in sel], self.rng,
                    None if self.p_nonempty is None else [self.p_nonempty[i] for i in sel]), device=self.device)
            if by_rows:
                parts = [it, it + n] + ([it[neg_t] + n] if three else [])
                rows = torch.stack(parts).to(torch.int32)
                if self.shard is not None:
                    sizes, offs = shard_sizes(nb, self.shard[1])
                    lo = offs[self.shard[0]]
                    rows = rows[:, lo: lo + sizes[self.shard[0]]]
                yield as_batch(rows.contiguous(), neg_t.to(torch.int32))
                continue
            roles = 3 if self.emit == "ids3" else 2
            ql, pl = self.q_lens[it], self.p_lens[it]
            if self.pad == "max":  # no read-back: every batch is padded to the collection's longest row
                qsrc, psrc = self._common()
                S = qsrc.shape[1]
                ids = torch.empty((roles, nb, S), dtype=torch.int32, device=self.device)
                torch.index_select(qsrc, 0, it, out=ids[0])
                torch.index_select(psrc, 0, it, out=ids[1])
                if roles == 3:
                    nt = it[neg_t]
                    torch.index_select(psrc, 0, nt, out=ids[2])
                    yield ids, torch.stack([ql, pl, self.p_lens[nt]])
                    continue
            else:  # pad to the longest row of this batch, like `padding=True` does per call
                q, p = self.q_ids[it], self.p_ids[it]
                if roles == 3:
                    nt = it[neg_t]
                    ng, nl = self.p_ids[nt], self.p_lens[nt]
                    S = int(torch.stack([ql.max(), pl.max(), nl.max()]).max())
                else:
                    S = int(torch.stack([ql.max(), pl.max()]).max())
                ids = torch.zeros((roles, nb, S), dtype=torch.int32, device=self.device)
                ids[0, :, : min(S, q.shape[1])] = q[:, :S]
                ids[1, :, : min(S, p.shape[1])] = p[:, :S]
                if roles == 3:
                    ids[2, :, : min(S, ng.shape[1])] = ng[:, :S]
                    yield ids, torch.stack([ql, pl, nl])
                    continue
            from .training import TripletBatch

            yield TripletBatch(ids=ids, lens=torch.stack([ql, pl]), neg_index=neg_t.to(torch.int32))
