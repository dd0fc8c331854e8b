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


def sample_negative_indices(query_ids: Sequence[int], rng=random) -> List[int]:
    """backend/data.py:113-137 on indices: neg_idx[i] = j, drawn like
    `_get_random_negative` (same `choice` call sequence, so the same seed gives the
    same draws as the reference) and rejected while query_ids[j] == query_ids[i].
    Like the reference this never terminates if every item shares one query_id and
    raises IndexError for a batch of one."""
    n = len(query_ids)
    out = []
    for i in range(n):
        while True:
            j = rng.choice([k for k in range(n) if k != i])
            if query_ids[j] != query_ids[i]:
                out.append(j)
                break
    return out


def sample_negative_indices_device(query_ids: torch.Tensor, generator: Optional[torch.Generator] = None,
                                   max_rounds: int = 64) -> torch.Tensor:
    """Same distribution as `sample_negative_indices` (uniform over the other in-batch items,
    redrawn while the query_id is equal — backend/data.py:113-137), drawn for the whole batch at
    once with tensor ops on `query_ids`' device: O(B) per round instead of the reference's O(B^2)
    Python list building per batch.  Not the same *draws* as the reference under a shared seed.
    Raises if some item has no admissible negative (the reference would loop forever)."""
    B = query_ids.numel()
    if B < 2:
        raise IndexError("cannot draw an in-batch negative from a batch of one")  # reference: random.choice([])
    dev = query_ids.device
    ar = torch.arange(B, device=dev)
    neg = torch.empty(B, dtype=torch.long, device=dev)
    todo = ar
    for _ in range(max_rounds):
        j = torch.randint(0, B - 1, (todo.numel(),), generator=generator, device=dev)
        j = j + (j >= todo).long()          # uniform over the B-1 items other than i
        ok = query_ids[j] != query_ids[todo]
        neg[todo[ok]] = j[ok]
        todo = todo[~ok]
        if todo.numel() == 0:
            return neg
    # leftovers: items whose admissible set is tiny or empty — enumerate it exactly
    for i in todo.tolist():
        cand = ar[query_ids != query_ids[i]]
        if cand.numel() == 0:
            raise RuntimeError("every item of the batch shares one query_id: no admissible negative")
        neg[i] = cand[torch.randint(0, cand.numel(), (1,), generator=generator, device=dev)]
    return neg


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


class TripletDataLoader:
    """Creates (queries, positives, negatives) string batches.  Mirrors backend/data.py:90-152."""

    def __init__(self, dataset, batch_size: int = 1024, num_workers: int = 4, device: Optional[torch.device] = None):
        self.dataset = dataset
        self.batch_size = batch_size
        self.dataloader = DataLoader(
            dataset,
            batch_size=batch_size,
            shuffle=True,
            num_workers=num_workers,
            pin_memory=False,  # the items are Python strings; nothing to pin (data.py:106-108 pins for cuda)
            persistent_workers=True if num_workers > 0 else False,
        )

    def create_triplets(self, batch: list) -> Triplet:
        queries = [item["query"] for item in batch]
        positives = [item["positive"] for item in batch]
        neg_idx = sample_negative_indices([int(item["query_id"]) for item in batch])
        negatives = [batch[j]["positive"] for j in neg_idx]
        return queries, positives, negatives

    def __iter__(self) -> Iterator[Triplet]:
        for batch in self.dataloader:
            batch_list = [
                {"query": batch["query"][i], "positive": batch["positive"][i], "query_id": batch["query_id"][i]}
                for i in range(len(batch["query"]))
            ]
            yield self.create_triplets(batch_list)

    def __len__(self) -> int:
        return len(self.dataloader)


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
                 packed: Optional[Tuple["PackedTokens", "PackedTokens", Sequence[int]]] = None):
        """sampler: "reference" — `sample_negative_indices`, the reference's draws under the same seed
                    (O(B^2) Python per batch, like backend/data.py:132-137); "device" — the same
                    distribution drawn with tensor ops (`sample_negative_indices_device`).
        packed: (queries, positives, query_ids) as PackedTokens instead of tokenising `dataset`.
        emit: "ids3"  -> (ids [3,B,S], lens [3,B])                      three gathers per batch
                 "index" -> TripletBatch(ids [2,B,S], lens, neg_index)     negatives by in-batch index
                 "cache" -> TripletBatch(cache_rows [2,B], neg_index)      rows of a frozen-feature cache
                            laid out [queries (N); positives (N)] — build it with
                            `trainer.build_feature_cache(*loader.all_texts())`"""
        if emit not in ("ids3", "index", "cache"):
            raise ValueError("emit must be 'ids3', 'index' or 'cache'")
        if sampler not in ("reference", "device"):
            raise ValueError("sampler must be 'reference' or 'device'")
        self.emit = emit
        self.sampler = sampler
        self.batch_size = batch_size
        self.device = torch.device(device)
        self.shuffle = shuffle
        self.rng = random.Random(seed) if seed is not None else random
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(seed if seed is not None else random.getrandbits(60))
        if packed is not None:
            pq, pp, qids = packed
            self.query_ids = [int(x) for x in qids]
            self.q_ids, self.q_lens = pq.padded(self.device)
            self.p_ids, self.p_lens = pp.padded(self.device)
        else:
            n = len(dataset)
            items = [dataset[i] for i in range(n)]
            self.query_ids = [int(it["query_id"]) for it in items]
            self.q_ids, self.q_lens = self._tokenize(tokenizer, [it["query"] for it in items], max_length)
            self.p_ids, self.p_lens = self._tokenize(tokenizer, [it["positive"] for it in items], max_length)
        self._qid_t = torch.tensor(self.query_ids, device=self.device)

    def _tokenize(self, tokenizer, texts, max_length):
        tok = tokenizer(texts, padding=True, truncation=True, return_tensors="pt", max_length=max_length)
        ids, mask = tok["input_ids"], tok["attention_mask"]
        lens = mask.sum(1)
        if not torch.equal(mask, (torch.arange(mask.shape[1]).unsqueeze(0) < lens.unsqueeze(1)).to(mask.dtype)):
            raise ValueError("TokenTripletLoader expects prefix (right-padded) attention masks")
        return ids.to(torch.int32).to(self.device), lens.to(torch.int32).to(self.device)

    def __len__(self) -> int:
        return (len(self.query_ids) + self.batch_size - 1) // self.batch_size

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

    def all_texts(self):
        """(ids [2N, S], lens [2N]) of every query then every positive — the input of the feature cache."""
        S = max(self.q_ids.shape[1], self.p_ids.shape[1])
        pad = lambda t: torch.nn.functional.pad(t, (0, S - t.shape[1]))
        return torch.cat([pad(self.q_ids), pad(self.p_ids)]).contiguous(), torch.cat([self.q_lens, self.p_lens]).contiguous()

    def __iter__(self):
        n = len(self.query_ids)
        order = list(range(n))
        if self.shuffle:
            self.rng.shuffle(order)
        for s in range(0, n, self.batch_size):
            idx = order[s: s + self.batch_size]
            it = torch.tensor(idx, device=self.device)
            if self.sampler == "device":
                neg_t = sample_negative_indices_device(self._qid_t[it], self.gen)
            else:
                neg_t = torch.tensor(sample_negative_indices([self.query_ids[i] for i in idx], self.rng), device=self.device)
            if self.emit != "ids3":
                from .training import TripletBatch

                ni = neg_t.to(torch.int32)
                if self.emit == "cache":
                    n_items = len(self.query_ids)
                    rows = torch.stack([it, it + n_items]).to(torch.int32).contiguous()
                    yield TripletBatch(cache_rows=rows, neg_index=ni)
                    continue
                q, ql = self.q_ids[it], self.q_lens[it]
                p, pl = self.p_ids[it], self.p_lens[it]
                S = int(max(int(ql.max()), int(pl.max())))
                ids = torch.zeros((2, len(idx), S), dtype=torch.int32, device=self.device)
                ids[0, :, : min(S, q.shape[1])] = q[:, :S]
                ids[1, :, : min(S, p.shape[1])] = p[:, :S]
                yield TripletBatch(ids=ids, lens=torch.stack([ql, pl]), neg_index=ni)
                continue
            nt = it[neg_t]
            q, ql = self.q_ids[it], self.q_lens[it]
            p, pl = self.p_ids[it], self.p_lens[it]
            ng, nl = self.p_ids[nt], self.p_lens[nt]
            # pad to the longest row of this batch, like `padding=True` does per call
            S = int(max(int(ql.max()), int(pl.max()), int(nl.max())))
            ids = torch.zeros((3, len(idx), S), dtype=torch.int32, device=self.device)
            ids[0, :, : min(S, q.shape[1])] = q[:, :S]
            ids[1, :, : min(S, p.shape[1])] = p[:, :S]
            ids[2, :, : min(S, ng.shape[1])] = ng[:, :S]
            yield ids, torch.stack([ql, pl, nl])
