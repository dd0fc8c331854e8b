"""TextRowStore — the token store behind STRING batches of the drop-in `train_epoch` route.

backend/training.py:36-42 hands `model.encode_*` three lists of strings per batch and backend/model.py:43-48 runs the
tokenizer on each of them, every batch of every epoch (9-15 epochs over the same texts, backend/main.py:18-85).  The ids
of a text never change, so here a text is tokenised the FIRST time it is seen: its ids go into one row of a
device-resident int32 store (`TT_PAD_ID` behind the text's own tokens: include/tt_hotpath.h), and from then on a batch is
`token_rows [3, B]` through `tt_engine_pool_rows` — 12 bytes per triplet, no tokenizer, no [B, S] int64 traffic.

Two ways a text finds its row:

* `TextColumn` lists (data.py: what this package's `TripletDataLoader` yields — plain `list[str]` to every other
  consumer) say which dataset item each string is; the row is arithmetic on that index, a batch costs a few numpy ops.
* any other `list[str]`: a dict lookup per text (hash + compare of the whole string: ~0.1 us per text for strings that
  are the same Python objects every epoch, ~1 us for strings re-created by DataLoader worker processes — host-bound at
  BASELINE configs[1]'s batch, but two orders of magnitude below the tokenizer it replaces).

Only a tokenizer that maps a text to the same ids whatever else is in the call may be cached: transformers'
tokenizers (`PreTrainedTokenizerBase`), or a callable that declares `tt_per_text = True`.
"""
from __future__ import annotations

import operator
import time
from typing import Callable, Sequence

import numpy as np
import torch

PAD32 = -(2 ** 31)  # TT_PAD_ID32 (include/tt_hotpath.h): weight 0 in every gather form, never read as a table row


def per_text_tokenizer(tok) -> bool:
    """True where tokenising a text alone or inside any batch gives the same ids (padding aside)."""
    flag = getattr(tok, "tt_per_text", None)
    if flag is not None:
        return bool(flag)
    try:
        from transformers import PreTrainedTokenizerBase
    except Exception:  # pragma: no cover  (transformers is optional)
        return False
    return isinstance(tok, PreTrainedTokenizerBase)


def is_hf_tokenizer(tok) -> bool:
    try:
        from transformers import PreTrainedTokenizerBase
    except Exception:  # pragma: no cover
        return False
    return isinstance(tok, PreTrainedTokenizerBase)


def _backend_ids(tok, texts, max_length: int):
    """The ids of the tokenizer's public list-form call, straight from its Rust backend (`tok.backend_tokenizer.encode_batch`
    after the same `set_truncation_and_padding` the public call performs) — without transformers' per-text Python
    (`_convert_encoding`: 9 us per text, 1.9 of the 3.2 s the public call takes for 211,000 texts on the GPU box's 16 cores,
    where the Rust tokenizer itself needs 1.1 s).  None where the tokenizer has no such backend, or where the first call's
    self-check against the public call disagrees (then the public call is used for good)."""
    state = getattr(tok, "_tt_backend_ids", None)
    if state is False:
        return None
    try:
        from transformers.tokenization_utils_base import TruncationStrategy
        from transformers.utils import PaddingStrategy

        backend, setter = tok.backend_tokenizer, tok.set_truncation_and_padding

        def encode(batch):
            setter(padding_strategy=PaddingStrategy.DO_NOT_PAD, truncation_strategy=TruncationStrategy.LONGEST_FIRST,
                   max_length=max_length, stride=0, pad_to_multiple_of=None, padding_side=None)
            return [e.ids for e in backend.encode_batch(batch, add_special_tokens=True)]

        if state is None:  # once per tokenizer object: the backend's ids must be the public call's
            probe = list(texts[:16])
            want = tok(probe, truncation=True, max_length=max_length, return_attention_mask=False, return_token_type_ids=False)["input_ids"]
            ok = encode(probe) == [list(x) for x in want]
            try:
                tok._tt_backend_ids = ok
            except Exception:
                return None
            if not ok:
                return None
        return encode(list(texts))
    except Exception:
        try:
            tok._tt_backend_ids = False
        except Exception:
            pass
        return None


def hf_token_rows(tok, texts: Sequence[str], max_length: int = 512):
    """The ids of `tok(texts, padding=True, truncation=True, return_tensors="pt", max_length=512)` (backend/model.py:43-45)
    as (rows int32 [n, longest] numpy with TT_PAD_ID behind each text's own tokens, lens int32 [n]) — the same tokenizer,
    the same truncation, the same ids; asked for as plain lists without padding, mask or tensors (from the Rust backend
    directly where there is one: `_backend_ids`), and padded here with numpy.  transformers' Python-side conversion of a
    padded batch into tensors costs more than the tokenisation itself (BertTokenizerFast, 4096 texts of 128 tokens on 8
    cores: 0.71 s for the reference's call, 0.18 s inside the Rust tokenizer, 0.26 s this way)."""
    import itertools

    t0 = time.perf_counter()
    ids = _backend_ids(tok, texts, max_length) if len(texts) else None
    if ids is not None and hasattr(tok, "tt_seconds"):  # a bench's timing wrapper (bench_eval.timed_tokenizer) sees public calls only
        tok.tt_seconds += time.perf_counter() - t0
        tok.tt_texts += len(ids)
    if ids is None:
        ids = tok(list(texts), truncation=True, max_length=max_length, return_attention_mask=False, return_token_type_ids=False)["input_ids"]
    n = len(ids)
    lens = np.fromiter(map(len, ids), dtype=np.int64, count=n)
    total = int(lens.sum())
    flat = np.fromiter(itertools.chain.from_iterable(ids), dtype=np.int64, count=total)
    if total and (flat.min() < 0 or flat.max() > 2 ** 31 - 1):
        raise ValueError("token ids outside int32")
    W = max(int(lens.max()) if n else 1, 1)
    rows = np.full((n, W), PAD32, dtype=np.int32)
    rows[np.arange(W)[None, :] < lens[:, None]] = flat
    return rows, lens.astype(np.int32)


class TextRowStore:
    """ids [capacity, width] int32 on `device`; rows are handed out once and filled once."""

    def __init__(self, device, max_S: int = 512, rows: int = 4096, width: int = 64):
        self.device, self.max_S = torch.device(device), int(max_S)
        self.ids = torch.full((int(rows), int(width)), PAD32, dtype=torch.int32, device=self.device)
        self.filled = np.zeros(int(rows), dtype=bool)
        self.n = 0                 # rows handed out
        self._by_text = {}         # tokenizer key -> {text: row}
        self._by_source = {}       # id(dataset) -> (dataset, base row, len(dataset))
        self.tokenize_s = 0.0      # host seconds inside tokenizer calls
        self.texts_tokenized = 0
        self.tokens_stored = 0

    # -- rows ----------------------------------------------------------------------
    def _reserve(self, k: int) -> int:
        base = self.n
        self.n += int(k)
        if self.n > self.filled.shape[0]:
            cap = max(self.n, 2 * self.filled.shape[0])
            self.filled = np.concatenate([self.filled, np.zeros(cap - self.filled.shape[0], dtype=bool)])
        return base

    def _room(self, rows_needed: int, width_needed: int) -> None:
        """The device tensor holds rows [0, rows_needed) at width >= width_needed.  A re-allocation leaves the old tensor
        to the caching allocator; gathers still reading it hold it through `record_stream` (TripletTrainer.gather)."""
        cap, W = self.ids.shape
        if width_needed > self.max_S:
            raise ValueError(f"a text of {width_needed} tokens exceeds {self.max_S} (the tokenizer's max_length, backend/model.py:44)")
        new_W = W if width_needed <= W else min(self.max_S, -(-width_needed // 64) * 64)
        new_cap = cap if rows_needed <= cap else max(rows_needed, 2 * cap)
        if (new_cap, new_W) != (cap, W):
            grown = torch.full((new_cap, new_W), PAD32, dtype=torch.int32, device=self.device)
            grown[:cap, :W].copy_(self.ids)
            self.ids = grown

    def _fill(self, rows: np.ndarray, texts: Sequence[str], tokenize: Callable) -> None:
        t0 = time.perf_counter()
        enc = tokenize(list(texts))  # backend/model.py:43-45 on the texts not seen before (CPU tensors, or packed rows)
        self.tokenize_s += time.perf_counter() - t0
        if isinstance(enc, tuple):  # hf_token_rows: int32 rows with TT_PAD_ID already behind each text's tokens
            packed, lens = torch.from_numpy(enc[0]), enc[1]
            self.tokens_stored += int(lens.sum())
            self._room(int(rows.max()) + 1, packed.shape[1])
            self.ids[:, : packed.shape[1]].index_copy_(0, torch.from_numpy(rows.astype(np.int64)).to(self.device), packed.to(self.device))
            self.filled[rows] = True
            self.texts_tokenized += len(texts)
            return
        ids = torch.as_tensor(enc["input_ids"])
        mask = enc.get("attention_mask")
        if ids.dim() != 2 or ids.shape[0] != len(texts):
            raise ValueError("tokenizer returned input_ids that are not [len(texts), S]")
        if int(ids.max()) > 2 ** 31 - 1 or int(ids.min()) < 0:
            raise ValueError("token ids outside int32")
        packed = ids.to(torch.int32)
        if mask is not None:
            mask = torch.as_tensor(mask)
            if not bool(((mask == 0) | (mask == 1)).all()):
                raise ValueError("a weight mask (values other than 0/1) cannot be stored as rows")
            packed = torch.where(mask != 0, packed, torch.full((), PAD32, dtype=torch.int32))
            self.tokens_stored += int(mask.sum())
        else:
            self.tokens_stored += ids.numel()
        self._room(int(rows.max()) + 1, packed.shape[1])
        self.ids[:, : packed.shape[1]].index_copy_(0, torch.from_numpy(rows.astype(np.int64)).to(self.device),
                                                   packed.contiguous().to(self.device))
        self.filled[rows] = True
        self.texts_tokenized += len(texts)

    # -- the two look-ups ------------------------------------------------------------
    def rows_of_texts(self, key, texts: Sequence[str], tokenize: Callable) -> np.ndarray:
        d = self._by_text.setdefault(key, {})
        if len(texts) == 0:
            return np.zeros(0, dtype=np.int64)
        try:
            got = operator.itemgetter(*texts)(d) if len(texts) > 1 else (d[texts[0]],)
        except KeyError:
            missing = [t for t in dict.fromkeys(texts) if t not in d]
            base = self._reserve(len(missing))
            self._fill(np.arange(base, base + len(missing)), missing, tokenize)
            d.update(zip(missing, range(base, base + len(missing))))
            got = operator.itemgetter(*texts)(d) if len(texts) > 1 else (d[texts[0]],)
        return np.fromiter(got, dtype=np.int64, count=len(texts))

    def rows_of_column(self, col, tokenize: Callable) -> np.ndarray:
        """`col`: a data.TextColumn — text i is field `col.field` of item `col.rows[i]` of dataset `col.source`."""
        ent = self._by_source.get(id(col.source))
        if ent is None or ent[2] != len(col.source):
            n = len(col.source)
            ent = self._by_source[id(col.source)] = (col.source, self._reserve(2 * n), n)  # queries, then positives
        _, base, n = ent
        rows = np.asarray(col.rows, dtype=np.int64)
        if rows.shape[0] != len(col) or (rows.size and (rows.min() < 0 or rows.max() >= n)):
            raise ValueError("TextColumn.rows does not index its source dataset")
        rows = rows + (base if col.field == "query" else base + n)
        new = ~self.filled[rows]
        if new.any():
            idx = np.flatnonzero(new)
            uniq, first = np.unique(rows[idx], return_index=True)
            self._fill(uniq, [col[int(i)] for i in idx[first]], tokenize)
        return rows

    def stats(self) -> dict:
        return {"rows": int(self.filled.sum()), "width": int(self.ids.shape[1]), "texts_tokenized": self.texts_tokenized,
                "tokens_stored": self.tokens_stored, "tokenize_s": self.tokenize_s,
                "device_bytes": int(self.ids.numel() * 4)}
