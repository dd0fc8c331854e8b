"""Host-side mirror of the training loop of backend/training.py (`train_epoch`,
`train_epoch_optimized`, `run_training`) over the HIP hot path.

Two ways to run the batch body of backend/training.py:36-53:

* `train_epoch(model, dataloader, criterion, optimizer)` — the reference's
  signature and call sequence on the mirrored nn.Modules (each op is one HIP
  kernel behind torch.autograd); any torch optimizer works, `FusedAdam` runs K4.
* `TripletTrainer` (trainer.py; re-exported here) — the same arithmetic as a few C
  calls per batch on the fused engine (flat parameters, no autograd graph,
  device-side loss accumulation instead of a per-batch `loss.item()` sync),
  data-parallel over `torch.distributed` (RCCL on GPUs) with one flat-bucket
  all-reduce per step that overlaps the next batch's gather.  With `FusedAdam`
  (optim.py) `train_epoch` itself routes its batch body to it (`_EngineRoute`).
"""
from __future__ import annotations

import collections
import random
from typing import Optional

import torch
import torch.distributed as dist

from .engine import flat_param_slices
from .model import TripletLoss, TwoTowersModel

try:  # logging only (backend/training.py:13,57-58); optional here
    import wandb  # type: ignore

    _HAVE_WANDB = hasattr(wandb, "log")
except Exception:  # pragma: no cover
    wandb = None
    _HAVE_WANDB = False


from .optim import FusedAdam  # noqa: E402,F401  (re-exported: the reference's users import everything from `training`)
from .trainer import (Batch, TrainerConfig, TripletBatch, TripletTrainer, _TorchEvent, _proj_dim,  # noqa: E402,F401
                      _require_same_frozen_encoder, as_batch, shard_batch, shard_sizes)


# --------------------------------------------------------------------------- #
# Reference-signature loops (module path)
# --------------------------------------------------------------------------- #
def _encode3(model, queries, positives, negatives):
    """The three encodes of backend/training.py:40-42.  Batches may be the reference's string lists or pre-tokenised
    (input_ids, attention_mask) pairs.  The positives and the negatives both go through the DOCUMENT tower, so they are
    encoded in one call — one gather+pool launch and one projection instead of two of each, and one backward through the
    tower (its gradient is the sum over both row blocks either way) — and split afterwards; padding a shorter block
    with masked-out zeros changes nothing (masked tokens are never read)."""
    if isinstance(queries, (list, tuple)) and len(queries) and isinstance(queries[0], str):
        dt = model.document_tower
        if getattr(dt, "encoder_mode", None) != "embed":  # `bert` mode: the encoder wants every tokenizer field; two calls
            return model.encode_queries(queries), model.encode_documents(positives), model.encode_documents(negatives)
        tp, tn = dt.tokenize(positives), dt.tokenize(negatives)  # the reference's two tokenizer calls (model.py:43-48)
        query_embeds = model.encode_queries(queries)
        _, p_emb, n_emb = _encode3(model, None, (tp["input_ids"], tp.get("attention_mask")),
                                   (tn["input_ids"], tn.get("attention_mask")))
        return query_embeds, p_emb, n_emb
    (ip, mp), (i_n, mn) = _ids_mask(positives), _ids_mask(negatives)
    enc_q = (lambda: None) if queries is None else (lambda: model.encode_query_ids(*queries))
    if ip.dim() != 2 or ip.dtype != i_n.dtype or (mp is not None and mn is not None and mp.dtype != mn.dtype):
        return (enc_q(), model.encode_document_ids(ip, mp), model.encode_document_ids(i_n, mn))
    blocks = [(ip, mp), (i_n, mn)]
    qt, dt = model.query_tower, model.document_tower
    if (queries is not None and getattr(qt, "encoder_mode", None) == "embed" == getattr(dt, "encoder_mode", None)
            and qt.embedding_table().data_ptr() == dt.embedding_table().data_ptr()):
        iq, mq = _ids_mask(queries)
        if iq.dim() == 2 and iq.dtype == ip.dtype and (mq is None or mp is None or mq.dtype == mp.dtype):
            blocks.insert(0, (iq, mq))  # both towers gather from the SAME frozen table: one gather+pool for all three roles
    S = max(i.shape[1] for i, _ in blocks)
    if all(m is None for _, m in blocks) and all(i.shape[1] == S for i, _ in blocks):
        ids, mask = torch.cat([i for i, _ in blocks]), None
    else:
        pad = lambda t, fill: t if t.shape[1] == S else torch.nn.functional.pad(t, (0, S - t.shape[1]), value=fill)
        ids = torch.cat([pad(i, 0) for i, _ in blocks])
        mask = torch.cat([pad(torch.ones_like(i) if m is None else m, 0) for i, m in blocks])
    if len(blocks) == 3:
        pooled = dt.pool_tokens({"input_ids": ids} if mask is None else {"input_ids": ids, "attention_mask": mask})
        nq = blocks[0][0].shape[0]
        query_embeds = qt.project(pooled[:nq])
        docs = dt.project(pooled[nq:])
    else:
        query_embeds = enc_q()
        docs = model.encode_document_ids(ids, mask)
    positive_embeds, negative_embeds = torch.split(docs, [ip.shape[0], i_n.shape[0]])
    return query_embeds, positive_embeds, negative_embeds


def _ids_mask(pair):
    if isinstance(pair, torch.Tensor):
        return pair, None
    return (pair[0], pair[1] if len(pair) > 1 else None)


def train_epoch(model: TwoTowersModel, dataloader, criterion: TripletLoss, optimizer: torch.optim.Optimizer,
                log_wandb: bool = True, *, fused: Optional[bool] = None, text_cache: Optional[bool] = None) -> float:
    """Train for one epoch.  Same signature, call sequence and return value as backend/training.py:26-63.

    `fused` (additive): None = run the batch body on the fused engine when the arguments allow it — the reference's own
    `torch.optim.Adam(model.parameters(), lr)` (backend/training.py:436; any Adam whose single param group has the
    arithmetic K4 implements: no weight decay / amsgrad / maximize / capturable) or `FusedAdam` over this model's
    trainable tensors, a plain `TripletLoss`, both towers in `embed` mode on one frozen table on the GPU, no per-batch
    wandb logging — with the same results (the engine is held to the same reference fixtures as this loop) at the
    engine's speed: no autograd graph, one gather for the three roles running two batches ahead on its own stream,
    Adam state and parameters staying visible through `optimizer.state` / `model.state_dict()` in the optimizer's OWN
    form (a routed epoch's `optimizer.state_dict()` loads into a fresh optimizer of the same class).  False = the
    module-by-module loop below; True = raise if the engine cannot take this call.
    `text_cache` (additive; string batches on the engine route): None = texts are tokenised the first time they are seen
    and live as rows of a device-resident token store afterwards (textstore.py) where the towers' tokenizer is a
    per-text function (a transformers tokenizer); False = the three tokenizer calls of backend/model.py:43-48 on every
    batch, as the reference does; True = raise if the tokenizer cannot be cached."""
    if fused is not False:
        route = _engine_route(model, criterion, optimizer, log_wandb)
        if route is not None:
            return route.run_epoch(route.admit(dataloader, text_cache))
        if fused:
            raise ValueError("train_epoch(fused=True): needs torch.optim.Adam (default arithmetic) or FusedAdam over the model's "
                             "trainable tensors, TripletLoss, `embed` towers sharing one table on a GPU, and no per-batch "
                             "wandb logging")
    total_loss = 0.0
    num_batches = 0
    pending = []  # device-side losses not yet read: `loss.item()` (training.py:53) stalls the host on every batch, so the
    #               values are read when something needs them — the every-10th-batch print, wandb, the return — in order
    for queries, positives, negatives in dataloader:
        optimizer.zero_grad()
        query_embeds, positive_embeds, negative_embeds = _encode3(model, queries, positives, negatives)
        loss = criterion(query_embeds, positive_embeds, negative_embeds)
        loss.backward()
        optimizer.step()
        pending.append(loss.detach())
        num_batches += 1
        if log_wandb and _HAVE_WANDB:
            wandb.log({"batch_loss": loss.item(), "batch": num_batches})
        if num_batches % 10 == 0:
            for v in torch.stack(pending).tolist():  # the same float additions, in the same order, as a per-batch .item()
                total_loss += v
            pending.clear()
            print(f"Batch {num_batches}, Loss: {loss.item():.4f}")
    if pending:
        for v in torch.stack(pending).tolist():
            total_loss += v
    return total_loss / num_batches


def _engine_route(model, criterion, optimizer, log_wandb):
    """The _EngineRoute for (model, optimizer) if train_epoch's batch body can run on the fused engine, else None."""
    form = _adam_form(optimizer)
    if form is None or type(criterion) is not TripletLoss or not isinstance(model, TwoTowersModel):
        return None
    if log_wandb and _HAVE_WANDB:  # per-batch wandb.log needs every batch's loss on the host: the module loop
        return None
    qt, dt = model.query_tower, model.document_tower
    if getattr(qt, "encoder_mode", None) != "embed" or getattr(dt, "encoder_mode", None) != "embed":
        return None
    table = qt.embedding_table()
    if not table.is_cuda or table.data_ptr() != dt.embedding_table().data_ptr() or table.dtype != torch.float32:
        return None
    if len(optimizer.param_groups) != 1:
        return None
    want = model.trainable_parameters()
    have = [p for p in optimizer.param_groups[0]["params"] if p.requires_grad]
    if len(have) != len(want) or any(a is not b for a, b in zip(have, want)):
        return None
    route = getattr(optimizer, "_tt_route", None)
    if route is None or route.model is not model:
        route = _EngineRoute(model, optimizer)
        optimizer._tt_route = route
    route.form = form
    route.margin = float(criterion.margin)
    return route


def _adam_form(optimizer) -> Optional[str]:
    """"fused" for FusedAdam; "torch" for the reference's optimizer (backend/training.py:436 `torch.optim.Adam(
    model.parameters(), lr)`) when its param group asks for exactly the update K4 computes — Adam with bias correction,
    eps outside the square root, no weight decay / amsgrad / maximize, step counts kept on the host; else None (the
    module loop runs whatever the optimizer is)."""
    if isinstance(optimizer, FusedAdam):
        return "fused"
    if type(optimizer) is not torch.optim.Adam or len(optimizer.param_groups) != 1:
        return None
    g = optimizer.param_groups[0]
    if (g.get("weight_decay", 0) != 0 or g.get("amsgrad") or g.get("maximize") or g.get("capturable") or g.get("differentiable")
            or g.get("fused") or g.get("decoupled_weight_decay")):
        return None
    if isinstance(g["lr"], torch.Tensor) and g["lr"].numel() != 1:
        return None
    return "torch"


def _step_value(form: str, n: int):
    """An optimizer step count in the form `optimizer.state[p]["step"]` has for this optimizer class: FusedAdam keeps an
    int, torch.optim.Adam a 0-d float tensor on the host (torch/optim/adam.py `_init_group`, not capturable / fused)."""
    if form != "torch":
        return int(n)
    try:
        from torch.optim.optimizer import _get_scalar_dtype

        dtype = _get_scalar_dtype()
    except Exception:  # pragma: no cover
        dtype = torch.float32
    return torch.tensor(float(n), dtype=dtype)


class _EngineRoute:
    """train_epoch's batch body (backend/training.py:36-53) on the fused engine, behind the reference's signature.
    Parameters and Adam moments live in the engine's flat buffers; the nn.Parameters and `optimizer.state[p]` are views
    of them, so `model.state_dict()`, `optimizer.state_dict()` and a later module-path `optimizer.step()` see the same
    values."""

    MAX_S = 512  # the tokenizer's max_length (backend/model.py:44)

    def __init__(self, model, optimizer):
        self.model, self.optimizer = model, optimizer
        self.trainer = None
        self.margin = 0.3
        self.form = "fused"  # how optimizer.state[p]["step"] is kept (see _step_value)
        self._bucket = None  # un-applied gradients between epochs: what the reference keeps in `.grad`
        self.text_store = None  # textstore.TextRowStore behind string batches
        self.text_cache = None  # train_epoch's `text_cache` for the running epoch

    def _build(self, B: int) -> None:
        group = self.optimizer.param_groups[0]
        old = self.trainer
        # local_only: under an initialised process group the reference's loop still trains every process on its own
        # (data parallel is TripletTrainer / run_training)
        tr = TripletTrainer(self.model, max_batch=B, max_seq_len=self.MAX_S, margin=self.margin, lr=float(group["lr"]),
                            betas=tuple(group["betas"]), eps=group["eps"], pack_roles=True, local_only=True)
        params = self.model.trainable_parameters()
        steps = set()
        for p, (_, off, shape) in zip(params, flat_param_slices(tr.D, tr.P)):
            n = p.numel()
            st = self.optimizer.state.get(p) or {}
            if st:  # moments of steps taken so far (module path, or a smaller engine)
                tr.exp_avg[off: off + n].copy_(st["exp_avg"].reshape(-1))
                tr.exp_avg_sq[off: off + n].copy_(st["exp_avg_sq"].reshape(-1))
                steps.add(int(st["step"]))
            self.optimizer.state[p] = {"step": _step_value(self.form, 0), "exp_avg": tr.exp_avg[off: off + n].view(shape),
                                       "exp_avg_sq": tr.exp_avg_sq[off: off + n].view(shape)}
        if len(steps) > 1:
            raise ValueError("train_epoch(fused): the optimizer's tensors have taken different numbers of steps")
        tr.step_count = steps.pop() if steps else 0
        self._sync_steps(tr)
        if old is not None:
            old.engine.close()
        self.trainer = tr

    def _grad_views(self):
        tr = self.trainer
        if self._bucket is None or self._bucket.numel() != tr.flat.numel():
            self._bucket = torch.zeros_like(tr.flat)
        return [self._bucket[off: off + p.numel()].view(shape)
                for p, (_, off, shape) in zip(self.model.trainable_parameters(), flat_param_slices(tr.D, tr.P))]

    def _take_left_over_grads(self) -> bool:
        """`.grad` tensors left by an earlier epoch (training.py:66-133 never clears what its last, unfinished window
        accumulated; training.py:36-53 leaves the last batch's) -> the trainer's bucket.  True if there were any."""
        params = self.model.trainable_parameters()
        if all(p.grad is None for p in params):
            return False
        for p, view in zip(params, self._grad_views()):
            if p.grad is None:
                view.zero_()
            elif p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
        return True

    def _leave_grads(self, have: bool) -> None:
        views = self._grad_views() if have else None
        for i, p in enumerate(self.model.trainable_parameters()):
            p.grad = views[i] if have else None

    def _sync_steps(self, tr) -> None:
        for p in self.model.trainable_parameters():
            self.optimizer.state[p]["step"] = _step_value(self.form, tr.step_count)

    @staticmethod
    def _stack(queries, positives, negatives) -> "TripletBatch":
        """(ids, mask) x 3 as the tokenizer returned them -> one batch of three role blocks.  Nothing is copied here: the
        engine packs the blocks (int64 -> int32 ids, attention masks -> prefix lengths, roles padded to the longest) in
        one pass on the gather stream (tt_engine_pool_roles).  Round 4 built [3, B, S] int64 id and mask stacks on the
        caller's stream instead — two copy kernels of ~10 us each in the projection chain — and the gather re-read the
        int64 mask on each of its column passes (17 us): 202-219 us per configs[1] step against 170-175 for packed ids.
        Integer / bool masks only.  A floating-point mask is refused like the module path refuses it (ops.gather_pool_norm:
        "attention_mask: expected int32 or int64"): to the reference's `_mean_pooling` (backend/model.py:63-72) it would be a
        WEIGHT mask, and truncating it to integers (round 5 did: 0.5 -> 0, 1.5 -> 1) pools something else."""
        blocks = []
        for ids, mask in (_ids_mask(x) for x in (queries, positives, negatives)):
            if mask is not None and mask.dtype not in (torch.int32, torch.int64):
                if mask.dtype.is_floating_point or mask.dtype.is_complex:
                    raise TypeError(f"attention_mask: expected an integer or bool 0/1 mask, got {mask.dtype} (a weight mask "
                                    "cannot be packed; the tokenizer's masks are int64, backend/model.py:43-48)")
                mask = mask.to(torch.int64)  # bool / uint8 / int8 / int16 masks
            blocks.append((ids.contiguous(), None if mask is None else mask.contiguous()))
        return TripletBatch(role_blocks=blocks)

    @staticmethod
    def _is_strings(x) -> bool:
        return isinstance(x, (list, tuple)) and len(x) > 0 and isinstance(x[0], str)

    def _text_batch(self, queries, positives, negatives) -> "TripletBatch":
        """Three lists of strings -> rows of the route's token store (textstore.py): a text is tokenised the first time it
        is seen (the towers' own tokenizer call, backend/model.py:43-45), then a batch is token_rows [3, B] int32."""
        import numpy as np

        from .data import TextColumn
        from .textstore import TextRowStore, hf_token_rows, is_hf_tokenizer

        if not (len(queries) == len(positives) == len(negatives)) or len(queries) == 0:
            raise ValueError("train_epoch(fused): queries, positives and negatives of a batch must have one length (>= 1)")
        if self.text_store is None:
            self.text_store = TextRowStore(self.model.query_tower.embedding_table().device, max_S=self.MAX_S)
        store = self.text_store
        rows = []
        for tower, texts in ((self.model.query_tower, queries), (self.model.document_tower, positives),
                             (self.model.document_tower, negatives)):
            tok = tower.tokenizer
            if is_hf_tokenizer(tok):  # the same ids without transformers' tensor conversion (textstore.hf_token_rows)
                call = lambda t, tok=tok: hf_token_rows(tok, t)
            else:
                call = lambda t, tok=tok: tok(t, padding=True, truncation=True, return_tensors="pt", max_length=512)
            if isinstance(texts, TextColumn) and texts.field in ("query", "positive"):
                rows.append(store.rows_of_column(texts, call))
            else:
                rows.append(store.rows_of_texts(id(tok), texts, call))
        # (the 24 KB row copy rides on the caller's stream; issuing it on the gather's stream instead was measured and is not
        #  faster: 210-213 vs 200-202 us per batch on boxes whose pre-tokenised step was 184-191 vs 183-187)
        host = torch.from_numpy(np.stack(rows).astype(np.int32)).pin_memory()
        return TripletBatch(token_rows=host.to(store.device, non_blocking=True), token_store=store.ids)

    def _use_text_cache(self) -> bool:
        from .textstore import per_text_tokenizer

        return (self.text_cache is not False and per_text_tokenizer(self.model.query_tower.tokenizer)
                and per_text_tokenizer(self.model.document_tower.tokenizer))

    def _batches(self, dataloader):
        dt, qt = self.model.document_tower, self.model.query_tower
        for queries, positives, negatives in dataloader:
            if self._is_strings(queries):
                if self._use_text_cache():
                    yield self._text_batch(queries, positives, negatives)
                    continue
                tq, tp, tn = qt.tokenize(queries), dt.tokenize(positives), dt.tokenize(negatives)  # model.py:43-48, three calls
                queries, positives, negatives = ((t["input_ids"], t.get("attention_mask")) for t in (tq, tp, tn))
            yield self._stack(queries, positives, negatives)

    def admit(self, dataloader, text_cache: Optional[bool]):
        """Look at the first batch before anything is enqueued or any state is touched: a mask the engine cannot pack, or
        `text_cache=True` with a tokenizer that cannot be cached, raises here.  Returns the same sequence (the first batch
        chained back).  A bad batch LATER in the epoch raises from run_epoch (nothing half-applied stays behind)."""
        import itertools

        self.text_cache = text_cache
        it = iter(dataloader)
        first = next(it, None)
        if first is None:
            return iter(())
        if self._is_strings(first[0]):
            if text_cache and not self._use_text_cache():
                raise ValueError("train_epoch(text_cache=True): the towers' tokenizer is not a per-text function "
                                 "(a transformers tokenizer, or a callable with tt_per_text = True)")
        else:
            self._stack(*first)
        return itertools.chain([first], it)

    def _rebind(self) -> None:
        """The nn.Parameters and `optimizer.state[p]` must still BE views of the trainer's flat buffers.  After
        `optimizer.load_state_dict()` (torch replaces the state tensors), a hand-assigned `p.data`, or a cleared state they
        are not: copy what the caller put there into the flat buffers and alias again, so that the fused epoch trains with
        the caller's weights and moments and not with stale ones (round-4 advisor finding)."""
        tr = self.trainer
        for p, (_, off, shape) in zip(self.model.trainable_parameters(), flat_param_slices(tr.D, tr.P)):
            n = p.numel()
            view = tr.flat[off: off + n].view(shape)
            if p.data.data_ptr() != view.data_ptr():
                view.copy_(p.data)
                p.data = view
            st = self.optimizer.state.get(p)
            if not st:  # a cleared optimizer state: zero moments, no steps taken (what torch does lazily)
                st = self.optimizer.state[p] = {"step": _step_value(self.form, 0)}
            for key, buf in (("exp_avg", tr.exp_avg), ("exp_avg_sq", tr.exp_avg_sq)):
                v = buf[off: off + n].view(shape)
                t = st.get(key)
                if t is None:
                    v.zero_()
                elif t.data_ptr() != v.data_ptr():
                    v.copy_(t.reshape(shape))
                st[key] = v
            st.setdefault("step", _step_value(self.form, 0))

    def run_epoch(self, dataloader, accumulation_steps: Optional[int] = None) -> float:
        """One epoch.  accumulation_steps None: train_epoch (training.py:36-53: zero_grad, backward, step per batch, a
        print every tenth batch).  An integer k: train_epoch_optimized (training.py:66-133: loss / k, step + zero_grad
        after every k-th batch of THIS epoch, whatever `.grad` held on entry is added to, whatever the last unfinished
        window accumulated stays in `.grad`).  A batch larger than the engine was built for drains the pipeline and
        rebuilds the engine (the reference's loop takes batches of any size); whatever raises inside the epoch — the
        dataloader, an over-long row — leaves no half-applied batches behind in the trainer."""
        it = self._batches(dataloader)
        first = next(it, None)
        if first is None:
            raise ZeroDivisionError("train_epoch: empty dataloader")  # what the reference's `total_loss / num_batches` raises
        self._check(first)
        if self.trainer is None or first.size > self.trainer.engine.max_B:
            self._build(first.size)
            if first.size * self._seq_len(first) >= 65536:  # a batch worth timing: pick the backward schedule / fence placement
                self.trainer.calibrate(first)             # for this device and shape (state-neutral, ~0.1 s once)
        self._rebind()
        accumulate = accumulation_steps is not None
        self._enter(accumulate, accumulation_steps)
        state = {"n": 0, "loss_sum": 0.0, "loss_n": 0}

        prints = collections.deque()  # (batch number, pinned host scalar, event): losses on their way to the host

        def flush(wait: bool) -> None:
            while prints and (wait or prints[0][2].query()):
                n, host, ev = prints.popleft()
                ev.synchronize()
                print(f"Batch {n}, Loss: {float(host):.4f}")

        def step(nxt) -> None:
            self.trainer.advance(nxt)
            state["n"] += 1
            if not accumulate and state["n"] % 10 == 0:
                # the reference prints this batch's loss here (training.py:59-61).  A synchronous read would drain the
                # two-deep pipeline every ten batches (measured ~10 us per step at configs[1]); the value is copied out
                # stream-ordered instead and printed as soon as it has arrived — the same lines in the same order,
                # a few batches later at most (all of them before the epoch returns)
                host = torch.empty(1, dtype=torch.float32, pin_memory=True)
                host.copy_(self.trainer.engine.last_loss(), non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                prints.append((state["n"], host, ev))
            if prints:
                flush(False)

        def grow(b) -> None:
            """`b` does not fit the engine: finish what is in flight, rebuild for b's size (weights, moments, step count,
            the un-applied gradient window and the epoch's loss statistics carry over) and go on."""
            old = self.trainer
            while old._pending:
                step(None)
            flush(True)
            ls, n, _ = old.engine.read_stats(reset=True)
            state["loss_sum"] += ls
            state["loss_n"] += n
            carried = (old._micro, old._carry)
            self._sync_steps(old)
            self._build(b.size)
            self._rebind()
            self._enter(accumulate, accumulation_steps, carried=carried)

        try:
            depth = self.trainer.SLOTS - 1 if self.trainer.side is not None else 1
            self.trainer.gather(first)
            nxt = next(it, None)
            while nxt is not None or self.trainer._pending:
                if nxt is not None:
                    self._check(nxt)
                    if nxt.size > self.trainer.engine.max_B:
                        grow(nxt)
                if nxt is not None and len(self.trainer._pending) < depth:  # prime (or re-prime) the pipeline
                    self.trainer.gather(nxt)
                    nxt = next(it, None)
                    continue
                if not self.trainer._pending:
                    self.trainer.gather(nxt)
                    nxt = next(it, None)
                    continue
                step(nxt)
                if nxt is not None:
                    nxt = next(it, None)
            tr = self.trainer
            flush(True)
            ls, n, _ = tr.engine.read_stats(reset=True)
            state["loss_sum"] += ls
            state["loss_n"] += n
            self._sync_steps(tr)
            if accumulate:
                self._leave_grads(tr._carry)
            else:  # `.grad` of the last batch, as the reference's loop leaves it (a later accumulation epoch adds to it)
                self._grad_views()
                tr.engine.reduce_grads(self._bucket, accumulate=False)
                self._leave_grads(True)
            return state["loss_sum"] / state["loss_n"]
        finally:
            tr = self.trainer
            if tr is not None:
                try:
                    if tr._pending:  # an exception mid-epoch: gathered-but-unapplied batches must not leak into the next call
                        tr.abort_pipeline()
                        self._sync_steps(tr)
                        try:
                            tr.engine.read_stats(reset=True)
                        except Exception:  # the flags it reports (weight mask, id out of range) must not replace the
                            pass           # exception that ended the epoch; the reset has happened either way
                finally:
                    tr.bucket, tr.accumulation_steps, tr._carry = None, 1, False

    def _enter(self, accumulate: bool, accumulation_steps, carried=None) -> None:
        """Hyper-parameters, step count and the accumulation window of this call -> the trainer.  `carried` = (micro, carry)
        of the engine this one replaces mid-epoch; None at the start of an epoch."""
        tr = self.trainer
        group = self.optimizer.param_groups[0]
        tr.lr, tr.betas, tr.eps, tr.margin = float(group["lr"]), tuple(group["betas"]), float(group["eps"]), self.margin
        steps = {int(self.optimizer.state[p]["step"]) for p in self.model.trainable_parameters()}
        if len(steps) != 1:
            raise ValueError("train_epoch(fused): the optimizer's tensors have taken different numbers of steps")
        tr.step_count = steps.pop()  # module-path steps taken with this optimizer in between count too
        tr.accumulation_steps = int(accumulation_steps) if accumulate else 1
        if carried is None:
            tr._micro = 0  # batch_idx restarts with the epoch (training.py:80)
            tr._carry = self._take_left_over_grads() if accumulate else False  # train_epoch clears .grad before its first batch
        else:
            tr._micro, tr._carry = carried  # the un-applied window lives in self._bucket, which outlives the engine
        if tr._carry or tr.accumulation_steps > 1:
            self._grad_views()
            tr.bucket = self._bucket
        else:
            tr.bucket = None  # Adam reads the gradient slabs directly
        tr.engine.read_stats(reset=True)

    @staticmethod
    def _seq_len(b) -> int:
        return b.token_store.shape[1] if b.role_blocks is None else max(i.shape[1] for i, _ in b.role_blocks)

    def _check(self, b) -> None:
        if b.role_blocks is None:  # rows of the text store: widths and lengths were checked when the texts were stored
            return
        S = max(i.shape[1] for i, _ in b.role_blocks)
        if S > self.MAX_S:
            raise ValueError(f"train_epoch(fused): rows of {S} tokens exceed {self.MAX_S} (the tokenizer's "
                             "max_length, backend/model.py:44)")
        if any(i.dim() != 2 or i.shape[0] != b.size for i, _ in b.role_blocks):
            raise ValueError("train_epoch(fused): queries, positives and negatives of a batch must have one length")


def train_epoch_optimized(model: TwoTowersModel, dataloader, criterion: TripletLoss,
                          optimizer: torch.optim.Optimizer, device=None, scaler=None,
                          accumulation_steps: int = 2, log_wandb: bool = True, *, fused: Optional[bool] = None,
                          text_cache: Optional[bool] = None) -> float:
    """Gradient-accumulation loop with the semantics of backend/training.py:66-133
    (`loss / accumulation_steps`, optimizer step + zero_grad every k-th batch,
    left-over gradients neither applied nor cleared at the end of the epoch), in
    fp32: the reference's fp16 autocast + GradScaler are not a parity target
    (SURVEY.md §2.2 O11), so `scaler` is accepted and ignored.
    `fused` as in train_epoch: None runs the batches on the fused engine when (model, criterion, optimizer) allow it —
    the un-applied gradients of an unfinished window are then views of the engine's bucket in `p.grad`."""
    model.train()
    route = _engine_route(model, criterion, optimizer, log_wandb) if fused in (None, True) else None
    if fused and route is None:
        raise ValueError("train_epoch_optimized(fused=True): needs TwoTowersModel(embed) on a GPU, TripletLoss and "
                         "torch.optim.Adam (default arithmetic) or FusedAdam")
    if route is not None:
        return route.run_epoch(route.admit(dataloader, text_cache), accumulation_steps=int(accumulation_steps))
    total_loss = 0.0
    num_batches = 0
    for batch_idx, (queries, positives, negatives) in enumerate(dataloader):
        query_embeds, positive_embeds, negative_embeds = _encode3(model, queries, positives, negatives)
        loss = criterion(query_embeds, positive_embeds, negative_embeds) / accumulation_steps
        loss.backward()
        if (batch_idx + 1) % accumulation_steps == 0:
            optimizer.step()
            optimizer.zero_grad()
        total_loss += loss.item() * accumulation_steps
        num_batches += 1
        if log_wandb and _HAVE_WANDB and batch_idx % 10 == 0:
            wandb.log({"batch_loss": loss.item() * accumulation_steps, "batch": num_batches})
    return total_loss / num_batches


# --------------------------------------------------------------------------- #
# run_training
# --------------------------------------------------------------------------- #
def run_training(num_epochs: int = 3, batch_size: int = 1024, learning_rate: float = 1e-4,
                 max_samples: int = 10000, projection_dim: int = 128, margin: float = 0.1,
                 project_name: str = "two-towers-retrieval", use_wandb: bool = True,
                 wandb_config: Optional[dict] = None, accumulation_steps: int = 2,
                 use_mixed_precision: bool = True, num_workers: int = 4,
                 run_comprehensive_test: bool = True, *, train_dataset=None, val_dataset=None, test_dataset=None,
                 tokenizer=None, pretrained_model=None, model_name: Optional[str] = None,
                 seed: Optional[int] = None, feature_cache: bool = True, sampler: str = "device",
                 encoder_mode: Optional[str] = None, calibrate: bool = True, three_encodes: bool = False,
                 encoder_autocast: Optional[str] = None) -> TwoTowersModel:
    """Same keyword surface and defaults as backend/training.py:383-397; builds the
    model / criterion / Adam exactly like :433-436 and runs `num_epochs` epochs on
    the fused engine.  Differences, all outside the hot path: the datasets are passed
    in (`train_dataset`, `val_dataset`, `test_dataset`: any map-style {"query_id","query","positive"}; the
    reference's MSMarcoDataset is a network fetch), wandb is not reproduced; the per-epoch validation NDCG@10
    (training.py:490) and the final comprehensive test (:505-521, `run_comprehensive_test`) run on the device
    (`evaluation.evaluate_model`) when their dataset is given; `use_mixed_precision` / `num_workers` are accepted for
    signature parity and ignored (fp32 is the parity path; tokenisation happens once, up front).
    `encoder_mode` None = the model's own default (`TwoTowersModel`: "bert" for a pretrained model with encoder layers —
    what the reference executes — else "embed"), so a default `run_training` and a default-constructed model that later
    loads its weights agree; the mode travels with the checkpoint (state_dict metadata) and a mismatch at load time
    warns.  "embed": word-embedding rows pooled directly (the HIP gather).  "bert": the
    reference's executed behaviour (frozen encoder between gather and pool, backend/model.py:51-52) — the encoder
    runs ONCE per text into the frozen-feature cache (it is frozen: its outputs never change), every epoch is then
    the fused cache-fed step, where the reference re-runs the encoder three times per step (99 % of its step
    time); needs feature_cache=True.  `sampler`: "device" draws the in-batch
    negatives with tensor ops (same distribution, no per-batch host work); "reference" reproduces
    backend/data.py:132-137's draws under `seed`, at its O(B^2) Python cost per batch.
    `encoder_autocast` ("fp16" | "bf16"; `bert` mode; NOT the parity path): the frozen encoder's one pass per text runs under
    `torch.autocast` — what `use_mixed_precision=True` means for the encoder in the reference's GPU loop (training.py:83);
    projections, loss and Adam stay fp32.  `use_mixed_precision` itself stays ignored so that a default call is the parity path."""
    from .data import TokenTripletLoader

    if train_dataset is None:
        raise ValueError("run_training: pass train_dataset= (MS MARCO loading is outside the hot path)")
    device = torch.device("cuda", torch.cuda.current_device())
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    if world > 1:
        # data parallel (launched one process per GPU, process group initialised by the caller): every rank builds
        # the SAME global batches — same loader seed — and trains on its contiguous shard of each
        s = torch.tensor([seed if seed is not None else random.getrandbits(60)], dtype=torch.int64, device=device)
        dist.broadcast(s, src=0)
        seed = int(s.item())
    kw = {}
    if model_name is not None:
        kw["model_name"] = model_name
    model = TwoTowersModel(projection_dim=projection_dim, tokenizer=tokenizer, pretrained_model=pretrained_model,
                           encoder_mode=encoder_mode, **kw)
    model = model.to(device)
    encoder_mode = model.query_tower.encoder_mode
    if max_samples and max_samples > 0 and len(train_dataset) > max_samples:
        train_dataset = torch.utils.data.Subset(train_dataset, range(max_samples))
    loader = TokenTripletLoader(train_dataset, model.query_tower.tokenizer, batch_size, device, seed=seed,
                                sampler=sampler, emit="rows3", shard=(rank, world))
    max_S = int(max(loader.q_ids.shape[1], loader.p_ids.shape[1]))
    trainer = TripletTrainer(model, max_batch=batch_size, max_seq_len=max_S, margin=margin, lr=learning_rate,
                             accumulation_steps=accumulation_steps)
    if encoder_mode != "embed" and not feature_cache:
        raise ValueError("run_training(encoder_mode='bert') runs on the frozen-feature cache: feature_cache=True "
                         "(or use train_epoch on the module path, which re-runs the encoder every step like the reference)")
    if feature_cache:
        # the encoder is frozen: pool every query / passage once, every epoch then only copies rows and
        # takes its negatives by in-batch index
        loader.emit = "cache"
        amp = {None: None, "fp16": torch.float16, "bf16": torch.bfloat16}[encoder_autocast]
        trainer.build_feature_cache(*_ids_lens(loader.all_texts()), encoder_autocast=(amp if encoder_mode != "embed" else None))
    else:
        # the token rows stay on the device and a batch names its texts by row; the negatives ARE
        # in-batch positives (backend/data.py:113-137), so by default they are taken by index from the positives'
        # embeddings (two gathers; same triplets, same numbers up to fp32 summation order) instead of being encoded
        # a third time (three_encodes=True: the reference's three encodes, training.py:40-42)
        loader.emit = "rows3" if three_encodes else "rows2"
        trainer.set_token_store(*loader.all_texts())
    sizes = loader.global_batch_sizes() if world > 1 else None  # the loader yields this rank's shard of every batch
    if calibrate and len(loader) > 0:
        trainer.calibrate(loader.peek())  # a first batch drawn WITHOUT advancing the loader's random streams
    if rank == 0:
        print(f"Starting training for {num_epochs} epochs...")
    for epoch in range(num_epochs):
        avg = trainer.train_epoch(loader, global_batch_sizes=sizes)
        if rank == 0:
            print(f"Epoch {epoch + 1}/{num_epochs}  avg_train_loss {avg:.6f}")
        if val_dataset is not None and rank == 0:  # per-epoch validation NDCG@10 (training.py:490)
            from .evaluation import evaluate_model

            val_ndcg = evaluate_model(model, val_dataset, batch_size=batch_size)
            print(f"Validation NDCG@10: {val_ndcg:.4f}")
    if test_dataset is not None and run_comprehensive_test and rank == 0:  # final comprehensive test (training.py:505-521)
        from .evaluation import evaluate_model

        evaluate_model(model, test_dataset, sample_size=len(test_dataset), min_query_groups=200, candidate_pool_size=-1,
                       comprehensive=True, log_wandb=False, wandb_prefix="final_", batch_size=batch_size)
    return model


def _ids_lens(t):
    ids, lens = t
    return ids, None, lens
