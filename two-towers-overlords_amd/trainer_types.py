"""Batch and configuration types of TripletTrainer, and the sharding helpers."""
from __future__ import annotations

import dataclasses
from typing import Optional, Tuple

import torch

from .model import TwoTowersModel

Batch = Tuple[torch.Tensor, Optional[torch.Tensor], Optional[torch.Tensor]]  # ids[3,B,S], mask[3,B,S]|None, lens[3,B]|None


@dataclasses.dataclass
class TripletBatch:
    """One batch for the fused engine.  Either token ids (`ids` [roles,B,S] + `mask` or `lens`),
    or rows of the trainer's token store (`token_rows` [roles,B] int32: the texts were tokenised
    once into `TripletTrainer.set_token_store`, a batch is 12 bytes per triplet), or rows of the
    trainer's frozen-feature cache (`cache_rows` [roles,B]).  roles = 3: query, positive,
    negative.  roles = 2 + `neg_index` (int32 [B]): the negatives are in-batch positives
    (backend/data.py:113-137), negative i = positive neg_index[i] — no third encode."""

    ids: Optional[torch.Tensor] = None
    mask: Optional[torch.Tensor] = None
    lens: Optional[torch.Tensor] = None
    neg_index: Optional[torch.Tensor] = None
    cache_rows: Optional[torch.Tensor] = None
    query_ids: Optional[torch.Tensor] = None  # int64 [B]; in-batch-negatives loss: same-query documents are not negatives
    token_rows: Optional[torch.Tensor] = None
    # recorded by the producer on ITS stream once the batch's tensors are complete (e.g. after the non_blocking
    # host->device copies on a copy stream); the gather then waits for it.  Not needed for tensors produced on the
    # stream that calls gather(): that stream is fenced anyway.
    ready_event: Optional["torch.cuda.Event"] = None
    # roles as SEPARATE (input_ids [B, S_r], attention_mask [B, S_r] | None) blocks, the tokenizer's own output per encode
    # call (backend/model.py:43-48): the engine packs them on the gather stream (tt_engine_pool_roles; single replica,
    # an engine built with pack_roles=True)
    role_blocks: Optional[list] = None
    # the store `token_rows` index, when it is not the trainer's own (set_token_store): ids [N, S] int32/int64 with
    # TT_PAD_ID behind each text's tokens — the drop-in route's growing text store hands every batch the tensor that was
    # current when the batch was made
    token_store: Optional[torch.Tensor] = None

    @property
    def size(self) -> int:
        for t in (self.ids, self.token_rows, self.cache_rows):
            if t is not None:
                return t.shape[1]
        if self.role_blocks is not None:
            return self.role_blocks[0][0].shape[0]
        raise ValueError("empty TripletBatch")

    def expand_negatives(self) -> "TripletBatch":
        """roles 2 + neg_index -> roles 3 (negatives materialised by a row gather)."""
        if self.neg_index is None:
            return self
        if self.role_blocks is not None:
            raise ValueError("role_blocks batches carry their negatives as a third block")
        ni = self.neg_index.long()
        cat = lambda t: None if t is None else torch.cat([t, t[1:2].index_select(1, ni)], dim=0).contiguous()
        return TripletBatch(cat(self.ids), cat(self.mask), cat(self.lens), None, cat(self.cache_rows), self.query_ids,
                            cat(self.token_rows), self.ready_event, token_store=self.token_store)


def as_batch(b) -> TripletBatch:
    return b if isinstance(b, TripletBatch) else TripletBatch(*b)


@dataclasses.dataclass
class TrainerConfig:
    """Every knob of TripletTrainer in one place (bench.py mirrors it: `--trainer FIELD=VALUE`).  The first block is the
    reference's hyper-parameters (backend/training.py:383-397, :436); the rest steer HOW the same arithmetic is run —
    none of them changes a result beyond fp32 summation order, and only `tuning["k1_form"]` does even that."""

    # -- what is computed (backend/training.py:433-436: TripletLoss(margin), Adam(lr), defaults of torch.optim.Adam)
    margin: float = 0.3
    lr: float = 1e-3
    betas: tuple = (0.9, 0.999)
    eps: float = 1e-8
    accumulation_steps: int = 1          # backend/training.py:66-133: loss / k, optimizer step every k-th batch of an epoch
    loss: str = "triplet"                # "triplet" (the reference's sampled negative) | "inbatch" (BASELINE.json configs[4])
    # -- how the two streams share the chip
    overlap_streams: Optional[bool] = None   # gathers of the next batches on a side stream (None: on a GPU with the HIP engine)
    k1_blocks_per_cu: Optional[int] = None   # tt_engine_config.k1_blocks_per_cu (None: TT_K1_SHARE_AUTO with a side stream)
    tuning: Optional[dict] = None            # tt_tuning fields for this trainer's engine (include/tt_hotpath.h)
    grad_slabs: int = 0                      # split-K slabs of the weight-gradient GEMMs (0: the engine's rule)
    event_scope: Optional[int] = None        # release scope of the cross-stream events: 1 no system fence (default), 2, 3
    fence_at_loss: Optional[bool] = None     # the gather's slot fence = the engine's loss event (None: calibrate() decides)
    join_carries_gather: bool = True         # the fork backward's join also carries the wait for the next gather
    use_graphs: Optional[bool] = None        # hipGraph replay of whole steps (opt-in; slower on ROCm 7.2)
    pack_roles: bool = False                 # engine staging for role_blocks batches (the drop-in train_epoch route)
    freeze_gc: bool = True                   # gc.freeze() once before the first epoch
    # -- data parallel
    exchange: Optional[bool] = None          # gradients are exchanged (None: iff more than one replica; True on one rank: tests)
    split_exchange: Optional[bool] = False   # the exchange in two parts (None: calibrate() times both forms on the fabric)
    gather_under_exchange: bool = True       # the next gather is issued right before the exchange
    local_only: bool = False                 # ignore an initialised process group: this process trains on its own (what the
    #                                          reference's train_epoch does under any launcher; the drop-in route sets it)


class _TorchEvent:
    """torch.cuda.Event behind the record(stream) / wait(stream) / query() interface of _lib.StreamEvent."""

    def __init__(self):
        self.ev = torch.cuda.Event()

    def record(self, stream) -> None:
        self.ev.record(stream)

    def wait(self, stream) -> None:
        stream.wait_event(self.ev)

    def query(self) -> bool:
        return self.ev.query()


def shard_sizes(global_B: int, world: int):
    """Contiguous split of `global_B` rows over `world` ranks: (sizes, offsets); the first
    global_B % world ranks hold one row more."""
    base, rem = divmod(int(global_B), int(world))
    sizes = [base + (1 if r < rem else 0) for r in range(world)]
    offs = [r * base + min(r, rem) for r in range(world)]
    return sizes, offs


def shard_batch(batch: Batch, rank: int, world: int) -> Tuple[Batch, int]:
    """Contiguous split of a global batch over `world` ranks; the
    last ranks get one triplet fewer when B % world != 0 (none at all when B < world: such a
    rank still takes part in the step's collectives).  Returns (shard, B_global)."""
    tb = as_batch(batch).expand_negatives() if world > 1 else as_batch(batch)
    B = tb.size
    sizes, offs = shard_sizes(B, world)
    lo, hi = offs[rank], offs[rank] + sizes[rank]
    cut = lambda t: None if t is None else t[:, lo:hi].contiguous()
    if isinstance(batch, TripletBatch):
        ni = tb.neg_index if world == 1 else None
        qid = None if tb.query_ids is None else tb.query_ids[lo:hi].contiguous()  # [B], not [roles, B]
        return TripletBatch(cut(tb.ids), cut(tb.mask), cut(tb.lens), ni, cut(tb.cache_rows), qid, cut(tb.token_rows),
                            tb.ready_event), B
    return (cut(tb.ids), cut(tb.mask), cut(tb.lens)), B


def _require_same_frozen_encoder(model: TwoTowersModel) -> None:
    """One feature cache serves both towers only if they hold the same frozen encoder — the reference loads the same
    pretrained model into each tower (backend/model.py:84-85), this package shares one object by default."""
    a, b = model.query_tower.pretrained_model, model.document_tower.pretrained_model
    if a is b:
        return
    sa, sb = a.state_dict(), b.state_dict()
    if sa.keys() != sb.keys() or any(not torch.equal(sa[k], sb[k]) for k in sa):
        raise ValueError("the two towers hold different frozen encoders: one feature cache cannot serve both")


def _proj_dim(params: dict) -> int:
    return params["query_tower.projection.0.bias"].shape[0]
