"""TripletTrainer — the batch body of backend/training.py:36-53 on the fused engine (flat parameters, no autograd graph,
device-side loss accumulation).  This file holds the pipeline (slots, streams, gather / apply / advance), the epoch loop and
the checkpoint; its batch / config types and sharding helpers live in trainer_types.py, calibration in
trainer_calibrate.py, the replicas' gradient exchange and the in-batch loss in trainer_exchange.py, hipGraph replay in
trainer_graphs.py.  The reference-signature loops that route here live in training.py."""
from __future__ import annotations

import collections
import dataclasses
import gc
import math
from typing import Iterable, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import _lib, ops
from .engine import HotPathEngine, flat_param_slices
from .model import TwoTowersModel
from .trainer_calibrate import CalibrationMixin
from .trainer_exchange import ExchangeMixin
from .trainer_graphs import GraphReplayMixin
from .trainer_types import (Batch, TrainerConfig, TripletBatch, _proj_dim, _require_same_frozen_encoder, _TorchEvent,  # noqa: F401
                            as_batch, shard_batch, shard_sizes)


class TripletTrainer(CalibrationMixin, ExchangeMixin, GraphReplayMixin):
    """The batch body of backend/training.py:36-53 on the fused engine.

    Data parallel: every rank holds the frozen table and the 8
    trainable tensors, takes its contiguous shard of the global batch, scales its
    hinge gradients by 1/B_global, sums the flat gradient bucket with ONE
    all-reduce per optimizer step and applies the identical Adam update.  The
    all-reduce is asynchronous and the next batch's gather+pool (which does not
    depend on the weights) is issued under it.

    `engine` may be injected (tests drive the data-parallel logic on CPU/gloo with
    an oracle-backed stand-in); the product default is the HIP engine.
    """

    _gc_frozen = False  # gc.freeze() is process-wide: done once
    SLOTS = 3  # == TT_ENGINE_SLOTS (include/tt_hotpath.h)

    def __init__(self, model: Optional[TwoTowersModel] = None, *, table: Optional[torch.Tensor] = None,
                 params: Optional[dict] = None, max_batch: int, max_seq_len: int, config: Optional["TrainerConfig"] = None,
                 process_group=None, engine=None, **overrides):
        """`config`: a TrainerConfig (every knob with its default and its meaning in one place); any of its fields may
        also be given by keyword (`TripletTrainer(model, max_batch=..., max_seq_len=..., lr=1e-4, loss="inbatch")`),
        which overrides the config's value.  `engine`: an injected engine (tests); `process_group`: the replicas."""
        cfg = dataclasses.replace(config or TrainerConfig(), **overrides)  # an unknown keyword raises TypeError, like a signature
        self.config = cfg
        margin, lr, betas, eps, accumulation_steps, loss = cfg.margin, cfg.lr, cfg.betas, cfg.eps, cfg.accumulation_steps, cfg.loss
        overlap_streams, k1_blocks_per_cu, use_graphs, freeze_gc = cfg.overlap_streams, cfg.k1_blocks_per_cu, cfg.use_graphs, cfg.freeze_gc
        exchange, split_exchange, gather_under_exchange = cfg.exchange, cfg.split_exchange, cfg.gather_under_exchange
        tuning, event_scope, fence_at_loss, grad_slabs = cfg.tuning, cfg.event_scope, cfg.fence_at_loss, cfg.grad_slabs
        join_carries_gather, pack_roles = cfg.join_carries_gather, cfg.pack_roles
        local_only = bool(cfg.local_only)
        if model is not None:
            # `embed`: K1 gathers from this table.  `bert` (backend/model.py:51-52 as executed: the frozen encoder
            # between gather and pool): the encoder is not part of the fused step — its pooled outputs are computed
            # once per text by build_feature_cache() (stock PyTorch ops + the dense pool kernel) and the step is fed
            # from that cache; the table is bound only because the engine wants one.
            encoder_mode = model.query_tower.encoder_mode
            table = model.query_tower.embedding_table().detach()
            tensors = [p.detach() for p in model.trainable_parameters()]
        else:
            if table is None or params is None:
                raise ValueError("pass a TwoTowersModel or table= and params=")
            encoder_mode = "embed"
            tensors = [params[name] for name, _, _ in flat_param_slices(table.shape[1], _proj_dim(params))]
        self.encoder_mode = encoder_mode
        if loss not in ("triplet", "inbatch"):
            raise ValueError("loss must be 'triplet' (the reference's sampled-negative triplet loss) or 'inbatch'")
        self.loss_kind = loss
        self.freeze_gc = bool(freeze_gc)  # gc.freeze() once before the first epoch (see train_epoch)
        self.device = table.device
        self.D = table.shape[1]
        self.P = tensors[1].shape[0]
        self.margin, self.lr, self.betas, self.eps = float(margin), float(lr), betas, float(eps)
        self.accumulation_steps = int(accumulation_steps)
        # Default on a GPU: the gathers of the next batches run on a side stream under this batch's
        # projection kernels; how the two share a CU is steered through the LDS a gather workgroup occupies
        # (TT_K1_SHARE_AUTO: the engine pairs the K1 form with the projection tile, docs/LOG_r04.md §5): +20 %
        # triplets/s over running them back to back.
        if overlap_streams is None:
            overlap_streams = self.device.type == "cuda" and engine is None
        if k1_blocks_per_cu is None:
            k1_blocks_per_cu = _lib.TT_K1_SHARE_AUTO if overlap_streams else 0
        # `tuning`: tt_tuning fields for THIS trainer's engine (include/tt_hotpath.h; K1 form, tiles, ...): per handle,
        # nothing process-wide — two trainers in one process may differ
        self.engine = engine if engine is not None else HotPathEngine(table.contiguous(), self.P, max_batch, max_seq_len,
                                                                      k1_blocks_per_cu=k1_blocks_per_cu, tuning=tuning,
                                                                      grad_slabs=grad_slabs, pack_roles=pack_roles)
        self.engine_k1_cap = int(k1_blocks_per_cu or 0)
        n = self.engine.n_params
        self.flat = torch.cat([t.reshape(-1).float() for t in tensors]).to(self.device).contiguous()
        assert self.flat.numel() == n
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.engine.bind(self.flat, self.exp_avg, self.exp_avg_sq)
        if model is not None:  # the modules now alias the flat buffer: state_dict() sees the trained weights
            for p, (_, off, shape) in zip(model.trainable_parameters(), flat_param_slices(self.D, self.P)):
                p.data = self.flat[off: off + p.numel()].view(shape)
        self.model = model
        self.pg = process_group
        self.world = (dist.get_world_size(process_group)
                      if dist.is_available() and dist.is_initialized() and not local_only else 1)
        self.rank = dist.get_rank(process_group) if self.world > 1 else 0
        # gradients are exchanged between replicas (see _all_reduce); `exchange=True` forces the data-parallel step's
        # code path on a single-rank group too (tests: the exchange is then an identity)
        if self.world > 1 and exchange is False:
            raise ValueError("exchange=False with more than one replica: the gradients must be summed over the ranks")
        self._exchange = (self.world > 1) if exchange is None else bool(exchange)
        self._gather_under_exchange = bool(gather_under_exchange)  # False: the next gather goes ahead of the projection kernels (A/B)
        self.bucket = torch.empty_like(self.flat) if (self._exchange or self.accumulation_steps > 1) else None
        # split_exchange=None: calibrate() times the data-parallel step with the exchange in one bucket and in two parts on
        # the fabric at hand and keeps the faster (every replica takes the slowest rank's verdict).
        # split_exchange=True: the exchange in two parts — the second-layer gradients (dW2, db2 of both
        # towers: complete before the dH -> dW1 chain starts) are reduced and all-reduced on a second stream UNDER that
        # chain, only the first-layer gradients travel after it (see _split_step).  Same sums per element, so with two
        # replicas bitwise the one-bucket step; off by default until RCCL hardware says which is faster.
        can_split = self._exchange and hasattr(self.engine, "reduce_grads_range")
        self._split = bool(split_exchange) and can_split
        self._split_auto = split_exchange is None and can_split and self.device.type == "cuda"  # calibrate() times both forms
        if self._split or self._split_auto:
            tn = self.engine.n_params // 2                      # floats per tower: W1 | b1 | W2 | b2
            n1 = self.P * self.D + self.P                       # W1 | b1
            self._late = [(0, n1), (tn, n1)]
            self._early = [(n1, tn - n1), (tn + n1, tn - n1)]   # W2 | b2 of the query tower, of the document tower
            self._comm = self._ev_dw2 = self._ev_early = None
            if self.device.type == "cuda":
                self._comm = torch.cuda.Stream(self.device)
                self._ev_dw2, self._ev_early = torch.cuda.Event(), torch.cuda.Event()
                self.engine.set_dw2_event(self._ev_dw2)
        self.step_count = 0        # optimizer steps taken
        # backend/training.py:80-102: the optimizer steps when (batch_idx + 1) % k == 0 and batch_idx restarts with every
        # epoch, while the gradient of an epoch's unpaired last batch stays in .grad — so the next epoch's first step
        # applies k + 1 micro-gradients.  `_micro` is that batch index (train_epoch() resets it), `_carry` says that
        # the bucket holds gradients no optimizer step has consumed yet (pinned by tests/golden/step_accum.npz).
        self._micro = 0
        self._carry = False
        if self.world > 1:
            self.sync_replicas()   # replicas start from rank 0's weights (what DDP's constructor broadcast does)
            # one all-reduce of the step's exact shape now, so that whatever the backend sets up lazily at the first use
            # of a collective / message size (connections, buffers, algorithm choice) is not paid by the first step
            self.bucket.zero_()
            self._all_reduce(self.bucket)
        self.side = torch.cuda.Stream(self.device) if (overlap_streams and self.device.type == "cuda") else None
        # gathered, not yet applied: (slot, ready event, local B, neg_index, query_ids)
        self._pending = collections.deque()
        self._next_slot = 0
        self._apply_stream = None             # the stream the last apply() ran phase B on
        if self.side is not None:             # events are reused: one "fence" and one "ready" per slot
            # The events that order the side stream and the caller's stream against each other are same-device events:
            # created without the system-scope fence a default hipEvent performs when it is recorded (a cache write-back
            # that the next kernel of the recording stream waits for — the fence sits between Adam and the next batch's
            # first projection kernel, on the critical chain: 179.5 -> 176.3 us per configs[1] step, 150.3 -> 145.5 at
            # configs[2]'s share).  event_scope (= tt_tuning.event_scope): 1 no system fence, 2 device-scope release,
            # 3 torch's default events; None / 0: the engine tuning's, else 1.
            if not event_scope:
                event_scope = int((tuning or {}).get("event_scope", 0)) or 1
            self.event_scope = int(event_scope)
            mk = (lambda: _TorchEvent()) if self.event_scope == 3 else (lambda: _lib.StreamEvent(self.event_scope))
            self._slot_fence = [mk() for _ in range(self.SLOTS)]
            self._slot_fence2 = [mk() for _ in range(self.SLOTS)]
            self._slot_ready = [mk() for _ in range(self.SLOTS)]
            # fence_at_loss: the gather of batch i + 2 may overwrite its pooled slot once phase B of batch i - 1 has
            # finished with it.  Instead of recording a fence event of its own on the caller's stream at the start of
            # step i (an event packet between Adam and the first projection kernel of the next batch: the critical
            # chain), the side stream may wait for the event the engine records right after batch i's loss kernel anyway —
            # in the fork backward that record IS the fork point — which lies behind all of batch i - 1 in stream order.
            # The gather then starts ~half a step later, beside the backward instead of the forward: measured
            # (profiles/r04_fence_at_loss_ab.txt) configs[1] 175.3 -> 173.4 us, in-batch loss B = 4096 619.5 -> 591.1 us,
            # but configs[2]'s share 142.5 -> 148.3 us (its 65 us gather is better spent beside the forward), and with a
            # gradient exchange the gather belongs UNDER the exchange (207.8 vs 186.7 us at a 60 us stand-in).  So: never
            # with an exchange; None = on for the in-batch loss, and for the triplet loss whatever calibrate() measures
            # on the device at hand (off until then).
            self._loss_event = None
            self._use_loss_fence = False
            # (not with a gradient exchange: measured with the 60 us stand-in, configs[1] 220.9 -> 228.3 us, configs[2]'s
            # share 187.4 -> 189.1 — there the gather is issued under the exchange and is still running at the join)
            self._join_carries_gather = (join_carries_gather and self.event_scope != 3 and not self._exchange
                                         and hasattr(self.engine, "set_join_wait"))
            if fence_at_loss is not False and self.event_scope != 3 and hasattr(self.engine, "set_loss_event") and not self._exchange:
                self._loss_event = _lib.StreamEvent(self.event_scope)
                self.engine.set_loss_event(self._loss_event)
                self._use_loss_fence = bool(fence_at_loss) or (fence_at_loss is None and self.loss_kind == "inbatch")
            self._fence_arg = fence_at_loss
            # tt_engine_create cleared the workspace on the main stream: order the side stream behind it
            self.side.wait_stream(torch.cuda.current_stream(self.device))
        else:
            self._loss_event, self._use_loss_fence, self._fence_arg = None, False, fence_at_loss
            self._join_carries_gather = False
        self._loss_armed = self._loss_event is not None  # tt_engine_arm_loss_event's current value (set_loss_event arms)
        self._prep = self._prep_fence = None  # third stream for the role-packing pass of role_blocks batches (see _timed_pool)
        if pack_roles and self.side is not None and self.event_scope != 3:
            self._prep = torch.cuda.Stream(self.device)
            self._pack_done = [_lib.StreamEvent(self.event_scope) for _ in range(self.SLOTS)]
            self._prep.wait_stream(torch.cuda.current_stream(self.device))  # behind tt_engine_create's workspace memset
        # With a gradient exchange the next gather sits UNDER the exchange and the chain runs alone on the chip: there the
        # fused backward (one stream, no fork / join events) is the shorter one — the fork's events cost 11 + 7 us of gaps
        # at configs[2]'s share (gpurun_out/r05_base/timeline_c3_x60.txt) and nothing shares the chip to fill them.  The
        # engine's own AUTO rule decides for a chain that shares the chip with gathers; calibrate() re-measures both in
        # the regime at hand and may override this.
        if (self._exchange and self._gather_under_exchange and self.side is not None and hasattr(self.engine, "set_backward")
                and not (tuning or {}).get("backward")):
            self.engine.set_backward("fused")
        self._prewaited = None                # slot whose "ready" event the caller's stream already waits for
        self.k1_events = None                 # set to a list to collect (start, end) HIP events of K1 launches
        self.k1_event_every = 4               # ... of every 4th launch
        self.k1_event_pool = []               # optional pre-created (start, end) pairs for those
        self._k1_count = 0
        self.feature_cache = None             # [N_texts, D] pooled-normalised vectors (build_feature_cache)
        self.token_store = None               # (ids [N_texts, S], lens [N_texts] | None) behind TripletBatch.token_rows
        self.calibration = None               # what calibrate() measured and chose
        self.step_trace = None                # set to a list to collect per-step timing events (see _trace_mark)
        self._dry = False                     # calibrate(): apply() stops after fwd + bwd (+ reduce + exchange under DP), no optimizer
        self._dry_bucket = None               # scratch bucket of those dry data-parallel steps
        self.table = table
        # hipGraph replay of whole steps (single replica, no accumulation; opt-in).  Measured on
        # ROCm 7.2 / MI355X it does NOT pay: replaying the 3-stream, ~15-node step graph costs the
        # host ~280 us per replay (vs ~70 us for the same launches issued one by one), so the
        # step becomes host-bound (289 vs 233 us).  Kept for runtimes whose graph launch is O(1).
        self.use_graphs = bool(use_graphs) and self.side is not None and self.world == 1 and self.accumulation_steps == 1
        self.eager_every = None               # if set (bench), every n-th step is run eagerly (for K1 event timing)
        self._graphs = {}
        self._ring_ids = self._ring_lens = None
        self._ring_fill = [None] * self.SLOTS
        self._steps_done = 0
        # device-side Adam step counter only where replay needs it: its one-thread prep kernel
        # (two double pow) sits on the critical chain and costs ~20 us per step
        self._auto_step = self.use_graphs and self.bucket is None and hasattr(self.engine, "adam_auto")
        if self._auto_step:
            self.engine.set_step(0)

    # -- lifetime ---------------------------------------------------------------
    def close(self) -> None:
        """Release the captured step graphs BEFORE the engine.  A hipGraph captured by `_graph()` holds nodes that ran on
        the engine's own second stream and on its events; attribute teardown on a plain refcount drop destroys `engine`
        (assigned first) before `_graphs` — the graph execs then outlive a stream they reference, and destroying them
        afterwards corrupted the HIP runtime's heap: the NEXT trainer's first graph launch crashed in
        hip::Graph::UpdateStreams, depending on what else had been allocated and freed in the process (round 6,
        tools/r06_graph_probe2.py)."""
        graphs = self.__dict__.get("_graphs")
        if graphs:
            if self.device.type == "cuda" and not torch.cuda.is_current_stream_capturing():
                torch.cuda.synchronize(self.device)
            graphs.clear()
        eng = self.__dict__.get("engine")
        if eng is not None and hasattr(eng, "close"):
            eng.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers ---------------------------------------------------------------
    def named_parameters(self) -> dict:
        return {name: self.flat[off: off + int(math.prod(shape))].view(shape)
                for name, off, shape in flat_param_slices(self.D, self.P)}

    # -- pipeline: gathers run ahead of the weight-dependent part ---------------------
    def gather(self, batch: Batch, fence=None) -> None:
        """Enqueue K1 (gather + pool + normalise) for a batch that a later `apply()` will
        consume.  Weight-independent, so it may run up to SLOTS-1 batches ahead; with a side
        stream it runs concurrently with the projection kernels of earlier batches.
        `fence` (apply() passes it to `issue_next`): an event already recorded on the caller's stream behind
        everything this gather must wait for — the batch's tensors were produced before that record and the phase B
        that last read the slot lies before it; the side stream then waits for IT and no event is recorded here."""
        if len(self._pending) >= self.SLOTS:
            raise RuntimeError("gather(): all pooled-input slots are in use; call apply() first")
        slot = self._next_slot
        self._next_slot = (slot + 1) % self.SLOTS
        batch = as_batch(batch)
        if batch.neg_index is not None and self.world > 1:
            batch = batch.expand_negatives()  # an in-batch index does not survive sharding across ranks
        if batch.size == 0:  # an empty shard (global batch smaller than the world): nothing to pool
            self._pending.append((slot, None, 0, None, batch.query_ids))
            return
        if self.side is None:
            self._timed_pool(slot, batch, None)
            self._pending.append((slot, None, batch.size, batch.neg_index, batch.query_ids))
            return
        # Order the side stream behind everything the caller's stream holds at this point: (a) the kernels or
        # copies that PRODUCE this batch's ids / mask / lens / rows (a loader's index ops, a non_blocking
        # H2D copy), (b) the fwd_bwd that last read this pooled-input slot (enqueued earlier on the same
        # stream), (c) graph replays.  K1 runs two batches ahead, so this wait is off the critical path.
        cur = torch.cuda.current_stream(self.device)
        if fence is not None and (self._apply_stream is None or cur == self._apply_stream):
            fence.wait(self.side)  # recorded by the engine behind (a), (b) and (c): nothing to record here
        else:
            fence = self._slot_fence[slot]
            fence.record(cur)
            fence.wait(self.side)
            if self._apply_stream is not None and cur != self._apply_stream:
                # the batch was produced on a stream of its own (a copy stream): (b) lives on the stream apply() uses
                fence2 = self._slot_fence2[slot]
                fence2.record(self._apply_stream)
                fence2.wait(self.side)
        # the packing pass of a role_blocks batch may run on the third stream behind this ONE fence — not when the order
        # also hangs on a second fence or on the producer's own event
        self._prep_fence = fence if (batch.ready_event is None and
                                     (self._apply_stream is None or cur == self._apply_stream)) else None
        if batch.ready_event is not None:  # produced on a stream of its own (a copy stream)
            self.side.wait_event(batch.ready_event)
        # K1 reads these tensors on the side stream: tell the caching allocator, so that memory the producer
        # drops right after this call is not handed out again before K1 has finished with it
        for t in (batch.ids, batch.mask, batch.lens, batch.cache_rows, batch.token_rows, batch.token_store,
                  *(x for blk in (batch.role_blocks or ()) for x in blk)):
            if t is not None:
                t.record_stream(self.side)
                if self._prep is not None and batch.role_blocks is not None:
                    t.record_stream(self._prep)
        # launch on the side stream by handle (no stream context switch on the host)
        self._timed_pool(slot, batch, self.side)
        ev = self._slot_ready[slot]
        ev.record(self.side)
        self._pending.append((slot, ev, batch.size, batch.neg_index, batch.query_ids))

    def _timed_pool(self, slot, batch: TripletBatch, stream) -> None:
        timed = False
        if self.k1_events is not None:
            self._k1_count += 1
            timed = self._k1_count % self.k1_event_every == 0  # events cost ~5 us of stream time each
        raw = None if stream is None else stream.cuda_stream
        if timed:
            pool = self.k1_event_pool  # events created ahead of time (bench): no hipEventCreate inside a timed region
            e0, e1 = pool.pop() if pool else (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            e0.record(stream)
        if hasattr(self.engine, "set_sharing"):  # a row copy from the cache does not compete with phase B
            self.engine.set_sharing(batch.cache_rows is None and self.side is not None)
        if batch.cache_rows is None and self.encoder_mode != "embed":
            raise RuntimeError("`bert` encoder mode: the frozen encoder is not part of the fused step; call "
                               "build_feature_cache() once and feed cache_rows batches (TokenTripletLoader emit='cache')")
        if batch.cache_rows is not None:
            if self.feature_cache is None:
                raise RuntimeError("batch carries cache_rows but build_feature_cache() has not been called")
            self.engine.pool_cached(slot, self.feature_cache, batch.cache_rows, stream=raw)
        elif batch.role_blocks is not None:
            if self.world > 1:
                raise RuntimeError("role_blocks batches are not sharded: hand a data-parallel trainer packed ids")
            if self._prep is not None and stream is not None and self._prep_fence is not None:
                # The gather stream is the busier one (it runs K1 back to back): the packing pass goes on a third stream,
                # ordered behind the same fence as the gather, and the gather waits for it — by then it has long finished
                # (the gather of the batch before is still running when this one is enqueued)
                self._prep_fence.wait(self._prep)
                S = self.engine.pack_roles(slot, batch.role_blocks, stream=self._prep.cuda_stream)
                ev = self._pack_done[slot]
                ev.record(self._prep)
                ev.wait(stream)
                self.engine.pool_packed(slot, len(batch.role_blocks), batch.size, S, stream=raw)
            else:
                self.engine.pool_roles(slot, batch.role_blocks, stream=raw)
        elif batch.token_rows is not None:
            if batch.token_store is not None:
                self.engine.pool_rows(slot, batch.token_store, None, batch.token_rows, stream=raw)
            elif self.token_store is None:
                raise RuntimeError("batch carries token_rows but set_token_store() has not been called")
            else:
                self.engine.pool_rows(slot, self.token_store[0], self.token_store[1], batch.token_rows, stream=raw)
        else:
            self.engine.pool(slot, batch.ids, batch.mask, batch.lens, stream=raw)
        if timed:
            e1.record(stream)
            self.k1_events.append((e0, e1))

    def set_token_store(self, ids: torch.Tensor, lens: Optional[torch.Tensor] = None) -> None:
        """The collection's texts, tokenised once: ids [N, S] int32/int64 zero-padded rows,
        lens [N] int32 prefix lengths (None: full length), resident on the device.  Batches then name their
        texts by row (`TripletBatch.token_rows`) and K1 reads the ids through that indirection."""
        if ids.dim() != 2 or ids.device != self.device or ids.dtype not in (torch.int32, torch.int64):
            raise ValueError("set_token_store: ids must be an int32 / int64 [N, S] tensor on the trainer's device")
        if lens is not None and (lens.dim() != 1 or lens.shape[0] != ids.shape[0] or lens.dtype.is_floating_point):
            raise ValueError("set_token_store: lens must be an integer [N] tensor, one prefix length per row of ids")
        if ids.shape[1] > self.engine.max_S:
            raise ValueError(f"set_token_store: rows of {ids.shape[1]} tokens exceed the engine's max_seq_len")
        if lens is not None:
            # The lengths go INTO the rows: tokens behind a text's length become TT_PAD_ID (weight 0 in every gather form,
            # include/tt_hotpath.h), and the gather runs without a per-text length load — beside the projection chain that
            # dependent load cost the column-sliced gather 12 us per configs[1] launch (profiles/r05_module_path.txt); a whole
            # epoch through the loader: 10.65 -> 11.04 M triplets/s (192.3 -> 185.5 us per step, same box, interleaved).
            pad = -(2 ** 31) if ids.dtype == torch.int32 else -(2 ** 63)
            keep = torch.arange(ids.shape[1], device=ids.device).unsqueeze(0) < lens.to(ids.device).unsqueeze(1)
            ids = torch.where(keep, ids, torch.full((), pad, dtype=ids.dtype, device=ids.device))
        self.token_store = (ids.contiguous(), None)

    @torch.no_grad()
    def build_feature_cache(self, ids: torch.Tensor, mask: Optional[torch.Tensor] = None,
                            lens: Optional[torch.Tensor] = None, chunk: Optional[int] = None,
                            encoder_autocast: Optional[torch.dtype] = None) -> torch.Tensor:
        """Frozen-feature cache: the encoder is frozen (backend/model.py:29-30,51),
        so every text's pooled-normalised vector is computed ONCE here and later batches only copy rows
        (`TripletBatch.cache_rows`).  `embed` mode: K1 over ids [N,S].  `bert` mode: the frozen encoder
        (stock PyTorch-ROCm ops, `chunk` texts at a time) + the dense pool kernel — what the reference
        recomputes three times per step in every epoch (training.py:40-42) runs once per text, and every
        epoch after that is the fused cache-fed step.  Returns the [N, D] cache.
        `encoder_autocast` (`bert` mode; NOT the parity path): run the frozen encoder under `torch.autocast` in this dtype
        (torch.float16 is what the reference's GPU loop does, backend/training.py:83; torch.bfloat16) — its hidden states are
        pooled in fp32 as always, but they carry half-precision rounding (~1e-3): off by default."""
        N = ids.shape[0]
        out = torch.empty((N, self.D), dtype=torch.float32, device=self.device)
        bert = self.encoder_mode != "embed"
        if bert:
            if self.model is None:
                raise RuntimeError("`bert` mode needs the TwoTowersModel the trainer was built from")
            _require_same_frozen_encoder(self.model)
            if mask is None and lens is not None:
                mask = (torch.arange(ids.shape[1], device=ids.device).unsqueeze(0) < lens.unsqueeze(1)).to(torch.int64)
        chunk = chunk or (512 if bert else 65536)
        if bert:
            # encoder cost grows with the padded length: walk the texts in length order, each chunk cut to its own
            # longest row (queries are a tenth of a passage) — the per-chunk read-back is set-up cost, once per run
            if mask is not None:
                order = mask.sum(1).argsort()
            else:
                order = torch.arange(N, device=ids.device)
            # the cache holds the EVAL-mode encoder's outputs (what backend/training.py:26-63, the reference's CPU
            # loop, sees: from_pretrained leaves the encoder in eval and train_epoch never calls model.train()); a
            # caller that has switched the model to train() would otherwise bake one sample of dropout noise into it
            pm = self.model.query_tower.pretrained_model
            was_training = pm.training
            pm.eval()
            try:
                for s in range(0, N, chunk):
                    sel = order[s: s + chunk]
                    cut = ids.shape[1] if mask is None else max(int(mask[sel].sum(1).max()), 1)
                    tokens = {"input_ids": ids[sel][:, :cut].long().contiguous()}
                    if mask is not None:
                        tokens["attention_mask"] = mask[sel][:, :cut].contiguous()
                    if encoder_autocast is not None:
                        with torch.autocast(self.device.type, dtype=encoder_autocast):
                            out[sel] = self.model.query_tower.pool_tokens(tokens)
                    else:
                        out[sel] = self.model.query_tower.pool_tokens(tokens)
            finally:
                pm.train(was_training)
        else:
            for s in range(0, N, chunk):
                e = min(N, s + chunk)
                ops.gather_pool_norm(self.table, ids[s:e].contiguous(), None if mask is None else mask[s:e].contiguous(),
                                     None if lens is None else lens[s:e].contiguous(), out=out[s:e])
        self.feature_cache = out
        return out

    def apply(self, global_B: Optional[int] = None, issue_next=None) -> None:
        """fwd + bwd of the oldest gathered batch, gradient exchange, optimizer.
        `issue_next(fence)` (optional) is called where enqueueing further gathers overlaps best — it should call
        `gather(next_batch, fence=fence)` for a batch whose tensors were produced BEFORE this call: single replica:
        right behind this batch's forward + backward launches (the side stream waits for the engine's loss event,
        see __init__), or ahead of them when that event is not in use; with a gradient exchange: right before the
        all-reduce."""
        slot, ev, local_B, neg_index, query_ids = self._pending.popleft()
        global_B = global_B or local_B * self.world
        if global_B <= 0:
            raise ValueError("apply(): empty global batch")
        k = self.accumulation_steps
        self._steps_done += 1
        mark = self._trace_mark if self.step_trace is not None else None
        if mark:
            mark(0)  # step start (caller's stream)
        loss_fence = self._loss_event if (self.side is not None and self._use_loss_fence) else None
        if self._loss_event is not None and (loss_fence is not None) != self._loss_armed:
            self._loss_armed = loss_fence is not None   # the fused backward records the event only when someone waits for it
            self.engine.arm_loss_event(self._loss_armed)
        if self.side is not None:
            # Where the next gather goes.  Single replica: behind this batch's phase-B launches in HOST order, waiting
            # for the loss event (both streams stay busy back to back, and the caller's stream carries no fence packet).
            # With a gradient exchange: right before the all-reduce (below), so that the
            # gather starts when the chain goes idle for the exchange and the chain's own kernels run beside less
            # of it — measured with a stand-in exchange on one GPU (bench.py --simulate-allreduce-us): placed ahead
            # of the projection kernels instead, the gather is over before the exchange starts and the exchange's
            # latency is fully exposed (a 60 us exchange: 286 vs 249 us per step).
            if issue_next is not None and loss_fence is None and not (self._exchange and self._gather_under_exchange):
                issue_next(None)
                issue_next = None
            self._apply_stream = torch.cuda.current_stream(self.device)
            if ev is not None and self._prewaited != slot and not ev.query():  # (a gather issued two batches ago has
                ev.wait(self._apply_stream)  # usually finished: then no wait packet goes into this stream at all)
            self._prewaited = None
            # The NEXT batch's gather must have finished before its forward starts.  In the fork backward this stream
            # waits for the engine's join anyway: the engine's second stream — idle by then — takes the wait for the
            # gather ahead of recording the join, and this stream carries one wait packet per step instead of two.
            if self._join_carries_gather and self._pending and local_B > 0:
                nslot, nev, *_ = self._pending[0]
                if nev is not None and not nev.query() and self.engine.set_join_wait(nev):
                    self._prewaited = nslot
        if self.device.type == "cuda":  # read by phase B on this stream, possibly allocated on a producer's stream
            for t in (neg_index, query_ids):
                if t is not None:
                    t.record_stream(torch.cuda.current_stream(self.device))
        ran = False
        if self.loss_kind == "inbatch":
            self._inbatch_fwd_bwd(slot, local_B, query_ids, k, global_B)
            ran = local_B > 0
        elif local_B > 0:
            self.engine.fwd_bwd(slot, local_B, self.margin, 1.0 / (global_B * k), float(global_B), neg_index=neg_index)
            ran = True
        fence = None
        if loss_fence is not None and ran:  # the engine has recorded it (in C) behind this batch's loss kernel
            loss_fence.recorded = True
            fence = loss_fence
        issue = (lambda: issue_next(fence)) if issue_next is not None else None
        if issue is not None and self.side is not None and not (self._exchange and self._gather_under_exchange):
            issue()
            issue = None
        if self._dry:
            # calibrate(): no optimizer, no counters.  With a gradient exchange the timed step must be the data-parallel
            # step as it will run — slab reduce -> (scratch) bucket, next gather issued, the exchange on this stream — or
            # the schedules are ranked beside a gather that, in training, sits under the exchange instead (round 5)
            if self._exchange and self._dry_bucket is not None and self._split:
                self._split_step(local_B, True, True, issue, dry=True)
                return
            if self._exchange and self._dry_bucket is not None:
                if local_B > 0:
                    self.engine.reduce_grads(self._dry_bucket, accumulate=False)
                if issue is not None:
                    issue()
                self._all_reduce(self._dry_bucket)
                return
            if issue is not None:
                issue()
            return
        self._micro += 1
        do_step = self._micro % k == 0
        if self.bucket is None:  # single replica, no accumulation: Adam folds the slab sum
            if local_B == 0:  # nothing was computed: the slabs still hold the previous batch's gradients
                if issue is not None:
                    issue()
                return
            self.step_count += 1
            self._prewait_next()
            if mark:
                mark(1)
                mark(2)  # no exchange on this path
            if self._auto_step:  # step counter on the device: the same call sequence is graph-replayable
                self.engine.adam_auto(self.lr, None, self.betas[0], self.betas[1], self.eps)
            else:
                self.engine.adam(self.step_count, self.lr, None, self.betas[0], self.betas[1], self.eps)
            if mark:
                mark(3)
            if issue is not None:
                issue()
            return
        first_micro = not self._carry  # nothing un-applied in the bucket: this batch's gradients overwrite it
        self._carry = not do_step
        if self._split:
            self._split_step(local_B, first_micro, do_step, issue, mark=mark)
            return
        if local_B > 0:
            self.engine.reduce_grads(self.bucket, accumulate=not first_micro)
        elif first_micro:  # an empty shard contributes zero gradients but still takes part in the all-reduce
            self.bucket.zero_()
        if issue is not None:
            issue()  # weight-independent; enqueued BEFORE the exchange so that its fence does not wait for it
        if mark:
            mark(1)  # forward + loss + backward + slab reduce done
        if do_step and self._exchange:
            self._all_reduce(self.bucket)
        if mark:
            mark(2)  # exchange done
        if do_step:
            self.step_count += 1
            self._prewait_next()
            self.engine.adam(self.step_count, self.lr, self.bucket, self.betas[0], self.betas[1], self.eps)
        if mark:
            mark(3)  # optimizer done

    def _trace_mark(self, i: int) -> None:
        """step_trace (a list, or None = off): every step appends [e0, e1, e2, e3] timing events recorded on the caller's
        stream at step start / after forward+loss+backward(+slab reduce) / after the exchange / after Adam — what
        bench.py's `dp_breakdown` is computed from.  Each record is an event packet on the critical chain (a few us): a
        traced pass is its own short run, never the timed region."""
        if i == 0:
            self.step_trace.append([None, None, None, None])
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream(self.device))
        self.step_trace[-1][i] = ev

    def _prewait_next(self) -> None:
        """The NEXT batch's gather must have finished before its forward starts.  If it has not yet (the host runs ahead
        of the GPU, so at enqueue time it usually has not), the wait goes into the caller's stream HERE, ahead of this
        step's Adam — in the fork backward right behind the join, where the stream stops for an event anyway — rather
        than between Adam and the next batch's first projection kernel."""
        if self.side is None or not self._pending:
            return
        slot, ev, *_ = self._pending[0]
        if self._prewaited == slot:  # rides on the engine's join (apply())
            return
        if ev is not None and not ev.query():
            ev.wait(torch.cuda.current_stream(self.device))
            self._prewaited = slot

    # -- checkpoint / resume of the optimizer side ---------------------------------------------
    def state_dict(self) -> dict:
        """Everything a later `load_state_dict` needs to continue this run exactly: the flat parameters (the
        model's `state_dict()` holds the same values under the reference's key names, backend/main.py:137-150
        saves only those), both Adam moments, the optimizer-step and micro-batch counters and the gradient
        bucket of an unfinished accumulation window (backend/training.py:99-102 carries it across epochs).
        The reference itself cannot resume: it saves no optimizer state."""
        if self._pending:
            raise RuntimeError("state_dict(): batches are still in flight; call it between epochs")
        cpu = lambda t: None if t is None else t.detach().to("cpu", copy=True)
        return {"flat": cpu(self.flat), "exp_avg": cpu(self.exp_avg), "exp_avg_sq": cpu(self.exp_avg_sq),
                "bucket": cpu(self.bucket) if self.accumulation_steps > 1 else None,
                "step_count": self.step_count, "micro": self._micro, "carry": self._carry, "D": self.D, "P": self.P,
                # the gather form calibrate() chose where it had a choice (forms differ in fp32 summation order)
                "k1_form": (self.calibration or {}).get("k1_form"),
                "hyper": {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "margin": self.margin,
                          "accumulation_steps": self.accumulation_steps}}

    def load_state_dict(self, sd: dict) -> None:
        """Inverse of `state_dict()`.  Under data parallelism load on every rank, or on one rank and call
        `sync_replicas()`."""
        if self._pending:
            raise RuntimeError("load_state_dict(): batches are still in flight")
        if sd["D"] != self.D or sd["P"] != self.P or sd["flat"].numel() != self.flat.numel():
            raise ValueError("checkpoint was written for another embedding / projection size")
        for name in ("flat", "exp_avg", "exp_avg_sq"):
            getattr(self, name).copy_(sd[name].to(self.device))   # in place: the engine and the modules alias these
        if sd.get("bucket") is not None and self.bucket is not None:
            self.bucket.copy_(sd["bucket"].to(self.device))
        self.step_count, self._micro = int(sd["step_count"]), int(sd["micro"])
        self._carry = bool(sd.get("carry", self._micro % self.accumulation_steps != 0))
        form = sd.get("k1_form")
        if form is not None and hasattr(self.engine, "set_tuning"):
            # the run being resumed had calibrate() choose between the binned and the streaming gather: keep ITS choice (a
            # later calibrate() on this trainer leaves the form alone)
            base = {k: v for k, v in dict(self.engine.tuning or {}).items() if k != "k1_form"}
            self.engine.set_tuning({**base, **({"k1_form": form} if form != "shape rule" else {})})
            self._k1_form_loaded = form
            self.calibration = {**(self.calibration or {}), "k1_form": form}
        if self._auto_step:
            self.engine.set_step(self.step_count)

    def advance(self, nxt: Optional[Batch] = None, global_B: Optional[int] = None) -> None:
        """One step of the pipeline: weight-dependent part of the oldest gathered batch, and the
        gather of `nxt` (the batch that enters the pipeline now, if any) — as one hipGraph replay
        when both are eligible, else eagerly (`apply` + `gather`)."""
        nxt = None if nxt is None else as_batch(nxt)
        entry = self._pending[0]
        if not self._graphable(entry, nxt):
            self.apply(global_B, issue_next=(lambda fence=None: self.gather(nxt, fence=fence)) if nxt is not None else None)
            return
        slot_fb, ev, B, _, _ = self._pending.popleft()
        self._prewaited = None  # a pre-wait belongs to the eager step that enqueued it (round-4 advisor finding)
        slot_k1 = None
        if nxt is not None:
            slot_k1 = self._next_slot
            self._next_slot = (slot_k1 + 1) % self.SLOTS
            self._ring_load(slot_k1, nxt)
        g = self._graph(slot_fb, slot_k1)
        if ev is not None:  # pooled by an eager K1 on the side stream
            ev.wait(torch.cuda.current_stream(self.device))
        g.replay()
        self._steps_done += 1
        self._micro += 1
        self.step_count += 1
        if slot_k1 is not None:
            self._pending.append((slot_k1, None, B, None, None))

    def abort_pipeline(self) -> None:
        """Forget every gathered-but-unapplied batch (an epoch that ended in an exception).  Their gathers complete on the
        side stream on their own; the slots are handed out again in stream order (gather() fences the side stream behind
        the caller's stream), so nothing has to be waited for here."""
        self._pending.clear()
        self._prewaited = None
        if hasattr(self.engine, "clear_join_wait"):
            self.engine.clear_join_wait()

    # -- public ------------------------------------------------------------------
    def train_step(self, batch: Batch, global_batch: Optional[int] = None) -> None:
        """One batch (this rank's shard).  No host synchronisation."""
        self.gather(batch)
        self.apply(global_batch)

    def train_epoch(self, batches: Iterable[Batch], global_batch_sizes: Optional[Sequence[int]] = None) -> float:
        """All batches of an epoch; returns the mean of the per-batch (global) losses
        like backend/training.py:63.  One host sync at the end."""
        if self.freeze_gc and not TripletTrainer._gc_frozen:
            # The loop keeps the host ~2x ahead of the GPU; a full pass of Python's cyclic GC over the ~10^6
            # objects torch's import leaves behind (60-90 ms) used to land in an epoch now and then and drain
            # that lead.  Freezing the existing objects once makes later passes look at new objects only.
            gc.collect()
            gc.freeze()
            TripletTrainer._gc_frozen = True
        self._micro = 0  # batch_idx of backend/training.py:80 restarts with the epoch; a carried gradient does not
        it = iter(batches)
        depth = self.SLOTS - 1 if self.side is not None else 1
        while len(self._pending) < depth:  # prime the pipeline
            b = next(it, None)
            if b is None:
                break
            self.gather(b)
        if not self._pending:
            return float("nan")
        i = 0
        while self._pending:
            gB = global_batch_sizes[i] if global_batch_sizes is not None else None
            self.advance(next(it, None), gB)
            i += 1
        return self.epoch_loss()

    def epoch_loss(self) -> float:
        loss_sum, n, _ = self.engine.read_stats(reset=True)
        if self.world > 1:  # per-rank sums are hinge_sum / B_global: adding them gives the global batch losses
            t = torch.tensor([loss_sum], dtype=torch.float64, device=self.device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)
            loss_sum = float(t.item())
            c = torch.tensor([n], dtype=torch.int64, device=self.device)  # a rank whose shard was empty counted fewer
            dist.all_reduce(c, op=dist.ReduceOp.MAX, group=self.pg)
            n = int(c.item())
        return loss_sum / max(n, 1)
