"""The data-parallel side of TripletTrainer: the gradient bucket's all-reduce in one
piece or two, replica synchronisation, and the in-batch-negatives loss with its one real activation exchange (all-gather of
document embeddings, all-reduce of their gradient).  A mixin: the method bodies are TripletTrainer's."""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.distributed as dist

from . import _lib, ops
from .trainer_types import shard_sizes


class ExchangeMixin:
    def _split_step(self, local_B: int, first_micro: bool, do_step: bool, issue_next, mark=None, dry: bool = False) -> None:
        """Slab reduce + exchange + Adam with the bucket cut in two (split_exchange=True).  `early` = dW2 / db2 of both
        towers: their slabs are complete when the engine records `_ev_dw2` (inside fwd_bwd, before dH -> dW1), so a
        second stream reduces and all-reduces them while the caller's stream is still computing dW1 / db1; `late` =
        dW1 / db1, reduced and exchanged on the caller's stream after that chain, as the whole bucket is otherwise."""
        eng, bucket = self.engine, (self._dry_bucket if dry else self.bucket)
        cuda = self.device.type == "cuda"
        main = torch.cuda.current_stream(self.device) if cuda else None

        def reduce(parts):
            for off, n in parts:
                if local_B > 0:
                    eng.reduce_grads_range(bucket, off, n, accumulate=not first_micro)
                elif first_micro:  # an empty shard contributes zeros but takes part in every collective
                    bucket[off: off + n].zero_()

        if cuda:
            if local_B > 0:
                self._comm.wait_event(self._ev_dw2)   # recorded by the engine once dW2 / db2 are complete
            else:
                self._comm.wait_stream(main)
            with torch.cuda.stream(self._comm):
                reduce(self._early)
                if do_step:
                    self._all_reduce_parts(self._early, bucket)
                self._ev_early.record(self._comm)
        else:
            reduce(self._early)
            if do_step:
                self._all_reduce_parts(self._early, bucket)
        reduce(self._late)
        if issue_next is not None:
            issue_next()
        if mark:
            mark(1)
        if do_step:
            self._all_reduce_parts(self._late, bucket)
        if cuda:
            main.wait_event(self._ev_early)  # before Adam — or, on a micro-batch, before the next fwd_bwd rewrites the slabs
        if mark:
            mark(2)
        if do_step and not dry:
            self.step_count += 1
            eng.adam(self.step_count, self.lr, bucket, self.betas[0], self.betas[1], self.eps)
        if mark:
            mark(3)

    def _all_reduce_parts(self, parts, bucket=None) -> None:
        """The views of one part of the flat bucket (W2 | b2 of each tower, or W1 | b1 of each), summed over the
        replicas: one public `dist.all_reduce` per view, issued back to back on the current stream (the early part's
        stream is off the critical chain; per-element sums do not depend on how the bucket is cut)."""
        bucket = self.bucket if bucket is None else bucket
        for off, n in parts:
            dist.all_reduce(bucket[off: off + n], op=dist.ReduceOp.SUM, group=self.pg, async_op=False)

    def _all_reduce(self, bucket: torch.Tensor) -> None:
        """The step's one gradient exchange: sum of the flat bucket over the replicas (RCCL over xGMI on GPUs).
        A synchronous-op call on purpose: torch's NCCL backend enqueues such a collective on the CALLER's stream,
        where an async_op=True call hops to the backend's own stream and back — two cross-queue event waits of
        10-15 us each on this hardware (tools/nccl_stream_probe.py: 45 vs 31 us per iteration), right on the critical
        chain reduce -> exchange -> Adam.  Nothing blocks on the host either way; the gather of the next batch runs
        beside it on the side stream."""
        dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.pg, async_op=False)

    # -- data parallel: replicas --------------------------------------------------------
    def _src0(self) -> int:
        return dist.get_global_rank(self.pg, 0) if self.pg is not None else 0

    def sync_replicas(self) -> None:
        """Every replica takes rank 0's parameters, Adam moments and step count (what
        DistributedDataParallel's constructor does for the parameters).  Called at construction —
        nn.Linear's default init comes from each process's own RNG, and identical all-reduced Adam
        updates never bring different starting points together — and after loading a checkpoint
        on one rank."""
        if self.world == 1:
            return
        src = self._src0()
        for t in (self.flat, self.exp_avg, self.exp_avg_sq):
            dist.broadcast(t, src=src, group=self.pg)
        if self.bucket is not None:
            dist.broadcast(self.bucket, src=src, group=self.pg)  # an unfinished accumulation window travels too
        steps = torch.tensor([self.step_count, self._micro, int(self._carry)], dtype=torch.int64, device=self.device)
        dist.broadcast(steps, src=src, group=self.pg)
        self.step_count, self._micro, self._carry = int(steps[0]), int(steps[1]), bool(steps[2])

    def replica_divergence(self) -> float:
        """max over ranks of |parameter checksum - rank 0's| (0.0 when the replicas are in step);
        a debug check, one small all-gather."""
        if self.world == 1:
            return 0.0
        mine = torch.stack([self.flat.double().sum(), self.flat.double().abs().sum()])
        allc = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(allc, mine, group=self.pg)
        return float(max((c - allc[0]).abs().max() for c in allc))

    def _inbatch_fwd_bwd(self, slot: int, B: int, query_ids, k: int, global_B: int) -> None:
        """In-batch-negatives loss between the two halves of the engine's phase B.  Data parallel
        (SURVEY.md §8e/f4): the documents' embeddings (and query ids) are all-gathered, each rank
        scores its queries against every document of the global batch, and the documents' gradient
        is summed over ranks before each rank keeps its own rows — the one real exchange of
        activations on this path.  Shards may be uneven (`shard_batch`: the first global_B % world
        ranks hold one row more) or empty; every rank takes part in every collective."""
        eng = self.engine
        if global_B < 2:
            raise ValueError("in-batch-negatives loss: a (global) batch of one has no negatives")
        if self.world == 1:
            eng.forward(slot, B, 2)
            emb, dout = eng.embeddings(B, 2), eng.d_out(B, 2)
            qid = None if query_ids is None else query_ids.to(torch.int64).contiguous()
            z = float(B * (B - 1))
            loss, _, _ = ops.inbatch_triplet_fwd_bwd(emb[0], emb[1], self.margin, None, qid, qid, 1.0 / (z * k), z,
                                                     dq_out=dout[0], dd_out=dout[1], workspace=self._inbatch_ws(B, B))
            eng.add_loss(loss)
            eng.backward(slot, B, 2)
            return
        W, r = self.world, self.rank
        sizes, offs = shard_sizes(global_B, W)
        if B != sizes[r]:
            raise ValueError(f"in-batch loss: rank {r} holds {B} rows of a global batch of {global_B}; the contiguous "
                             f"split (shard_batch) gives it {sizes[r]}")
        mb = sizes[0]  # largest shard: all-gathers run on equal, zero-padded blocks
        have_qid = query_ids is not None
        pad_d = torch.zeros((mb, self.P), dtype=torch.float32, device=self.device)
        pad_q = torch.zeros(mb, dtype=torch.int64, device=self.device) if have_qid else None
        if B > 0:
            eng.forward(slot, B, 2)
            emb, dout = eng.embeddings(B, 2), eng.d_out(B, 2)
            pad_d[:B].copy_(emb[1])
            if have_qid:
                pad_q[:B].copy_(query_ids.to(torch.int64))
        d_pad = torch.empty((W * mb, self.P), dtype=torch.float32, device=self.device)
        dist.all_gather_into_tensor(d_pad, pad_d, group=self.pg)
        even = sizes[-1] == mb
        rows = None if even else self._valid_rows(sizes, mb)
        d_all = d_pad if even else d_pad.index_select(0, rows)
        qid_all = None
        if have_qid:
            q_pad = torch.empty(W * mb, dtype=torch.int64, device=self.device)
            dist.all_gather_into_tensor(q_pad, pad_q, group=self.pg)
            qid_all = q_pad if even else q_pad.index_select(0, rows)
        z = float(global_B * (global_B - 1))
        if B > 0:
            pos = (torch.arange(B, device=self.device, dtype=torch.int32) + offs[r]).contiguous()
            loss, _, dd_all = ops.inbatch_triplet_fwd_bwd(emb[0], d_all.contiguous(), self.margin, pos,
                                                          pad_q[:B].contiguous() if have_qid else None,
                                                          None if qid_all is None else qid_all.contiguous(),
                                                          1.0 / (z * k), z, dq_out=dout[0], workspace=self._inbatch_ws(B, global_B))
        else:
            dd_all = torch.zeros((global_B, self.P), dtype=torch.float32, device=self.device)
        dist.all_reduce(dd_all, op=dist.ReduceOp.SUM, group=self.pg)
        if B > 0:
            dout[1].copy_(dd_all[offs[r]: offs[r] + B])
            eng.add_loss(loss)
            eng.backward(slot, B, 2)

    def _inbatch_ws(self, Bq: int, Bd: int) -> Optional[torch.Tensor]:
        """Grow-only workspace of the in-batch loss (the [Bq, Bd] similarity matrix and the gradient slabs), kept between
        steps: stream-ordered reuse on the stream phase B runs on."""
        if self.device.type != "cuda":
            return None
        need = (int(_lib.load().tt_inbatch_workspace_bytes(Bq, Bd, self.P)) + 3) // 4
        ws = getattr(self, "_inbatch_workspace", None)
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.float32, device=self.device)
            self._inbatch_workspace = ws
        return ws

    def _valid_rows(self, sizes, mb: int) -> torch.Tensor:
        """Row indices of the real (non-padding) rows inside the [world * mb] all-gathered blocks."""
        key = (tuple(sizes), mb)
        cache = getattr(self, "_valid_rows_cache", None)
        if cache is None or cache[0] != key:
            idx = torch.cat([torch.arange(n) + i * mb for i, n in enumerate(sizes)])
            cache = (key, idx.to(self.device))
            self._valid_rows_cache = cache
        return cache[1]
