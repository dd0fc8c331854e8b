"""hipGraph replay of whole steps (opt-in: on ROCm 7.2 / MI355X replaying the 3-stream step graph costs the host more than
issuing its launches one by one — TripletTrainer.__init__).  A mixin: the method bodies are TripletTrainer's."""
from __future__ import annotations

from typing import Optional

import torch

from .trainer_types import TripletBatch


class GraphReplayMixin:
    def _graphable(self, entry, nxt: Optional[TripletBatch]) -> bool:
        if not self.use_graphs:
            return False
        if self.eager_every and self._steps_done % self.eager_every == self.eager_every - 1:
            return False
        _, _, B, neg_index, _ = entry
        if neg_index is not None or B != self.engine.max_B or self.loss_kind != "triplet":
            return False
        if nxt is None:
            return True
        return (nxt.ids is not None and nxt.ids.shape[0] == 3 and nxt.ids.shape[1] == B
                and nxt.ids.shape[2] <= self.engine.max_S and nxt.mask is None and nxt.neg_index is None
                and nxt.cache_rows is None and nxt.token_rows is None)

    def _ring_load(self, slot: int, nxt: TripletBatch) -> None:
        """Stage a batch's ids in the fixed ring buffer the captured K1 launch reads (main stream)."""
        if self._ring_ids is None:
            B, S = self.engine.max_B, self.engine.max_S
            self._ring_ids = torch.zeros((self.SLOTS, 3, B, S), dtype=torch.int32, device=self.device)
            self._ring_lens = torch.zeros((self.SLOTS, 3, B), dtype=torch.int32, device=self.device)
        S = nxt.ids.shape[2]
        if nxt.ids.data_ptr() != self._ring_ids[slot].data_ptr():  # a caller may fill the ring slot itself
            self._ring_ids[slot][:, :, :S].copy_(nxt.ids)
        if nxt.lens is not None:
            self._ring_lens[slot].copy_(nxt.lens)
            self._ring_fill[slot] = None
        elif self._ring_fill[slot] != S:
            self._ring_lens[slot].fill_(S)
            self._ring_fill[slot] = S

    def _graph(self, slot_fb: int, slot_k1: Optional[int]):
        key = (slot_fb, slot_k1)
        g = self._graphs.get(key)
        if g is not None:
            return g
        B = self.engine.max_B
        if self._ring_ids is None and slot_k1 is not None:
            raise RuntimeError("ring not initialised")
        # Nothing but this step's launches may happen between capture_begin and capture_end: a cyclic-garbage collection
        # in there can run the finaliser of some dead engine or tensor (device synchronise, frees) and invalidate the capture.
        import gc

        gc.collect()
        was_enabled = gc.isenabled()
        gc.disable()
        try:
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            cap = torch.cuda.Stream(self.device)
            with torch.cuda.graph(g, stream=cap):
                if slot_k1 is not None:  # branch: the gather of the batch two steps ahead
                    self.side.wait_stream(cap)
                    self.engine.pool(slot_k1, self._ring_ids[slot_k1], None, self._ring_lens[slot_k1],
                                     stream=self.side.cuda_stream)
                self.engine.fwd_bwd(slot_fb, B, self.margin, 1.0 / B, float(B))
                self.engine.adam_auto(self.lr, None, self.betas[0], self.betas[1], self.eps)
                if slot_k1 is not None:
                    cap.wait_stream(self.side)
        finally:
            if was_enabled:
                gc.enable()
        self._graphs[key] = g
        return g
