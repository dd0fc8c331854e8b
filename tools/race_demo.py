#!/usr/bin/env python3
"""Shows that the hazard tests/test_gpu_model_api.py::test_gather_is_ordered_behind_batches_produced_on_the_callers_stream
guards against is real on this hardware: the same late-arriving, immediately dropped batches through a trainer whose
gather() is round 1's (side stream not fenced behind the producer, batch tensors neither kept nor record_stream-ed)
train on wrong data; the shipped trainer reproduces the pre-staged run bit for bit."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import two_towers_oracle as orc
from two_towers_overlords_amd.training import TripletTrainer, as_batch


class Round1Gather(TripletTrainer):
    def gather(self, batch):
        slot = self._next_slot
        self._next_slot = (slot + 1) % self.SLOTS
        batch = as_batch(batch)
        self._timed_pool(slot, batch, self.side)       # no fence on the producer's stream, no record_stream
        ev = self._slot_ready[slot]
        ev.record(self.side)
        self._pending.append((slot, ev, batch.size, batch.neg_index, batch.query_ids))


V, D, P, B, S, NB = 5000, 384, 64, 256, 64, 24
table = orc.synthetic_table(V, D, seed=1) * 30
params = orc.init_params(D, P, seed=2)
host = []
for i in range(NB):
    b = orc.synthetic_batch(B, S, V, seed=900 + i, ragged=True)
    host.append((torch.stack([r[0] for r in b]).int().pin_memory(), torch.stack([r[1].sum(1) for r in b]).int().pin_memory()))
mk = lambda cls: cls(table=table.cuda(), params={k: v.cuda() for k, v in params.items()}, max_batch=B, max_seq_len=S,
                     margin=0.3, lr=1e-3, overlap_streams=True)
staged = [(i.cuda(), None, l.cuda()) for i, l in host]
torch.cuda.synchronize()
ref = mk(TripletTrainer)
ref_loss = ref.train_epoch(staged)
busy = torch.randn(4096, 4096, device="cuda")


def late():
    for ids_h, lens_h in host:
        for _ in range(6):
            busy @ busy
        ids, lens = ids_h.to("cuda", non_blocking=True), lens_h.to("cuda", non_blocking=True)
        yield (ids, None, lens)
        del ids, lens
        s1 = torch.full((3, B, S), V - 1, dtype=torch.int32, device="cuda")
        s2 = torch.full((3, B), 1, dtype=torch.int32, device="cuda")
        del s1, s2


for name, cls in (("shipped trainer", TripletTrainer), ("round-1 gather()", Round1Gather)):
    tr = mk(cls)
    try:
        loss = tr.train_epoch(late())
    except Exception as e:  # the engine reports ids outside the table: K1 read memory the copy had not filled yet
        torch.cuda.synchronize()
        print(f"{name:18s}: {type(e).__name__}: {e}")
        continue
    print(f"{name:18s}: loss {loss:.9f} (pre-staged {ref_loss:.9f}); parameters "
          f"{'identical' if torch.equal(tr.flat, ref.flat) else 'DIFFER, max |d| = %.3e' % float((tr.flat - ref.flat).abs().max())}")
