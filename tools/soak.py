#!/usr/bin/env python3
"""Soak: many steps of the default pipeline; checks for leaks, drift into NaN and stalls (diagnostic)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import two_towers_overlords_amd as pkg
from two_towers_overlords_amd.training import TripletTrainer, TripletBatch
from two_towers_overlords_amd.engine import flat_param_slices
V, D, P, B, S = 30522, 384, 384, 2048, 128
dev = torch.device("cuda")
table = (torch.randn(V, D) * 0.02).to(dev)
params = {n: (torch.rand(*sh) * 0.1 - 0.05).to(dev) for n, _, sh in flat_param_slices(D, P)}
tr = TripletTrainer(table=table, params=params, max_batch=B, max_seq_len=S)
g = torch.Generator().manual_seed(0)
batches = [TripletBatch(ids=torch.randint(0, V, (3, B, S), generator=g, dtype=torch.int32).to(dev)) for _ in range(16)]
n_epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
mem0 = torch.cuda.memory_allocated()
t0 = time.perf_counter()
for ep in range(n_epochs):
    loss = tr.train_epoch(batches * 32)  # 512 steps per epoch
    if ep % 5 == 0 or ep == n_epochs - 1:
        print(f"epoch {ep}: loss {loss:.6f} steps {tr.step_count} mem +{(torch.cuda.memory_allocated() - mem0) / 1e6:.1f} MB "
              f"{(ep + 1) * 512 * B / (time.perf_counter() - t0) / 1e6:.2f} M triplets/s", flush=True)
    assert loss == loss and torch.isfinite(tr.flat).all()
print("soak ok")
