#!/usr/bin/env python3
"""Lifecycle stress: engines created / used / destroyed repeatedly, two trainers interleaved on one device. Diagnostic."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import two_towers_overlords_amd as pkg
from two_towers_overlords_amd.training import TripletTrainer, TripletBatch
from two_towers_overlords_amd.engine import flat_param_slices
pkg.load()
dev = torch.device("cuda")
V, D, P, B, S = 30522, 384, 384, 512, 128
table = (torch.randn(V, D) * 0.02).to(dev)
g = torch.Generator().manual_seed(0)
mk = lambda: {n: (torch.rand(*sh, generator=g) * 0.1 - 0.05).to(dev) for n, _, sh in flat_param_slices(D, P)}
batches = [TripletBatch(ids=torch.randint(0, V, (3, B, S), generator=g, dtype=torch.int32).to(dev)) for _ in range(6)]
mem0 = None
for it in range(30):
    a = TripletTrainer(table=table, params=mk(), max_batch=B, max_seq_len=S)
    b = TripletTrainer(table=table, params=mk(), max_batch=B, max_seq_len=S, overlap_streams=False)
    la, lb = a.train_epoch(batches), b.train_epoch(batches)
    la2 = a.train_epoch(batches)
    assert la == la and lb == lb and la2 < la + 1e-3, (la, lb, la2)
    a.engine.close(); b.engine.close()
    del a, b
    torch.cuda.synchronize()
    if it == 2: mem0 = torch.cuda.memory_allocated()
    if it % 10 == 9: print(f"iter {it}: losses {la:.5f} {lb:.5f} {la2:.5f} mem {torch.cuda.memory_allocated() / 1e6:.0f} MB (ref {mem0 / 1e6:.0f})", flush=True)
assert torch.cuda.memory_allocated() <= mem0 * 1.05 + 1e6
print("lifecycle ok")
