#!/usr/bin/env python3
"""Two identical training runs of N overlapped steps at the metric shape end in bitwise identical parameters (diagnostic)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import two_towers_overlords_amd as pkg
from two_towers_overlords_amd.training import TripletTrainer, TripletBatch
from two_towers_overlords_amd.engine import flat_param_slices
V, D, P, B, S = 30522, 384, 384, 2048, 128
dev = torch.device("cuda")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
g = torch.Generator().manual_seed(0)
table = (torch.randn(V, D, generator=g) * 0.02).to(dev)
init = {n: (torch.rand(*sh, generator=g) * 0.1 - 0.05) for n, _, sh in flat_param_slices(D, P)}
batches = [TripletBatch(ids=torch.randint(0, V, (3, B, S), generator=g, dtype=torch.int32).to(dev)) for _ in range(16)]
out = []
for run in range(2):
    tr = TripletTrainer(table=table, params={k: v.clone().to(dev) for k, v in init.items()}, max_batch=B, max_seq_len=S)
    loss = tr.train_epoch(batches * (steps // 16))
    torch.cuda.synchronize()
    out.append((loss, tr.flat.clone()))
    print(f"run {run}: loss {loss!r} |params| {out[-1][1].norm().item():.9f}", flush=True)
same = out[0][0] == out[1][0] and torch.equal(out[0][1], out[1][1])
print("bitwise identical" if same else "DIFFERENT")
sys.exit(0 if same else 1)
