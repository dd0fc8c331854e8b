#!/usr/bin/env python3
"""Repeated identical training runs of N overlapped steps at the metric shape end in bitwise identical parameters — and
the same parameters whichever way the gather stream is fenced (an event of its own / the engine's loss event) and whichever
backward schedule runs: a missing ordering edge between the streams shows up here as a difference (diagnostic).
    python tools/determinism_probe.py [steps] [triplet|inbatch]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import two_towers_overlords_amd as pkg
from two_towers_overlords_amd.training import TripletTrainer, TripletBatch
from two_towers_overlords_amd.engine import flat_param_slices
V, D, P, B, S = 30522, 384, 384, 2048, 128
dev = torch.device("cuda")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
loss_kind = sys.argv[2] if len(sys.argv) > 2 else "triplet"
g = torch.Generator().manual_seed(0)
table = (torch.randn(V, D, generator=g) * 0.02).to(dev)
init = {n: (torch.rand(*sh, generator=g) * 0.1 - 0.05) for n, _, sh in flat_param_slices(D, P)}
roles = 3 if loss_kind == "triplet" else 2
batches = [TripletBatch(ids=torch.randint(0, V, (roles, B, S), generator=g, dtype=torch.int32).to(dev)) for _ in range(16)]
out = {}
for fence in (False, True, True, False):
    for bwd in ("fork", "fused"):
        tr = TripletTrainer(table=table, params={k: v.clone().to(dev) for k, v in init.items()}, max_batch=B, max_seq_len=S,
                            loss=loss_kind, fence_at_loss=fence, tuning={"backward": bwd})
        loss = tr.train_epoch(batches * (steps // 16))
        torch.cuda.synchronize()
        out.setdefault((fence, bwd), []).append((loss, tr.flat.clone()))
        print(f"fence_at_loss={fence} backward={bwd}: loss {loss!r} |params| {tr.flat.norm().item():.9f}", flush=True)
        tr.engine.close()
ref = out[(False, "fork")][0][1]
same_params = all(torch.equal(ref, p) for runs in out.values() for _, p in runs)
same_loss_per_schedule = all(len({l for l, _ in out[(f, b)]} | {l for l, _ in out[(not f, b)]}) == 1 for f in (False,) for b in ("fork", "fused"))
print("parameters bitwise identical over all runs, fences and schedules" if same_params else "PARAMETERS DIFFER")
print("losses identical per schedule" if same_loss_per_schedule else "LOSSES DIFFER within a schedule")
sys.exit(0 if (same_params and same_loss_per_schedule) else 1)
