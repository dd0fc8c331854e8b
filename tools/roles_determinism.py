#!/usr/bin/env python3
"""The drop-in route's three-stream pipeline (caller's stream: chain; side stream: gather; third stream: role packing) run
twice over N overlapped steps of role_blocks batches with ragged 0/1 masks must end in bitwise identical parameters — and
in the parameters of the same batches handed over as packed int32 ids with TT_PAD_ID padding (what the packing pass
writes).  A missing ordering edge between the three streams shows up here as a difference (diagnostic, round 5).
    python tools/roles_determinism.py [steps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import two_towers_overlords_amd as pkg  # noqa: F401
from two_towers_overlords_amd.training import TripletTrainer, TripletBatch
from two_towers_overlords_amd.engine import flat_param_slices
V, D, P, B = 30522, 384, 384, 2048
S = (40, 128, 112)  # every role padded to its own longest text
dev = torch.device("cuda")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
g = torch.Generator().manual_seed(0)
table = (torch.randn(V, D, generator=g) * 0.02).to(dev)
init = {n: (torch.rand(*sh, generator=g) * 0.1 - 0.05) for n, _, sh in flat_param_slices(D, P)}
blocks, packed = [], []
for _ in range(8):
    roles, rows = [], torch.full((3, B, max(S)), -(2 ** 31), dtype=torch.int32)
    for r, s in enumerate(S):
        ids = torch.randint(0, V, (B, s), generator=g)
        lens = torch.randint(1, s + 1, (B,), generator=g)
        mask = (torch.arange(s).unsqueeze(0) < lens.unsqueeze(1)).long()
        roles.append((ids.to(dev), mask.to(dev)))
        rows[r, :, :s] = torch.where(mask.bool(), ids, torch.full_like(ids, -(2 ** 31))).int()
    blocks.append(TripletBatch(role_blocks=roles))
    packed.append(TripletBatch(ids=rows.to(dev)))
out = {}
for name, batches, kw in (("roles, run 1", blocks, dict(pack_roles=True)), ("roles, run 2", blocks, dict(pack_roles=True)),
                          ("packed ids + TT_PAD_ID", packed, {}), ("roles on the gather stream (no third stream)", blocks, dict(pack_roles=True, event_scope=3))):
    tr = TripletTrainer(table=table, params={k: v.clone().to(dev) for k, v in init.items()}, max_batch=B, max_seq_len=max(S), **kw)
    loss = tr.train_epoch([batches[i % len(batches)] for i in range(steps)])
    torch.cuda.synchronize()
    out[name] = (loss, tr.flat.clone())
    print(f"{name:48s} loss {loss:.9f}  |flat| {float(tr.flat.double().abs().sum()):.9f}", flush=True)
    tr.engine.close()
ref = out["roles, run 1"]
for name, (loss, flat) in out.items():
    assert loss == ref[0] and torch.equal(flat, ref[1]), name
print("bitwise identical:", ", ".join(out))
