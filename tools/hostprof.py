#!/usr/bin/env python3
"""cProfile of the host side of the step loop (diagnostic)."""
import cProfile, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import two_towers_overlords_amd as pkg
from two_towers_overlords_amd.training import TripletTrainer, TripletBatch
from two_towers_overlords_amd.engine import flat_param_slices
V, D, P, B, S = 30522, 384, 384, 2048, 128
dev = torch.device('cuda')
table = (torch.randn(V, D) * 0.02).to(dev)
params = {n: (torch.rand(*sh) * 0.1 - 0.05).to(dev) for n, _, sh in flat_param_slices(D, P)}
tr = TripletTrainer(table=table, params=params, max_batch=B, max_seq_len=S)
ids = torch.randint(0, V, (3, B, S), dtype=torch.int32, device=dev)
b = TripletBatch(ids=ids)
def loop(n):
    issued = [0]
    def feed():
        while issued[0] < n and len(tr._pending) < 2:
            tr.gather(b); issued[0] += 1
    feed()
    while tr._pending:
        tr.apply(None, issue_next=feed); feed()
loop(50); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); loop(1000); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
