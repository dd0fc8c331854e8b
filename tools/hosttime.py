#!/usr/bin/env python3
"""Host-issue time vs total time per step (diagnostic)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import two_towers_overlords_amd as pkg
from two_towers_overlords_amd.training import TripletTrainer, TripletBatch
from two_towers_overlords_amd.engine import flat_param_slices
V, D, P, B, S = 30522, 384, 384, 2048, 128
dev = torch.device('cuda')
table = (torch.randn(V, D) * 0.02).to(dev)
params = {n: (torch.rand(*sh) * 0.1 - 0.05).to(dev) for n, _, sh in flat_param_slices(D, P)}
use_stream = len(sys.argv) > 1 and sys.argv[1] == "stream"
ctx = torch.cuda.stream(torch.cuda.Stream()) if use_stream else torch.cuda.stream(torch.cuda.current_stream())
with ctx:
    for overlap in (False, True):
        tr = TripletTrainer(table=table, params={k: v.clone() for k, v in params.items()}, max_batch=B, max_seq_len=S, overlap_streams=overlap)
        ids = torch.randint(0, V, (3, B, S), dtype=torch.int32, device=dev)
        b = TripletBatch(ids=ids)
        for rep in range(12):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 200
            issued = 0
            depth = 2 if overlap else 1
            def feed():
                global issued
                while issued < n and len(tr._pending) < depth:
                    tr.gather(b); issued += 1
            feed()
            while tr._pending:
                tr.apply(None, issue_next=feed); feed()
            t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
            print(f"overlap={overlap} rep {rep}: host {1e6*(t1-t0)/n:6.1f} us/step, total {1e6*(t2-t0)/n:6.1f} us/step", flush=True)
