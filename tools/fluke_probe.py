#!/usr/bin/env python3
"""Probe for the ~1-in-10 slow runs: K1-only loop with / without timing events, many short processes. Diagnostic."""
import os, subprocess, sys, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, time, torch
sys.path.insert(0, %r)
import two_towers_overlords_amd as pkg
from two_towers_overlords_amd import ops
pkg.load()
dev = torch.device("cuda")
V, D, N, S = 30522, 384, 6144, 128
table = (torch.randn(V, D) * 0.02).to(dev)
g = torch.Generator().manual_seed(0)
ids = [torch.randint(0, V, (N, S), generator=g, dtype=torch.int32).to(dev) for _ in range(4)]
out = torch.empty(N, D, device=dev)
every = int(os.environ.get("EVERY", "0"))
for i in range(20): ops.gather_pool_norm(table, ids[i %% 4], out=out)
torch.cuda.synchronize()
evs = []
t0 = time.perf_counter()
for i in range(300):
    timed = every and i %% every == 0
    if timed:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    ops.gather_pool_norm(table, ids[i %% 4], out=out)
    if timed:
        e1.record(); evs.append((e0, e1))
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(round((t2 - t0) / 300 * 1e6, 1), round((t1 - t0) / 300 * 1e6, 1))
''' % root
for every in (0, 4, 0, 4, 0, 4):
    res = []
    for rep in range(8):
        env = dict(os.environ, EVERY=str(every))
        o = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        res.append(o.stdout.strip().split()[0] if o.stdout.strip() else "ERR")
    print("events every", every, ":", " ".join(res), flush=True)
