#!/usr/bin/env python3
"""K1 on SHORT texts (queries: a handful of tokens) — per-text vs XCD-sliced form, row stride S=128 and tight S."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import two_towers_overlords_amd as pkg
from two_towers_overlords_amd import ops
pkg.load()
V, D = 30522, 384
dev = torch.device("cuda")
table = (torch.randn(V, D) * 0.02).to(dev)
g = torch.Generator().manual_seed(0)


def t(fn, n=50):
    for _ in range(5): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / n * 1e3


for N, lo, hi, S in ((2048, 4, 16, 128), (2048, 4, 16, 16), (2048, 8, 32, 128), (2048, 8, 32, 32), (4096, 32, 128, 128), (6144, 128, 128, 128)):
    ids = torch.randint(0, V, (N, S), generator=g, dtype=torch.int32).to(dev)
    lens = torch.randint(lo, hi + 1, (N,), generator=g, dtype=torch.int32).clamp_(max=S).to(dev)
    out = torch.empty(N, D, device=dev)
    a = t(lambda: ops.gather_pool_norm(table, ids, lens=lens, out=out, sliced=False))
    b = t(lambda: ops.gather_pool_norm(table, ids, lens=lens, out=out, sliced=True))
    rows = int(lens.sum())
    print(f"N={N} len U{{{lo}..{hi}}} S={S}: per-text {a:6.1f} us, sliced {b:6.1f} us  ({rows * D * 4 / 1e6:.0f} MB of rows)")
