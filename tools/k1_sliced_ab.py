#!/usr/bin/env python3
"""A/B: per-text K1 vs XCD-sliced K1, alone on the chip (diagnostic)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import two_towers_overlords_amd as pkg
from two_towers_overlords_amd import ops
pkg.load()
dev = torch.device("cuda")
D = int(os.environ.get("D", "384"))
CASES = [tuple(int(x) for x in c.split(",")) for c in os.environ["CASES"].split(";")] if os.environ.get("CASES") else None
for (V, N, S) in CASES or [(30522, 6144, 128), (30522, 6144, 64), (30522, 1536, 512), (2000, 6144, 128), (5000, 6144, 128),
                  (10000, 6144, 128), (20000, 6144, 128), (120000, 6144, 128), (30522, 6144, 32), (30522, 512, 128),
                  (30522, 64, 128)]:
    table = (torch.randn(V, D) * 0.02).to(dev)
    g = torch.Generator().manual_seed(0)
    ids = [torch.randint(0, V, (N, S), generator=g, dtype=torch.int32).to(dev) for _ in range(4)]
    if os.environ.get("ZIPF"):
        ids = [(torch.exp(torch.rand(N, S, generator=g, dtype=torch.float64) * torch.log(torch.tensor(float(V)))) - 1).long()
               .clamp_(0, V - 1).int().to(dev) for _ in range(4)]
    out = torch.empty(N, D, device=dev)
    a = ops.gather_pool_norm(table, ids[0], sliced=False); b = ops.gather_pool_norm(table, ids[0], sliced=True)
    print(f"V={V} N={N} S={S}: max|sliced - per-text| = {(a - b).abs().max().item():.3e}")
    for name, kw in [("per-text", {"sliced": False}), ("sliced", {"sliced": True}), ("per-text", {"sliced": False}), ("sliced", {"sliced": True})]:
        for i in range(5):
            ops.gather_pool_norm(table, ids[i % 4], out=out, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(40):
            ops.gather_pool_norm(table, ids[i % 4], out=out, **kw)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 40 * 1e3
        print(f"   {name:9s} {us:7.1f} us  {N * S * D * 4 / us / 1e6:6.2f} TB/s algorithmic", flush=True)
