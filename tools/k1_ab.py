#!/usr/bin/env python3
"""A/B of the two K1 forms in ONE process, interleaved rounds (guide rule 24): register
accumulation (product) vs LDS-DMA staged tiles (TT_K1_LDS=1 selects it at launch time)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import two_towers_overlords_amd as pkg
from two_towers_overlords_amd import ops
pkg.load()
dev = torch.device("cuda")
for (V, D, S, N) in ((30522, 384, 128, 6144), (250000, 384, 128, 6144)):
    table = (torch.randn(V, D) * 0.02).to(dev)
    ids = torch.randint(0, V, (N, S), dtype=torch.int32, device=dev)
    out_a, out_b = torch.empty(N, D, device=dev), torch.empty(N, D, device=dev)
    res = {"reg": [], "lds": []}
    for rnd in range(12):
        for name, out in (("reg", out_a), ("lds", out_b)):
            if name == "lds":
                os.environ["TT_K1_LDS"] = "1"
            else:
                os.environ.pop("TT_K1_LDS", None)
            for _ in range(3):
                ops.gather_pool_norm(table, ids, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.gather_pool_norm(table, ids, out=out)
            e1.record(); torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) / 10 * 1e3)
    by = N * (S * D * 4 + S * 4 + D * 4)
    assert torch.equal(out_a, out_b), "the two forms must agree bit for bit (same summation order)"
    for name in ("reg", "lds"):
        r = sorted(res[name]); med = r[len(r) // 2]
        print(f"V={V} D={D} S={S} N={N} {name}: median {med:7.1f} us  min {r[0]:7.1f} us  -> {by / med / 1e3:7.0f} GB/s algorithmic")
