#!/usr/bin/env python3
"""K-sweep of the NT projection GEMM (diagnostic): run under rocprofv3 --kernel-trace --stats."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import two_towers_overlords_amd as pkg
from two_towers_overlords_amd import ops
pkg.load()
dev = torch.device("cuda")
M, P = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 384
for D in (32, 64, 128, 256, 384, 768, 1536):
    x = torch.randn(M, D, device=dev); w1 = torch.randn(P, D, device=dev) * .05; b1 = torch.zeros(P, device=dev)
    w2 = torch.randn(P, P, device=dev) * .05; b2 = torch.zeros(P, device=dev)
    for _ in range(20):
        ops.proj_fwd(x, w1, b1, w2, b2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.proj_fwd(x, w1, b1, w2, b2)
    e1.record(); torch.cuda.synchronize()
    print(f"M={M} D={D}: fwd pair {e0.elapsed_time(e1) / 50 * 1e3:.1f} us")
