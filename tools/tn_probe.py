#!/usr/bin/env python3
"""Weight-gradient (TN, split-K) GEMM time vs rows, next to the forward pair (NT) on the same rows. Diagnostic;
run under rocprofv3 --kernel-trace --stats to separate the three launches of tt_proj_bwd_f32."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import two_towers_overlords_amd as pkg
from two_towers_overlords_amd import ops
pkg.load()
dev = torch.device("cuda")
D = P = 384
for M in (2048, 4096, 6144, 8192, 16384):
    x = torch.randn(M, D, device=dev); w1 = torch.randn(P, D, device=dev) * .05; b1 = torch.zeros(P, device=dev)
    w2 = torch.randn(P, P, device=dev) * .05; b2 = torch.zeros(P, device=dev)
    out, h = ops.proj_fwd(x, w1, b1, w2, b2)
    dout = torch.randn_like(out)
    for name, fn in (("fwd pair (2 NT)", lambda: ops.proj_fwd(x, w1, b1, w2, b2)), ("bwd (TN dW2 + NN dH + TN dW1)", lambda: ops.proj_bwd(dout, x, h, w2))):
        for _ in range(10): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"M={M:6d} {name:32s} {e0.elapsed_time(e1) * 20:7.1f} us", flush=True)
