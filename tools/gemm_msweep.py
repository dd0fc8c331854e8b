#!/usr/bin/env python3
"""M-sweep of the forward projection pair at D=P=384 (diagnostic): how does the time move with the
number of 64x64 tiles per CU (256 CUs: 512 tiles = 2 per CU, 768 = 3 per CU)?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import two_towers_overlords_amd as pkg
from two_towers_overlords_amd import ops
pkg.load()
dev = torch.device("cuda")
D = P = 384
for M in (2048, 2688, 2752, 4096, 5440, 5504, 6144, 8128, 8192, 8256, 10880, 10944):
    x = torch.randn(M, D, device=dev); w1 = torch.randn(P, D, device=dev) * .05; b1 = torch.zeros(P, device=dev)
    w2 = torch.randn(P, P, device=dev) * .05; b2 = torch.zeros(P, device=dev)
    for _ in range(20):
        ops.proj_fwd(x, w1, b1, w2, b2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        ops.proj_fwd(x, w1, b1, w2, b2)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 100 * 1e3
    tiles = ((M + 63) // 64) * 6
    print(f"M={M:6d} tiles64={tiles:5d} ({tiles / 256:.2f}/CU): fwd pair {us:6.1f} us  {2 * 2 * M * D * P / us / 1e6:6.1f} TF", flush=True)
