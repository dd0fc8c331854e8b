#!/usr/bin/env python3
"""Micro-timings of the individual C-ABI ops on one GPU (diagnostic; not the headline bench)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import two_towers_overlords_amd as pkg  # noqa: E402
from two_towers_overlords_amd import ops  # noqa: E402


def timeit(fn, n=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=6144)
    ap.add_argument("--D", type=int, default=384)
    ap.add_argument("--P", type=int, default=384)
    ap.add_argument("--S", type=int, default=128)
    ap.add_argument("--V", type=int, default=30522)
    a = ap.parse_args()
    pkg.load()
    dev = torch.device("cuda")
    M, D, P = a.M, a.D, a.P
    x = torch.nn.functional.normalize(torch.randn(M, D, device=dev), dim=1)
    w1, b1 = torch.randn(P, D, device=dev) * 0.05, torch.randn(P, device=dev) * 0.05
    w2, b2 = torch.randn(P, P, device=dev) * 0.05, torch.randn(P, device=dev) * 0.05
    out, h = ops.proj_fwd(x, w1, b1, w2, b2)
    d_out = torch.randn(M, P, device=dev) / M
    t = timeit(lambda: ops.proj_fwd(x, w1, b1, w2, b2))
    fl = 2 * M * P * (D + P)
    print(f"proj_fwd  M={M} D={D} P={P}: {t:8.1f} us  {fl / t / 1e6:7.1f} TFLOP/s (incl. 2 torch.empty)")
    t = timeit(lambda: ops.proj_bwd(d_out, x, h, w2))
    fl = 2 * M * P * (P + P + D)
    print(f"proj_bwd  M={M} D={D} P={P}: {t:8.1f} us  {fl / t / 1e6:7.1f} TFLOP/s (incl. reduce launches)")
    table = torch.randn(a.V, D, device=dev) * 0.02
    ids = torch.randint(0, a.V, (M, a.S), device=dev, dtype=torch.int32)
    o = torch.empty(M, D, device=dev)
    t = timeit(lambda: ops.gather_pool_norm(table, ids, out=o))
    by = M * (a.S * D * 4 + a.S * 4 + D * 4)
    print(f"gather_pool_norm N={M} S={a.S} D={D} V={a.V}: {t:8.1f} us  {by / t / 1e3:7.1f} GB/s algorithmic")
    q = torch.randn(M // 3, P, device=dev)
    t = timeit(lambda: ops.triplet_fwd_bwd(q, q, q, 0.3))
    print(f"triplet B={M // 3} P={P}: {t:8.1f} us")


if __name__ == "__main__":
    main()
