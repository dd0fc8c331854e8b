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


if __name__ == "__main__" and "--eval" not in sys.argv:
    main()


def eval_bench():
    """f2: Q queries x N documents, P=384: cosine-score GEMM + NDCG@{10,5,1} on the device vs the
    reference's per-query torch.cosine_similarity + sklearn ndcg_score on the CPU."""
    import time

    from oracle import two_towers_oracle as orc

    Q, N, P = 200, 10000, 384
    g = torch.Generator().manual_seed(0)
    q, d = torch.randn(Q, P, generator=g), torch.randn(N, P, generator=g)
    rel = (torch.rand(Q, N, generator=g) < 0.001).float()
    qd, dd, rd = q.cuda(), d.cuda(), rel.cuda()
    t_s = timeit(lambda: ops.cosine_scores(qd, dd), n=20)
    sc = ops.cosine_scores(qd, dd)
    t_n = timeit(lambda: ops.ndcg_at_k(sc, rd, ks=(10, 5, 1)), n=20)
    t0 = time.perf_counter()
    s_cpu = orc.cosine_scores(q, d)
    for k in (10, 5, 1):
        orc.ndcg_rows(s_cpu.numpy(), rel.numpy(), k)
    t_cpu = (time.perf_counter() - t0) * 1e6
    print(f"eval Q={Q} N={N} P={P}: cosine_scores {t_s:8.1f} us, ndcg@10/5/1 {t_n:8.1f} us (incl. sync); "
          f"CPU torch+sklearn {t_cpu:10.0f} us  ({t_cpu / (t_s + t_n):.0f}x)")


if __name__ == "__main__" and "--eval" in sys.argv:
    eval_bench()
