#!/usr/bin/env python3
"""How well does the gather (K1, own stream) overlap with each part of the chain?  For X in {projection forward,
projection backward, forward+loss+backward}: run nA gathers on one stream and nB X's on another, each sized to take
about the same time alone, then both together.  overlap factor = (T_A alone + T_B alone) / T_both: 1.0 = they merely
take turns, 2.0 = perfectly concurrent."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import two_towers_overlords_amd as pkg
from two_towers_overlords_amd import _lib
from two_towers_overlords_amd.engine import HotPathEngine, flat_param_slices

pkg.load()
V, D, P, B, S = 30522, 384, 384, 2048, 128
dev = torch.device("cuda")
table = (torch.randn(V, D) * 0.02).to(dev)
eng = HotPathEngine(table, P, B, S, k1_blocks_per_cu=_lib.TT_K1_SHARE_AUTO)
flat = torch.cat([(torch.rand(*sh) * 0.1 - 0.05).reshape(-1) for _, _, sh in flat_param_slices(D, P)]).to(dev)
eng.bind(flat, torch.zeros_like(flat), torch.zeros_like(flat))
ids = torch.randint(0, V, (3, B, S), dtype=torch.int32, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
eng.pool(0, ids); eng.pool(1, ids); eng.fwd_bwd(0, B, 0.3)
torch.cuda.synchronize()


def k1(n):
    for _ in range(n):
        eng.pool(1, ids, stream=sa.cuda_stream)


def run_b(kind, n):
    with torch.cuda.stream(sb):
        for _ in range(n):
            if kind == "forward":
                eng.forward(0, B, 3)
            elif kind == "backward":
                eng.backward(0, B, 3)
            else:
                eng.fwd_bwd(0, B, 0.3)


def timed(fa, fb):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if fa: fa()
    if fb: fb()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


for mode in ("fused", "fork"):
    eng.set_backward(mode)
    for kind in ("forward", "backward", "fwd_bwd"):
        # size both sides to ~20 ms alone
        ta1 = timed(lambda: k1(50), None) / 50
        tb1 = timed(None, lambda: run_b(kind, 50)) / 50
        nA, nB = int(20 / ta1), int(20 / tb1)
        for rep in range(2):
            ta = timed(lambda: k1(nA), None)
            tb = timed(None, lambda: run_b(kind, nB))
            tab = timed(lambda: k1(nA), lambda: run_b(kind, nB))
        print(f"bwd={mode:5s} X={kind:8s}: K1 alone {ta / nA * 1e3:6.1f} us, X alone {tb / nB * 1e3:6.1f} us; "
              f"{nA} K1 + {nB} X: alone {ta:5.1f} + {tb:5.1f} ms, together {tab:5.1f} ms -> overlap factor {(ta + tb) / tab:4.2f}")
