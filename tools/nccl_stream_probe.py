#!/usr/bin/env python3
"""Does torch's NCCL(RCCL) backend run a collective issued with async_op=False on the CALLER's stream (no hop to its
internal stream and back)?  One rank on one GPU: a chain of (small kernel -> 1-rank all_gather_into_tensor -> small
kernel) per iteration; a cross-stream hop costs ~10-15 us each way on this hardware (profiles/r02 timeline), so the
per-iteration GPU time tells the two implementations apart."""
import os, time
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = torch.randn(600_000, device="cuda")
y = torch.empty_like(x)
a = torch.randn(256, 256, device="cuda")


def loop(n, mode):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(n):
        a.add_(1.0)
        if mode == "async+wait":
            w = dist.all_gather_into_tensor(y, x, async_op=True)
            w.wait()
        elif mode == "sync":
            dist.all_gather_into_tensor(y, x, async_op=False)
        a.add_(1.0)
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / n * 1e3


for mode in ("none", "async+wait", "sync", "none", "async+wait", "sync"):
    loop(20, mode)
    print(f"{mode:11s}: {loop(300, mode):7.1f} us per iteration (GPU time)")
dist.destroy_process_group()
