#!/usr/bin/env python3
"""What an event packet costs the stream it sits on (MI355X, ROCm 7.2): a chain of ~25 us kernels with
(a) nothing between them, (b) an event record after each, (c) a wait on an already completed event of
another stream before each, (d) both.  Prints us per kernel."""
import time
import torch

dev = torch.device("cuda", 0)
a = torch.randn(1536, 1536, device=dev)
side = torch.cuda.Stream(dev)
N = 400


def run(record, wait):
    evs = [torch.cuda.Event() for _ in range(8)]
    done = torch.cuda.Event()
    with torch.cuda.stream(side):
        done.record()
    side.synchronize()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(N):
        if wait:
            torch.cuda.current_stream().wait_event(done)
        torch.mm(a, a)
        if record:
            evs[i % 8].record()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / N * 1e3


for _ in range(2):
    for name, r, w in (("plain", False, False), ("record after each", True, False), ("wait(completed) before each", False, True),
                       ("both", True, True)):
        print(f"{name:32s} {run(r, w):7.2f} us per kernel")
