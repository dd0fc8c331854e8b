"""TEST INFRASTRUCTURE: start `world` ranks of a worker function and collect ONE result without ever
blocking the pytest session: the result queue is read with a timeout, ranks are joined and whatever is
still alive afterwards is killed (by the exact Process objects started here).  A worker reports with
`q.put((tag, payload, extra))`; on an exception it puts ("error", text, None) itself — a rank that dies in
native code (or hangs in a collective) before it can report shows up as the timeout."""
import os
import queue as _queue

import torch.multiprocessing as mp


def free_port(base: int = 29000, span: int = 3000, salt: int = 0) -> int:
    return base + (os.getpid() * 7 + salt * 131) % span


def run_ranks(worker, world: int, args=(), timeout: float = 420.0, join: float = 60.0, port: int = None):
    """Runs worker(rank, world, port, *args, q) in `world` spawned processes; returns the first queue item.
    Fails (AssertionError) on an "error" item, on a timeout and on a non-zero exit code of any rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port(salt=world + len(args)) if port is None else port
    procs = [ctx.Process(target=worker, args=(r, world, port, *args, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        out = q.get(timeout=timeout)
    except _queue.Empty:
        out = ("error", f"no result from {world} rank(s) of {worker.__name__} within {timeout:.0f} s "
                        f"(exit codes so far: {[p.exitcode for p in procs]})", None)
    for p in procs:
        p.join(join)
        if p.is_alive():
            p.kill()
            p.join(10)
    assert out[0] != "error", out[1]
    assert all(p.exitcode == 0 for p in procs), f"exit codes {[p.exitcode for p in procs]}"
    return out
