import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import two_towers_overlords_amd as pkg
from two_towers_overlords_amd import ops
pkg.load()
def run(V, D, N, S):
    g = torch.Generator().manual_seed(0)
    table = (torch.randn(V, D, generator=g) * 0.02).cuda()
    ids = torch.randint(0, V, (N, S), generator=g, dtype=torch.int32).cuda()
    out = torch.empty(N, D, device="cuda")
    for form in ("columns", "binned", "per_text"):
        for _ in range(5): ops.gather_pool_norm(table, ids, form=form, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): ops.gather_pool_norm(table, ids, form=form, out=out)
        e1.record(); torch.cuda.synchronize()
        print(f"V={V} D={D} N={N} S={S} {form:9s} {e0.elapsed_time(e1) / 50 * 1e3:8.1f} us", flush=True)
run(30522, 384, 6144, 128)
run(30522, 384, 12288, 128)
run(100000, 384, 6144, 128)
