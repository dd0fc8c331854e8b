#!/usr/bin/env python3
"""Per-step timeline from a rocprofv3 --kernel-trace CSV: for a few steady-state steps, every kernel with its
queue, start offset and duration (us), so the overlap of the gather stream and the projection chain is visible.
usage: tools/timeline.py <ks_kernel_trace.csv> [first_step] [n_steps]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 120
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
def short(n):
    tail = n.split("gemm_f32")[-1]
    if "pool_binned" in n: return "K1 gather (binned)"
    if "pool_bincombine" in n: return "K1 combine (binned)"
    if "pool_bin_kernel" in n: return "K1 sort by class"
    if "pool_colsliced" in n: return "K1 gather (columns)"
    if "pool_colnorm" in n: return "K1 norm"
    if "pool_sliced" in n: return "K1 gather (rows)"
    if "pool_combine" in n: return "K1 combine"
    if "pool_norm_stream" in n: return "K1 stream"
    if "pool_norm" in n: return "K1 per-text"
    if "gemm_tn_nn" in n: return "dW2+dH (TN+NN)"
    if "adam_multi" in n: return "adam (multi)"
    if "adam_kernel" in n: return "adam"
    if "triplet_kernel" in n: return "triplet+mean" if "true" in n else "triplet"
    if "loss_mean" in n: return "loss mean"
    if "reduce_slabs" in n: return "reduce slabs"
    if "gemm_f32" in n:
        lay = tail.split("<")[1][:1] if "<" in tail else "?"
        return {"0": "gemm NT", "1": "gemm NN", "2": "gemm TN"}.get(lay, "gemm") + (" bk16" if "bk16" in n else "")
    return n[:30]


ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], short(r["Kernel_Name"])) for r in rows]
ks.sort()
adam = [i for i, k in enumerate(ks) if k[3].startswith("adam")]
if len(adam) < first + nsteps + 1:
    first = max(0, len(adam) - nsteps - 1)
lo, hi = ks[adam[first]][1], ks[adam[first + nsteps]][1]
t0 = lo
print(f"# steps {first + 1}..{first + nsteps}: {(hi - lo) / 1e3 / nsteps:.1f} us per step; columns: queue, start (us), duration (us), kernel")
for s, e, q, n in ks:
    if s >= lo - 200_000 and e <= hi:
        if e < lo:
            continue
        print(f"q{q:>2s} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  {n}")
