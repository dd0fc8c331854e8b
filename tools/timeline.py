#!/usr/bin/env python3
"""Per-step timeline from a rocprofv3 --kernel-trace CSV: for a few steady-state steps, every kernel with its
queue, start offset and duration (us), so the overlap of the gather stream and the projection chain is visible.
usage: tools/timeline.py <ks_kernel_trace.csv> [first_step] [n_steps]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 120
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
short = lambda n: ("K1 gather" if "pool_sliced" in n else "K1 combine" if "pool_combine" in n else "K1 stream" if "pool_norm_stream" in n else "K1 per-text" if "pool_norm" in n else
                   "dW2+dH (TN+NN)" if "gemm_tn_nn" in n else "adam" if "adam_kernel" in n else "triplet+mean" if "triplet_kernel" in n and "true" in n else
                   "triplet" if "triplet_kernel" in n else "loss mean" if "loss_mean" in n else
                   ("gemm TN" if "<2," in n.split("gemm_f32_kernel")[-1][:4] else "gemm NN" if "<1," in n.split("gemm_f32_kernel")[-1][:4] else "gemm NT") if "gemm_f32_kernel" in n else n[:30])
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], short(r["Kernel_Name"])) for r in rows]
ks.sort()
adam = [i for i, k in enumerate(ks) if k[3] == "adam"]
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
