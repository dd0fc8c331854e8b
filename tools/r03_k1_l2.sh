#!/bin/bash
# Round 3: L2 hit rate of the two XCD-sliced K1 forms at configs[1], counted directly (TCC_HIT / TCC_MISS summed over the
# channels; kernels alone: `bench.py --k1-only`; counter collection serialises kernels anyway).
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/k1l2; mkdir -p $out
rm -f $out/k1_l2_hit.txt
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; label=$2
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/pmc_$tag -o p -- python3 $root/bench.py --config c2 --k1-only --no-cpu-baseline --steps 8 --warmup 2 > /dev/null 2>&1 || return 1
  python3 - $out/pmc_$tag "$label" <<'PY' >> $out/k1_l2_hit.txt
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "pool_" in n:
            acc[n.split("<")[0].split("::")[-1]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("==", sys.argv[2])
for k, c in sorted(acc.items()):
    m = {n: sum(v[len(v) // 2:]) / max(len(v[len(v) // 2:]), 1) for n, v in c.items()}
    h, ms = m.get("TCC_HIT_sum", 0), m.get("TCC_MISS_sum", 0)
    print(f"{k:28s} TCC_HIT {h:12.0f}  TCC_MISS {ms:12.0f}  hit rate {h / max(h + ms, 1):.3f}  TCC_REQ {m.get('TCC_REQ_sum', 0):12.0f}  launches {len(c.get('TCC_HIT_sum', []))}")
PY
  rm -rf $out/pmc_$tag; }
run cols "column slices (default)"
TT_K1_COLS=0 run rows "row slices (TT_K1_COLS=0)"
TT_K1_SLICED=0 run pertext "per-text form (TT_K1_SLICED=0)"
cat $out/k1_l2_hit.txt
