#!/bin/bash
# Round 3: per-kernel times of the in-batch step (B = 4096, one stream) and the configs[1] cache-fed step with the exact-fp32
# MFMA kernels and with the staged split-bf16 kernels (TT_BF3=2).   -> gpurun_out/bf3stats/
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/bf3stats; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$tag -o ks -- python3 $root/bench.py --no-cpu-baseline --no-secondary "$@" > /dev/null 2>&1 || return 1
  python3 - $(find $out/prof_$tag -name "*kernel_stats.csv" | head -1) > $out/$tag.txt <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]:
    n = re.sub(r"\(anonymous namespace\)::|ttg::|void ", "", r["Name"])[:96]
    print(f"{n:96s} calls {int(r['Calls']):5d} avg_us {float(r['AverageNs'])/1e3:8.1f}")
PY
  rm -rf $out/prof_$tag; echo "== $tag"; cat $out/$tag.txt; }
TT_BF3=0 run inbatch_fp32 --loss inbatch --batch 4096 --no-overlap
TT_BF3=2 run inbatch_bf3s --loss inbatch --batch 4096 --no-overlap
TT_BF3=0 run cache_fp32 --inputs cache
TT_BF3=2 run cache_bf3s --inputs cache
