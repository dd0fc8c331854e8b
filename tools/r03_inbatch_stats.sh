#!/bin/bash
# Round 3: per-kernel times of the in-batch step (B = 4096) with the round's final kernels, beside the gathers and on one stream.
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/inbatch_stats; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$tag -o ks -- python3 $root/bench.py --no-cpu-baseline --no-secondary --loss inbatch --batch 4096 "$@" > $out/$tag.json 2>/dev/null || return 1
  python3 - $(find $out/prof_$tag -name "*kernel_stats.csv" | head -1) > $out/$tag.txt <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:16]:
    n = re.sub(r"\(anonymous namespace\)::|ttg::|void ", "", r["Name"])[:100]
    print(f"{n:100s} calls {int(r['Calls']):5d} avg_us {float(r['AverageNs'])/1e3:8.1f}")
PY
  rm -rf $out/prof_$tag; echo "== $tag"; cat $out/$tag.txt; }
run default
run one_stream --no-overlap
