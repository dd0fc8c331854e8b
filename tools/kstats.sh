#!/bin/bash
# usage: tools/kstats.sh TAG [bench args]  — rocprofv3 kernel stats of one bench run into gpurun_out/ks_TAG.txt
tag=$1; shift
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/ks_$tag -o ks -- python3 $root/bench.py --no-cpu-baseline "$@" > $root/gpurun_out/ks_$tag.json 2>/dev/null || exit 1
python3 - $root/gpurun_out/ks_$tag <<'PY' > $root/gpurun_out/ks_$tag.txt
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if int(r['Calls'])>=50: print(f"{r['Name'][:100]:100s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:8.1f}")
PY
