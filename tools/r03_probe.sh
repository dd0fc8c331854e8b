#!/bin/bash
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/probe; mkdir -p $out
cd $root
pick='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j["ms_per_step"]*1e3,1), "us")'
for kb in 0 40 100; do
  echo "cache TT_G16=1 pad=$kb: $(TT_G16=1 TT_G16_LDS_KB=$kb python3 bench.py --no-cpu-baseline --no-secondary --inputs cache | python3 -c "$pick")" >> $out/g16.txt
done
echo "cache TT_G16=0: $(TT_G16=0 python3 bench.py --no-cpu-baseline --no-secondary --inputs cache | python3 -c "$pick")" >> $out/g16.txt
cat $out/g16.txt
cd /tmp && export TMPDIR=/tmp
ks() { tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks_$tag -o ks -- python3 $root/bench.py --no-cpu-baseline --no-secondary "$@" > /dev/null 2>&1 || return 1
  python3 - $out/ks_$tag <<'PY' > $out/ks_$tag.txt
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if int(r['Calls'])>=50: print(f"{r['Name'][:110]:110s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:8.1f}")
PY
  rm -rf $out/ks_$tag; echo "== $tag"; cat $out/ks_$tag.txt; }
ks inbatch4096 --loss inbatch --batch 4096
TT_G16=1 ks cache_g16 --inputs cache
TT_G16=1 TT_G16_LDS_KB=100 ks cache_g16_pad100 --inputs cache
