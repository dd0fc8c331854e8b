#!/bin/bash
# Round 3 probes at configs[1]: 96x96 forward tiles (gemm16, NT only) where the chain has the chip to itself,
# and the in-batch loss alone (kernel stats).   usage (GPU box): tools/r03_probe.sh -> gpurun_out/probe/
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/probe; mkdir -p $out
cd $root
pick='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j["ms_per_step"]*1e3,1), "us")'
rm -f $out/g16.txt
for rep in 1 2; do
echo "cache auto(g16 NT): $(python3 bench.py --no-cpu-baseline --no-secondary --inputs cache | python3 -c "$pick")" >> $out/g16.txt
echo "cache TT_G16=0: $(TT_G16=0 python3 bench.py --no-cpu-baseline --no-secondary --inputs cache | python3 -c "$pick")" >> $out/g16.txt
echo "no-overlap auto: $(python3 bench.py --no-cpu-baseline --no-secondary --no-overlap | python3 -c "$pick")" >> $out/g16.txt
echo "no-overlap TT_G16=0: $(TT_G16=0 python3 bench.py --no-cpu-baseline --no-secondary --no-overlap | python3 -c "$pick")" >> $out/g16.txt
echo "default TT_G16=1 (NT only): $(TT_G16=1 python3 bench.py --no-cpu-baseline --no-secondary | python3 -c "$pick")" >> $out/g16.txt
echo "default: $(python3 bench.py --no-cpu-baseline --no-secondary | python3 -c "$pick")" >> $out/g16.txt
done
echo "inbatch b4096 no-overlap: $(python3 bench.py --no-cpu-baseline --no-secondary --loss inbatch --batch 4096 --no-overlap | python3 -c "$pick")" >> $out/g16.txt
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
ks inbatch4096_alone --loss inbatch --batch 4096 --no-overlap
