#!/bin/bash
# per-kernel durations of the chain with the K-split tiles (cache-fed step: no gather beside it)
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r05_s17; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in base ks; do
  t=""; [ $v = ks ] && t="--tune proj_tile=33"
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/c2_$v -o p -- python3 $root/bench.py --no-cpu-baseline --no-secondary --config c2 --inputs cache --steps 100 --warmup 20 $t > $out/c2_$v.json 2> $out/c2_$v.err
  f=$(find $out/c2_$v -name "*kernel_stats.csv" | head -1); echo "== c2 $v"; [ -n "$f" ] && cut -d, -f1-4 "$f" < /dev/null | head -12
  t=""; [ $v = ks ] && t="--tune proj_tile=32"
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/c3_$v -o p -- python3 $root/bench.py --no-cpu-baseline --no-secondary --config c3 --inputs cache --steps 100 --warmup 20 $t > $out/c3_$v.json 2> $out/c3_$v.err
  f=$(find $out/c3_$v -name "*kernel_stats.csv" | head -1); echo "== c3 $v"; [ -n "$f" ] && cut -d, -f1-4 "$f" < /dev/null | head -12
done
