#!/bin/bash
# usage: tools/ab_env.sh VAR "val1 val2" [bench args...]   — interleaved A/B of one env var on one box (diagnostic)
var=$1; vals=$2; shift 2
out=gpurun_out/ab_env.txt
for rep in 1 2 3; do for v in $vals; do
  r=$(env $var=$v python bench.py --no-cpu-baseline "$@" | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['ms_per_step']*1e3,1), round(j['roofline']['achieved']))") || exit 1
  echo "$var=$v $* : $r" >> $out
done; done
