#!/bin/bash
# usage: tools/ab_flag.sh VAR [bench args...]  — interleaved A/B on one box: VAR unset vs VAR=1 (3 repetitions each)
var=$1; shift
out=gpurun_out/ab_flag.txt
for rep in 1 2 3; do for v in off on; do
  if [ $v = on ]; then export $var=1; else unset $var; fi
  r=$(python bench.py --no-cpu-baseline "$@" | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['ms_per_step']*1e3,1))") || exit 1
  echo "$var=$v $* : $r" | tee -a $out
done; done
unset $var
