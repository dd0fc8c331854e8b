#!/bin/bash
# usage: tools/ab_grid.sh "VAR1=a,b,c" "VAR2=x,y" [bench args]  — every combination once per repetition (2 reps), one box
v1=${1%%=*}; l1=${1#*=}; v2=${2%%=*}; l2=${2#*=}; shift 2
out=gpurun_out/ab_grid.txt
for rep in 1 2; do for a in ${l1//,/ }; do for b in ${l2//,/ }; do
  r=$(env $v1=$a $v2=$b python bench.py --no-cpu-baseline --no-calibrate --warmup 100 "$@" | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['ms_per_step']*1e3,1), round(j['roofline']['avg_launch_ms']*1e3,1) if j['roofline'] else '')") || exit 1
  echo "$v1=$a $v2=$b $* : $r" | tee -a $out
done; done; done
