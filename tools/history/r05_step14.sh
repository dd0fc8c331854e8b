#!/bin/bash
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r05_s14; mkdir -p $out
cd $root
for r in 1 2 3 4 5; do
for v in true false; do
python bench.py --no-cpu-baseline --no-secondary --e2e --steps 1000 --trainer pack_roles=$v > $out/e2e_$v.json 2>/dev/null
python -c "
import json; j=json.loads(open('$out/e2e_$v.json').read().strip().splitlines()[-1]); print('staged=$v e2e rows3', round(j['value']/1e6,3), 'M/s', round(j['ms_per_step']*1e3,1), 'us/step', 'K1 in loop', round(j['roofline']['avg_launch_ms']*1e3,1))"
done; done
