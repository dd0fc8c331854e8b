#!/bin/bash
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r05_s9; mkdir -p $out
cd $root
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-secondary --path module --steps 200 > $out/module_$i.json 2>/dev/null; python -c "
import json; j=json.loads(open('$out/module_$i.json').read().strip().splitlines()[-1]); print(round(j['ms_per_step']*1e3,1), j['calibration'] and (j['calibration']['backward'], j['calibration']['fence_at_loss'], {k: round(v*1e3,1) for k,v in j['calibration']['ms_per_step_by_fence'].items()}))"; done
python bench.py --no-cpu-baseline --no-secondary --steps 200 > $out/engine.json 2>/dev/null; python -c "
import json; j=json.loads(open('$out/engine.json').read().strip().splitlines()[-1]); print(round(j['ms_per_step']*1e3,1), j['calibration'] and (j['calibration']['backward'], j['calibration']['fence_at_loss'], {k: round(v*1e3,1) for k,v in j['calibration']['ms_per_step_by_fence'].items()}))"
