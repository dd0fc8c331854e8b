#!/bin/bash
O=gpurun_out/r04_ramp; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-secondary"
for w in 5 5 50 200; do
  TT_TRACE=step $B --steps 20 --warmup $w > $O/k20_w$w.json 2> $O/k20_w$w.err
  python -c "import json;r=json.load(open('$O/k20_w$w.json'));print('w=$w step %.1f us  eff warmup %d'%(r['ms_per_step']*1e3, r['effective_warmup_steps']))"
  grep "TRACE gpu" $O/k20_w$w.err | cut -c1-260
done
TT_TRACE=1 $B --steps 200 --warmup 5 > $O/k200_w5.json 2> $O/k200_w5.err
python -c "import json;r=json.load(open('$O/k200_w5.json'));print('k200 w5 step %.1f us'%(r['ms_per_step']*1e3))"; grep "TRACE gpu" $O/k200_w5.err | cut -c1-260
