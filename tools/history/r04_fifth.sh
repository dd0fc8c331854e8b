#!/bin/bash
O=gpurun_out/r04_fifth; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo exit=$? >> $O/tests.log; tail -3 $O/tests.log
B="python bench.py --no-cpu-baseline --no-secondary"
for spec in "c2:" "c3:--config c3" "inb4096:--loss inbatch --batch 4096" "inb2048:--loss inbatch" "index:--inputs index" "cache:--inputs cache" "ragged:--ragged" "c4:--config c4 --steps 40 --warmup 8"; do
  tag=${spec%%:*}; args=${spec#*:}
  $B $args > $O/$tag.json 2> $O/$tag.err
  python -c "import json;r=json.load(open('$O/$tag.json'));print('$tag %.1f us host %.1f'%(r['ms_per_step']*1e3,r['host_issue_ms_per_step']*1e3), r.get('calibration'))"
done
