#!/bin/bash
O=gpurun_out/r04_fourth; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo exit=$? >> $O/tests.log; tail -3 $O/tests.log
B="python bench.py --no-cpu-baseline --no-secondary"
for i in 1 2; do
TT_TRACE=1 $B --loss inbatch --batch 4096 > $O/inbatch4096_$i.json 2> $O/inbatch4096_$i.err
python -c "import json;r=json.load(open('$O/inbatch4096_$i.json'));print('inbatch4096 %.1f us host %.1f'%(r['ms_per_step']*1e3,r['host_issue_ms_per_step']*1e3))"
done
$B --loss inbatch --batch 4096 --event-scope 3 > $O/inbatch4096_ev3.json 2>/dev/null
python -c "import json;r=json.load(open('$O/inbatch4096_ev3.json'));print('inbatch4096 ev3 %.1f us host %.1f'%(r['ms_per_step']*1e3,r['host_issue_ms_per_step']*1e3))"
python tools/hostprof_bench.py --no-cpu-baseline --no-secondary --loss inbatch --batch 4096 --steps 100 > /dev/null 2> $O/hostprof_inbatch4096.txt; head -50 $O/hostprof_inbatch4096.txt
python tools/hostprof_bench.py --path module --steps 100 --warmup 20 > $O/module_prof.json 2> $O/hostprof_module.txt; head -60 $O/hostprof_module.txt
$B --path module --steps 200 --warmup 20 > $O/c2_module.json 2>/dev/null; python -c "import json;r=json.load(open('$O/c2_module.json'));print('module path %.2f M/s %.1f us'%(r['value']/1e6,r['ms_per_step']*1e3))"
