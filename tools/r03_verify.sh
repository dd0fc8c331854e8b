#!/bin/bash
# Round-end verification the way the driver does it: the default bench line as the FIRST process on a fresh box, then
# smoke(), then the whole GPU test suite in one process.   -> gpurun_out/verify/
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/verify; mkdir -p $out
cd $root
python3 bench.py > $out/bench_default_first_process.json 2> $out/bench_default.err; rc0=$?
python3 -c "import json;j=json.loads(open('$out/bench_default_first_process.json').read().strip().splitlines()[-1]);r=j['roofline'];print('default:',round(j['value']/1e6,3),'M/s',round(j['ms_per_step']*1e3,1),'us K1 in-loop frac',round(r['frac'],3),'alone ms',round(r['alone']['avg_launch_ms'],4),'secondary',round(j['secondary']['ms_per_step'],3),'ms frac',round(j['secondary']['roofline']['frac'],3),'cpu',round(j['cpu_baseline']['value']),round(j['cpu_baseline_bert']['value'] or 0))"
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.txt 2>&1; rc1=$?
tail -2 $out/smoke.txt
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1; rc2=$?
tail -4 $out/pytest_gpu.txt | cut -c1-300
python3 bench.py --no-cpu-baseline > $out/bench_default_second.json 2>/dev/null
python3 -c "import json;j=json.loads(open('$out/bench_default_second.json').read().strip().splitlines()[-1]);r=j['roofline'];print('second:',round(j['value']/1e6,3),'M/s',round(j['ms_per_step']*1e3,1),'us alone ms',round(r['alone']['avg_launch_ms'],4))"
exit $((rc0 + rc1 + rc2))
